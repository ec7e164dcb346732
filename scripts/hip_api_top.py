#!/usr/bin/env python3
"""Longest HIP API calls of a rocprofv3 --hip-trace database (rocpd sqlite):  hip_api_top.py <db> [n] [min_ms]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "region" in t.lower() or "api" in t.lower()]
print("tables:", cand[:20])
for t in ("regions", "rocpd_region", "api"):
    if t in tabs:
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % t)]
        print(t, cols)
try:
    rows = list(con.execute("select name, start, end from regions order by (end-start) desc limit %d" % n))
    t0 = list(con.execute("select min(start) from regions"))[0][0]
    for name, s, e in rows:
        print("%10.3f ms  at %10.3f ms  %s" % ((e - s) / 1e6, (s - t0) / 1e6, name))
except Exception as ex:
    print("query failed:", ex)
