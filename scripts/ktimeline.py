#!/usr/bin/env python3
"""Time-ordered kernel list of a rocprofv3 --kernel-trace database (rocpd sqlite): start offset, duration and the idle gap in
front of every kernel, for the window that starts at the LAST launch of <first-kernel substring>:
    ktimeline.py <db> <first-kernel substring> [max rows]"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
first = max(i for i, r in enumerate(rows) if sys.argv[2] in r[0])
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
t0 = rows[first][1]
prev_end = t0
busy = 0.0
agg = {}
for name, s, e in rows[first:first + lim]:
    print("%9.1f us  +%8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:110]))
    busy += (e - s) / 1e3
    a = agg.setdefault(name[:70], [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0.0, (s - prev_end) / 1e3)
    prev_end = max(prev_end, e)
print("window %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, busy))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-70s x%4d  %9.1f us   gaps in front %8.1f us" % (k, a[0], a[1], a[2]))
