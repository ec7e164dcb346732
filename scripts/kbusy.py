#!/usr/bin/env python3
"""GPU busy fraction (union of kernel intervals) and per-kernel totals of a rocprofv3 --kernel-trace database over the window
[first launch of <substring>, end]:  kbusy.py <db> <substring>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end from kernels order by start"))
first = min(i for i, r in enumerate(rows) if sys.argv[2] in r[0])
rows = rows[first:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_s, cur_e = 0, rows[0][1], rows[0][2]
for _, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.1f ms, busy (union) %.1f ms = %.3f, sum of kernels %.1f ms" % ((t1 - t0) / 1e6, busy / 1e6, busy / (t1 - t0), sum(e - s for _, s, e in rows) / 1e6))
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n[:90], [0, 0]); a[0] += 1; a[1] += e - s
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %8.1f ms x%5d  %s" % (a[1] / 1e6, a[0], n))
