#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) per kernel: count, total, avg, min, max."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.2f ms" % tot)
print("%-100s %7s %10s %6s %9s %9s %9s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-100s %7d %10.2f %6.1f %9.1f %9.1f %9.1f" % (r[0][:100], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
