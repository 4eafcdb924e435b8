#!/usr/bin/env python3
"""Per-kernel summary (calls, average and total time) of a rocprofv3 --kernel-trace run from its
rocpd database (<name>_results.db), as a markdown table.  usage: prof_db_summary.py DB PASSES [TITLE]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
passes = int(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]
rows = list(db.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc"))
tot = sum(r[3] for r in rows)
print("# %s\n" % title)
print("total kernel time %.3f ms over %d hot-path passes (%.3f ms/pass), %d launches per pass\n"
      % (tot / 1e6, passes, tot / 1e6 / passes, round(sum(r[1] for r in rows) / passes)))
print("| kernel | calls | avg us | total ms | % |\n|---|---:|---:|---:|---:|")
for n, c, a, t in rows:
    if t / tot < 0.002:
        continue
    print("| `%s` | %d | %.1f | %.3f | %.2f |" % (n[:100].replace("|", "/"), c, a / 1e3, t / 1e6, 100 * t / tot))
