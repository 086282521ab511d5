#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into the per-kernel summary CSV kept under profiles/.
usage: scripts/rocpd_stats.py <trace_results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
    "max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "LDS"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / tot), r[4], r[5], r[6], r[7], r[8], r[9]])
print(open(sys.argv[2]).read()[:3000])
