#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as CSV.
usage: rocprof_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        w.writerow([short[:90], calls, f"{tot:.1f}", f"{avg:.3f}", f"{pct:.3f}"])
print(open(out).read()[:1500])
