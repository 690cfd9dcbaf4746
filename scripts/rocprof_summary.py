#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as CSV: one row per kernel AND launch
shape (workgroups x workgroup size) -- a bench run launches the same kernel at several batch sizes (a one-batch warm-up step next to
the eight-batch timed steps), and an average over both describes neither.
usage: rocprof_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute(
    "select name, grid_x / workgroup_x, grid_y / workgroup_y, grid_z / workgroup_z, workgroup_x, count(*), sum(duration) / 1000.0, "
    "avg(duration) / 1000.0 from kernels group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc"))
total = sum(r[6] for r in rows) or 1.0
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "workgroups", "threads", "calls", "total_us", "avg_us", "percent"])
    for name, gx, gy, gz, wx, calls, tot, avg in rows:
        short = name.split("(")[0].replace("void ", "")
        grid = "x".join(str(int(g)) for g in (gx, gy, gz) if g and int(g) > 1) or "1"
        w.writerow([short[:90], grid, int(wx), calls, f"{tot:.1f}", f"{avg:.3f}", f"{100.0 * tot / total:.3f}"])
print(open(out).read()[:2500])
