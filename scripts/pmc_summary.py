#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 `--pmc` passes (FETCH_SIZE, WRITE_SIZE; csv output).

usage: pmc_summary.py <fetch_dir> <write_dir> <out.csv>

Both counters are reported by rocprofv3 in KiB-like units of 1024 B per count (derived from the L2's memory-side
request counters).  The gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section) is applied to
the read side: FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is taken as reported
(uncalibrated, the guide says so).  Output: mean bytes per launch for each kernel and launch shape (workgroups).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def launch_shape(r):
    """workgroups of the launch (a bench run launches the same kernel at several batch sizes: one row per shape)"""
    try:
        return int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    except (KeyError, ValueError):
        return 0


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:90], launch_shape(r))
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "workgroups", "launches", "fetch_bytes_per_launch(x2 gfx950 corr.)", "write_bytes_per_launch", "hbm_bytes_per_launch"])
        for k in sorted(set(fe) | set(wr)):
            n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0], 1)
            f = 2.0 * 1024.0 * fe.get(k, [0, 0.0])[1] / max(fe.get(k, [1, 0])[0], 1)
            x = 1024.0 * wr.get(k, [0, 0.0])[1] / max(wr.get(k, [1, 0])[0], 1)
            w.writerow([k[0], k[1], n, f"{f:.0f}", f"{x:.0f}", f"{f + x:.0f}"])
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
