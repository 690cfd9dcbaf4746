#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one rocprofv3 `--pmc` pass (csv output), plus the ratios DESIGN.md quotes.

usage: pmc_sq_summary.py <pmc_dir> <out.csv>          (one row per kernel and launch shape = workgroups)

Counters (one pass, 8 SQ slots): SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY.  Ratios:
  valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of wave-resident cycles spent issuing VALU work)
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES           (waves parked on s_waitcnt / barriers)
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                try:
                    shape = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)      # one row per launch shape (batch size)
                except (KeyError, ValueError):
                    shape = 0
                k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:90], shape)
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    names = sorted({c for k in acc for c in acc[k]})
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "workgroups", "launches"] + [n + "_per_launch" for n in names] + ["valu_active_frac", "wait_frac", "mfma_busy_frac"])
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", [0, 0.0])[1]):
            m = {n: (acc[k][n][1] / max(acc[k][n][0], 1)) for n in names}
            wc, bc = m.get("SQ_WAVE_CYCLES", 0.0), m.get("SQ_BUSY_CYCLES", 0.0)
            w.writerow([k[0], k[1], max(v[0] for v in acc[k].values())] + [f"{m[n]:.0f}" for n in names] +
                       [f"{m.get('SQ_ACTIVE_INST_VALU', 0.0) / wc:.3f}" if wc else "", f"{m.get('SQ_WAIT_ANY', 0.0) / wc:.3f}" if wc else "",
                        f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / bc:.3f}" if bc else ""])
    print(open(out).read()[:4000])


if __name__ == "__main__":
    main()
