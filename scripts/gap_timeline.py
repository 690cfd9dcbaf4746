#!/usr/bin/env python3
"""Launch boundaries of a SINGLE-STREAM run, measured from INSIDE the kernels (an -DIHMR_TIMELINE build: every workgroup records kind,
waves, start, end on the constant 100 MHz clock): records are cut into launches (consecutive records of one kind), and for every
launch boundary the table gives

    gap      = first workgroup start of launch i + 1  -  last workgroup end of launch i      (what no workgroup of either kernel sees)
    ramp     = last workgroup START of a launch - its first start                             (dispatch of the grid)
    span     = last workgroup end - first start                                               (the kernel as the chip sees it)
    life     = mean workgroup lifetime

A rocprofv3 kernel trace cannot give this: its `start` of a dependent dispatch is the moment the previous one ended (gaps of 0.00 us,
scripts/gap_table.py), i.e. its durations CONTAIN the boundary.

usage: IHMR_HIP_LIBRARY=build/timeline.so IHMR_TIMELINE_OUT=/tmp/tl.npy python3 bench.py --steps 1 --warmup 1 --streams 1 --fuse 1 \
           --no-cpu-baseline --no-extras --no-work-counters ;  python3 scripts/gap_timeline.py /tmp/tl.npy out.csv"""
import csv
import sys
from collections import OrderedDict

import numpy as np

KIND = {1: "prep", 2: "dist", 3: "tail<f,f>", 4: "tail<t,f>", 5: "tail<t,t>", 6: "skin", 7: "adam_skel", 8: "bwd2", 9: "bwd3", 10: "sample_loss"}


def main():
    rec = np.load(sys.argv[1])
    out = sys.argv[2]
    kind = (rec[:, 0] >> np.uint64(58)).astype(int)
    t0 = (rec[:, 0] & np.uint64((1 << 52) - 1)).astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    order = np.argsort(t0, kind="stable")
    kind, t0, t1 = kind[order], t0[order], t1[order]
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # us (100 MHz)
    # launches: maximal runs of one kind in start order (single stream: kernels do not overlap); a run is also cut where a workgroup
    # starts after every earlier one of the run has ended AND more than 1 us later (two launches of the same kernel back to back)
    launches = []
    s = 0
    run_end = t1[0]
    for i in range(1, len(kind) + 1):
        cut = i == len(kind) or kind[i] != kind[s] or t0[i] > run_end + 1.0
        if cut:
            launches.append((kind[s], t0[s:i].min(), t0[s:i].max(), t1[s:i].max(), float((t1[s:i] - t0[s:i]).mean()), i - s))
            s = i
            if i < len(kind):
                run_end = t1[i]
        else:
            run_end = max(run_end, t1[i])
    gaps, kstat = OrderedDict(), OrderedDict()
    for (k0, a0, l0, e0, m0, n0), (k1, a1, l1, e1, m1, n1) in zip(launches, launches[1:]):
        g = a1 - e0
        if g > 40.0:                                    # a pause between graph replays / stages, not a boundary
            continue
        gaps.setdefault((k0, n0, k1, n1), []).append(g)
    for k, a, l, e, m, n in launches:
        kstat.setdefault((k, n), []).append((l - a, e - a, m))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["from", "to", "count", "gap_mean_us", "gap_median_us", "gap_p10_us", "gap_p90_us"])
        tot = 0.0
        for (k0, n0, k1, n1), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
            g = np.array(g)
            tot += g.sum()
            w.writerow([f"{KIND.get(k0, k0)} x{n0}", f"{KIND.get(k1, k1)} x{n1}", len(g), f"{g.mean():.3f}", f"{np.median(g):.3f}",
                        f"{np.percentile(g, 10):.3f}", f"{np.percentile(g, 90):.3f}"])
        w.writerow([])
        w.writerow(["kernel", "workgroups", "launches", "ramp_mean_us", "span_mean_us", "workgroup_life_mean_us"])
        tspan = 0.0
        for (k, n), v in sorted(kstat.items(), key=lambda kv: -len(kv[1])):
            v = np.array(v)
            tspan += v[:, 1].sum()
            w.writerow([KIND.get(k, k), n, len(v), f"{v[:, 0].mean():.3f}", f"{v[:, 1].mean():.3f}", f"{v[:, 2].mean():.3f}"])
        w.writerow([])
        w.writerow(["total_gap_us", f"{tot:.1f}", "total_span_us", f"{tspan:.1f}", "gap_share", f"{tot / max(tot + tspan, 1e-9):.4f}", "launches", len(launches)])
    print(open(out).read())


if __name__ == "__main__":
    main()
