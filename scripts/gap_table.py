#!/usr/bin/env python3
"""The time BETWEEN kernels: end(i) -> start(i + 1) of consecutive dispatches on one queue, from a rocprofv3 kernel trace (rocpd
sqlite), grouped by boundary kind (kernel i -> kernel i + 1, with their launch shapes), next to the kernels' own durations.

usage: gap_table.py <results.db> <out.csv> [--max-gap-us 50] [--grid-filter WG]
  --max-gap-us   gaps longer than this are host-side pauses between graph replays, not launch boundaries: counted, not averaged
  --grid-filter  keep only boundaries whose two kernels belong to a launch sequence of this many samples (matches the tail kernel's
                 workgroup count; 0 = everything)

Round 6: decomposes the 12.4 us per refinement iteration that is in no kernel at batch 64."""
import csv
import sqlite3
import sys
from collections import OrderedDict


def short(name):
    s = name.split("(")[0].replace("void ", "")
    for a, b in (("sdf_prep_kernel", "prep"), ("sdf_dist_kernel", "dist"), ("opt_tail_kernel", "tail"), ("lbs_skin_kernel", "skin"),
                 ("opt_adam_skel_kernel", "adam_skel"), ("lbs_bwd2_lds_kernel", "bwd2"), ("lbs_bwd2_kernel", "bwd2"), ("lbs_bwd3_kernel", "bwd3"),
                 ("opt_adam_kernel", "adam"), ("opt_select_kernel", "select"), ("link_kernel", "link")):
        if a in s:
            return b + (s[s.index("<"):] if "<" in s and a in ("opt_tail_kernel", "link_kernel") else "")
    return s[:40]


def main():
    args = sys.argv[1:]
    db, out = args[0], args[1]
    max_gap = float(args[args.index("--max-gap-us") + 1]) if "--max-gap-us" in args else 50.0
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end, grid_x / workgroup_x, workgroup_x, queue_id, lds_size, vgpr_count from kernels order by queue_id, start"))
    kinds = OrderedDict()
    durs = OrderedDict()
    skipped = 0
    for (n0, s0, e0, g0, w0, q0, _, _), (n1, s1, e1, g1, w1, q1, _, _) in zip(rows, rows[1:]):
        if q0 != q1:
            continue
        gap = (s1 - e0) / 1000.0
        if gap > max_gap:
            skipped += 1
            continue
        k = (f"{short(n0)} {int(g0)}x{int(w0)}", f"{short(n1)} {int(g1)}x{int(w1)}")
        kinds.setdefault(k, []).append(gap)
    for n, s, e, g, w, q, lds, vg in rows:
        durs.setdefault(f"{short(n)} {int(g)}x{int(w)}", []).append((e - s) / 1000.0)
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["from", "to", "count", "gap_mean_us", "gap_median_us", "gap_p10_us", "gap_p90_us", "from_kernel_mean_us", "to_kernel_mean_us"])
        for (a, b), g in sorted(kinds.items(), key=lambda kv: -len(kv[1])):
            g = sorted(g)
            q = lambda p: g[min(len(g) - 1, int(p * len(g)))]
            da, dbb = durs[a], durs[b]
            wr.writerow([a, b, len(g), f"{sum(g) / len(g):.3f}", f"{q(0.5):.3f}", f"{q(0.1):.3f}", f"{q(0.9):.3f}",
                         f"{sum(da) / len(da):.3f}", f"{sum(dbb) / len(dbb):.3f}"])
        wr.writerow(["# gaps longer than", f"{max_gap} us (pauses between replays, not boundaries):", skipped])
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
