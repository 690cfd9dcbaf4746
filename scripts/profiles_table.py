#!/usr/bin/env python3
"""The rows of profiles/README.md's table for one profile series, generated from the committed csv files (so the README cannot
drift from them: tests/test_host_cpu.py regenerates the rows and looks for them in the README).
usage: python3 scripts/profiles_table.py r5_v1        -> markdown rows on stdout"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("sdf_dist_kernel", "sdf_prep_kernel", "opt_tail_kernel<true, true>", "opt_tail_kernel<true, false>", "opt_tail_kernel<false, false>",
           "lbs_skin_kernel<true, 0", "lbs_skin_kernel<true, 2")
LABEL = {"f7": "7 batches = 448 samples: the driver's `--steps 20` run (three sequences of 7 + 7 + 6)", "f8": "8 batches = 512 samples",
         "f1": "one batch of 64 (the latency case)"}


def rows_of(path):
    try:
        with open(path, newline="") as fh:
            return list(csv.DictReader(fh))
    except OSError:
        return []


def main_row(prefix, size):
    """One table row: per kernel the mean launch duration, vector instructions and HBM bytes of its most-launched shape."""
    ks, sq, tr = (rows_of(f"{prefix}_{size}_{s}.csv") for s in ("kernel_stats", "pmc_sq", "pmc_traffic"))
    if not ks:
        return None
    top = lambda rows, name, key: max((r for r in rows if r["kernel"].startswith(name)), key=lambda r: int(r.get(key) or 0), default=None)
    parts = []
    for k in KERNELS:
        a = top(ks, k, "calls")
        if a is None:
            continue
        label = {"lbs_skin_kernel<true, 0": "lbs_skin_kernel (both blends)", "lbs_skin_kernel<true, 2": "lbs_skin_kernel (pose offsets kept)"}.get(k, k.replace(", ", ","))
        s = f"`{label}` {float(a['avg_us']):.1f} µs"
        b, c = top(sq, k, "launches"), top(tr, k, "launches")
        extra = []
        if b and b.get("SQ_INSTS_VALU_per_launch"):
            extra.append(f"{float(b['SQ_INSTS_VALU_per_launch']) / 1e6:.1f} M VALU instructions")
        if c:
            extra.append(f"{float(c['hbm_bytes_per_launch']) / 1e6:.1f} MB HBM")
        parts.append(s + (" (" + ", ".join(extra) + ")" if extra else ""))
    name = os.path.basename(prefix)
    return f"| `{name}_{size}_*` | {LABEL[size]} | " + "; ".join(parts) + " |"


def table(series):
    prefix = os.path.join(ROOT, "profiles", series)
    return [r for r in (main_row(prefix, s) for s in ("f7", "f8", "f1")) if r]


if __name__ == "__main__":
    print("\n".join(table(sys.argv[1])))
