#!/usr/bin/env python3
"""Static vector-instruction mix of the hot kernels from the device assembly (scripts/isa.sh writes it) and the issue cycles per
VALU instruction that mix implies, with the per-class rates measured by scripts/microbench_issue (MI355X: every class below
occupies its SIMD's issue port for 2.4-2.5 cycles per wave64 instruction; shifts / 3-operand logic / packed fp32: 4.2; compare /
select: 3.1; rcp / sqrt: 8.2):
   scripts/isa.sh /tmp/isa/ihmr.s && python3 scripts/isa_mix.py /tmp/isa/ihmr.s
A STATIC count (every instruction once, loops not weighted): the mix-weighted average is bench.py's `VALU_ISSUE_CYCLES` (3.0)."""
import re
import sys
from collections import Counter

# shader cycles one wave64 instruction occupies its SIMD's issue port with four waves per SIMD (scripts/microbench_issue on MI355X,
# profiles/r5_issue_rates.txt): fp32 fma / mul / add and 2-operand integer add 2.4-2.5; shifts, 3-operand logic, packed fp32 4.2;
# compare / select 3.1; rcp / sqrt 8.2.  (The architecture guide's "v_fma_f32: 2 cycles per wave64" + the loop's scalar overhead;
# round 4's bench line priced EVERY vector instruction at 4.)
CYCLES = dict(simple=2.45, slow=4.2, packed_f32=4.2, select=3.1, trans=8.2)
TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
SLOW = ("v_lshl", "v_lshr", "v_ashr", "v_and_or", "v_or3", "v_add3", "v_bfe", "v_bfi", "v_mad_u", "v_mad_i", "v_mul_lo", "v_mul_hi", "v_perm",
        "v_alignbit", "v_mbcnt", "v_xad", "v_lerp", "v_cvt_pk", "v_mad_u64")
KERNELS = ("sdf_dist_kernelILb0", "sdf_prep_kernelILb0ELi512", "opt_tail_kernelILb1ELb1", "opt_tail_kernelILb1ELb0", "opt_tail_kernelILb0ELb0")


def classify(op):
    if op.startswith("v_pk_"):
        return "packed_f32"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "select"
    if op.startswith(SLOW):
        return "slow"
    return "simple"


def main(path):
    txt = open(path).read()
    for kern in KERNELS:
        m = re.search(r"^(_Z\d+" + re.escape(kern) + r"[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M)
        if not m:
            continue
        c, other = Counter(), Counter()
        for line in m.group(2).splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((";", ".", "s_")) or t[0].endswith(":"):
                continue
            op = t[0]
            if op.startswith("v_") and not op.startswith(("v_mfma", "v_readlane", "v_readfirstlane", "v_writelane")):
                c[classify(op)] += 1
            else:
                other[op.split("_")[0]] += 1
        n = sum(c.values())
        avg = sum(CYCLES[k] * v for k, v in c.items()) / max(n, 1)
        print(f"{kern:28s} {n:6d} VALU instructions (static): " + ", ".join(f"{k} {100 * v / n:.1f} %" for k, v in c.most_common()) +
              f"  => {avg:.2f} issue cycles per VALU instruction;  memory / LDS / other: {dict(other.most_common(4))}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/ihmr.s")
