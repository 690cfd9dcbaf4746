#!/bin/bash
# Device-only assembly of the library + per-kernel resource usage:  scripts/isa.sh [out.s] [kernel-name regex] [extra hipcc flags...]
# (registers, spills and scratch per kernel; the kernel's own assembly goes to <out>.<n>.s for reading waitcnt placement)
out=${1:-/tmp/isa/ihmr.s}; pat=${2:-sdf_dist_kernelILb0|sdf_prep_kernelILb0ELi512|opt_tail_kernelILb1ELb1}; [ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
mkdir -p "$(dirname "$out")"
cd "$(dirname "$0")/../ihmr_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -S --cuda-device-only "$@" -o "$out" ihmr_hip.hip 2>&1 | grep -v "hip-link"
awk -v pat="$pat" '/\.name:/{n=$2} /\.private_segment_fixed_size|\.sgpr_spill_count|\.vgpr_count|\.vgpr_spill_count|\.group_segment_fixed_size/{a[$1]=$2}
     /\.wavefront_size/{ if (n ~ pat) printf "%-90s vgpr %s spill %s sgpr-spill %s scratch %s lds %s\n", substr(n,1,90), a[".vgpr_count:"], a[".vgpr_spill_count:"], a[".sgpr_spill_count:"], a[".private_segment_fixed_size:"], a[".group_segment_fixed_size:"] }' "$out"
