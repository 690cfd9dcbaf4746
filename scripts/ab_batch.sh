#!/bin/bash
# A/B of library builds at a launch batch: usage ab_batch.sh <B> lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B=$1; shift
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib); echo "== $lib (B=$B)"
  rm -rf gpurun_out/pb; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pb -o pb -- python3 bench.py --steps 2 --warmup 1 --streams 1 --fuse 1 --batch $B --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-135
  python3 scripts/rocprof_summary.py gpurun_out/pb/pb_results.db /tmp/pb.csv | grep -E "${ABK:-skin}"
  timeout 200 python3 bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-140
done
rm -rf gpurun_out/pb
