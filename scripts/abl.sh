#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  echo "== $lib"
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps 8 --warmup 8 --streams 1 --fuse 8 --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv | grep -E "sdf_prep|sdf_dist|opt_sample|lbs_bwd2" | cut -d, -f1,2,4,6
done
rm -rf gpurun_out/ab
