#!/bin/bash
# A/B of library builds: usage ab_dist.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  echo "== $lib"
  cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-135
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv | grep -E "sdf_dist|sdf_prep" 
  timeout 200 python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras --streams 4 2>&1 | tail -1 | cut -c95-135
done
