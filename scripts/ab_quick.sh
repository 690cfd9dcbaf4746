#!/bin/bash
# quick A/B of library builds (kernel trace only): usage ab_quick.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  echo "== $lib"
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-135
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv | grep -E "${ABK:-sdf_dist|sdf_prep}"
done
mv gpurun_out/ab/ab_results.db /tmp/ab_last.db; rm -rf gpurun_out/ab
