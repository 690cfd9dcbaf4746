#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  echo "== $lib"
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps 8 --warmup 8 --streams 1 --fuse 8 --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv > /dev/null; python3 -c "import csv; [print(r['kernel'], r['workgroups'], r['calls'], r['avg_us']) for r in csv.DictReader(open('/tmp/ab.csv')) if r['kernel'].startswith(('sdf_', 'opt_sample', 'lbs_', 'opt_adam_skel'))]"
done
rm -rf gpurun_out/ab
