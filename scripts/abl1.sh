#!/bin/bash
# A/B of experiment libraries on ONE batch of 64 per launch (the latency case): scripts/abl1.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  echo "== $lib"
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps 2 --warmup 1 --streams 1 --fuse 1 --no-cpu-baseline --no-extras --no-work-counters > gpurun_out/ab.json 2>/dev/null
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv > /dev/null; python3 -c "import csv; [print(r['kernel'], r['workgroups'], r['calls'], r['avg_us']) for r in csv.DictReader(open('/tmp/ab.csv')) if r['kernel'].startswith(('sdf_', 'opt_sample', 'lbs_', 'opt_adam_skel'))]"
  python3 -c "import json; print('images/s', round(json.load(open('gpurun_out/ab.json'))['value']))"
done
rm -rf gpurun_out/ab gpurun_out/ab.json
