#!/bin/bash
# A/B of library builds at a given launch size: kernel trace of the single-stream bench + the two-stream bench value
# usage: FUSE=8 scripts/abl2.sh lib1.so lib2.so ...   ("product" = the in-tree library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=${FUSE:-8}
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset IHMR_HIP_LIBRARY; else export IHMR_HIP_LIBRARY=$(realpath $lib); fi
  echo "== $lib"
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps $F --warmup $F --streams 1 --fuse $F --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv > /dev/null; python3 -c "import csv; [print(r['kernel'], r['workgroups'], r['calls'], r['avg_us']) for r in csv.DictReader(open('/tmp/ab.csv')) if r['kernel'].startswith(('sdf_', 'opt_', 'lbs_')) and int(r['calls']) > 150]"
  for rep in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-work-counters 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench 2x10:', round(d['value']), 'images/s')"; done
done
rm -rf gpurun_out/ab
