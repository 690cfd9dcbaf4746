#!/bin/bash
# per-kernel means of library builds at FUSE batches per launch (single stream, kernel trace only): usage FUSE=8 scripts/ab_kernels.sh lib1.so lib2.so ... ("product" = the in-tree library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=${FUSE:-8}
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset IHMR_HIP_LIBRARY; else export IHMR_HIP_LIBRARY=$(realpath $lib); fi
  rm -rf gpurun_out/ab; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps $F --warmup $F --streams 1 --fuse $F --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv > /dev/null
  echo "== $lib: $(python3 -c "import csv; print('; '.join(r['kernel'][:24] + ' ' + r['avg_us'] for r in csv.DictReader(open('/tmp/ab.csv')) if r['kernel'].startswith(('sdf_', 'opt_tail')) and int(r['calls']) > 150))")"
done
rm -rf gpurun_out/ab
