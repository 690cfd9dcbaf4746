#!/bin/bash
# kernel summary at a given launch batch (single stream): usage prof_batch.sh <batch>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pb; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pb -o pb -- python3 bench.py --steps 2 --warmup 1 --streams 1 --fuse 1 --batch $1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-135
python3 scripts/rocprof_summary.py gpurun_out/pb/pb_results.db /tmp/pb.csv | head -${2:-11}
rm -rf gpurun_out/pb
