#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job2.log
: > $O
IHMR_HIP_LIBRARY=$PWD/build/handlog.so timeout 600 python3 scripts/experiments/hand_work_log.py 64 >> $O 2>&1
# timeline again, keeping the records: per-launch spans of the distance kernel by stage
IHMR_HIP_LIBRARY=$PWD/build/timeline.so IHMR_TIMELINE_OUT=$PWD/gpurun_out/r6_tl.npy timeout 300 python3 bench.py --steps 1 --warmup 1 --streams 1 --fuse 1 --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>> $O
ls -la gpurun_out/r6_tl.npy >> $O
tail -3 $O
