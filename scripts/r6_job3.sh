#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job3.log
: > $O
echo "== tail stamps, one batch of 64" >> $O
IHMR_HIP_LIBRARY=$PWD/build/tailstamps.so timeout 600 python3 scripts/tail_stamps.py 1 >> $O 2>&1
echo "== sdf stamps (list search), one batch" >> $O
IHMR_HIP_LIBRARY=$PWD/build/stamps1.so timeout 600 python3 scripts/sdf_stamps.py 1 >> $O 2>&1
echo "== sdf stamps (full search), one batch" >> $O
IHMR_HIP_LIBRARY=$PWD/build/stamps2.so timeout 600 python3 scripts/sdf_stamps.py 1 >> $O 2>&1
tail -3 $O
