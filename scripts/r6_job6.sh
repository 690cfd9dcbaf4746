#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job6.log
: > $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "in_numbers or fused_tail or candidate_lists or static_hand or batch64 or batch512 or fused_batches or dense_grid or bwd2_forms or 200_iterations or trajectory_matches" >> $O 2>&1
grep "in numbers" $O > gpurun_out/r6_translated_numbers.txt
echo "== latency qmask vs fold" >> $O
for r in 1 2; do for lib in qmask fold; do
  echo "$lib: $(IHMR_HIP_LIBRARY=$PWD/build/ab/$lib.so timeout 300 python3 scripts/latency.py 64 9 2>/dev/null)" >> $O
done; done
echo "== kernels + bench qmask vs fold" >> $O
MODE=both FUSE=7 REPS=2 bash scripts/ab.sh build/ab/qmask.so build/ab/fold.so >> $O 2>&1
tail -32 $O
