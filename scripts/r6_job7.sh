#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job7.log
: > $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "in_numbers or fused_tail or candidate_lists or static_hand or batch64 or batch512 or fused_batches or skin_keeps or 200_iterations or trajectory_matches or mixed_parameter or single_step or replays_reference" >> $O 2>&1
grep "in numbers" $O > gpurun_out/r6_translated_numbers.txt
echo "== latency noshape vs shape" >> $O
for r in 1 2; do for lib in noshape shape; do
  echo "$lib: $(IHMR_HIP_LIBRARY=$PWD/build/ab/$lib.so timeout 300 python3 scripts/latency.py 64 9 2>/dev/null)" >> $O
done; done
echo "== kernels + bench noshape vs shape" >> $O
MODE=both FUSE=7 REPS=3 bash scripts/ab.sh build/ab/noshape.so build/ab/shape.so >> $O 2>&1
tail -28 $O
