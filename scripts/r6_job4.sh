#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job4.log
: > $O
echo "== parity: fused tail, lists, static reuse, convention switches, ragged replays" >> $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or candidate_lists or static_hand or convention_switches or replays_reference or batch64 or translated or single_step or forward_losses" >> $O 2>&1
echo "== latency head vs qmask" >> $O
for r in 1 2; do for lib in head qmask; do
  echo "$lib: $(IHMR_HIP_LIBRARY=$PWD/build/ab/$lib.so timeout 300 python3 scripts/latency.py 64 9 2>/dev/null)" >> $O
done; done
echo "== kernels + bench head vs qmask" >> $O
MODE=both FUSE=7 REPS=2 bash scripts/ab.sh build/ab/head.so build/ab/qmask.so >> $O 2>&1
tail -30 $O
