#!/bin/bash
# Baseline training step per library build: usage ab_train.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib); echo -n "$lib run $r: "
  timeout 300 python3 -m ihmr_amd.run_train_baseline --num_samples 256 --batchSize 64 --total_epoch 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step')"
done; done
