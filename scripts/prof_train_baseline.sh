#!/bin/bash
# kernel summary of the IHMR-Baseline training step: usage (GPU box) scripts/prof_train_baseline.sh <tag>
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/ptb
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ptb -o ptb -- python3 -m ihmr_amd.run_train_baseline --num_samples 256 --batchSize 64 --total_epoch 3 > gpurun_out/${tag}_train_baseline.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/ptb/ptb_results.db gpurun_out/${tag}_train_baseline_kernel_stats.csv | head -28
tail -3 gpurun_out/${tag}_train_baseline.log
rm -rf gpurun_out/ptb
