#!/bin/bash
# kernel summaries of the two rows added after v8: the IHMR-MLP training step and the batched preprocessing
# usage (GPU box): scripts/prof_new_rows.sh <tag>   -> gpurun_out/<tag>_train_kernel_stats.csv, <tag>_preprocess_kernel_stats.csv
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pt gpurun_out/pp
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pt -o pt -- python3 scripts/bench_mlp_train.py 128 50 > gpurun_out/${tag}_train.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/pt/pt_results.db gpurun_out/${tag}_train_kernel_stats.csv | head -30
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o pp -- python3 scripts/bench_preprocess.py 64 > gpurun_out/${tag}_preprocess.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/pp/pp_results.db gpurun_out/${tag}_preprocess_kernel_stats.csv | head -6
tail -8 gpurun_out/${tag}_train.log; tail -2 gpurun_out/${tag}_preprocess.log
rm -rf gpurun_out/pt gpurun_out/pp
