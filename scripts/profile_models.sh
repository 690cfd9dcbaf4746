#!/bin/bash
# rocprofv3 summaries of the secondary workloads (IHMR-Baseline B=64, IHMR-MLP B=128): kernel trace + SQ counters
# usage: scripts/profile_models.sh <tag>  -> gpurun_out/<tag>_{baseline,mlp}_kernel_stats.csv, <tag>_baseline_pmc_sq.csv
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in baseline mlp; do
  rm -rf gpurun_out/pm
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pm -o pm -- python3 scripts/bench_models.py $w > gpurun_out/${tag}_${w}.log 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/pm/pm_results.db gpurun_out/${tag}_${w}_kernel_stats.csv | head -8
  tail -1 gpurun_out/${tag}_${w}.log | cut -c1-300
done
rm -rf gpurun_out/pm
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pm -o pm -- python3 scripts/bench_models.py baseline > /dev/null 2>&1
python3 scripts/pmc_sq_summary.py gpurun_out/pm gpurun_out/${tag}_baseline_pmc_sq.csv | cut -c1-200 | head -8
rm -rf gpurun_out/pm
