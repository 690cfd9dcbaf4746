cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/kmt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kmt -o k -- python3 scripts/bench_mlp_train.py 128 50 > gpurun_out/mlp_train_kt.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/kmt/k_results.db gpurun_out/mlp_train_kernel_stats.csv | head -30
tail -6 gpurun_out/mlp_train_kt.log
rm -rf gpurun_out/kmt
