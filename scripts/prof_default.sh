#!/bin/bash
# kernel summary of the DEFAULT bench configuration (4 streams x 2 fused batches): durations under contention
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pd; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pd -o pd -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c95-135
python3 scripts/rocprof_summary.py gpurun_out/pd/pd_results.db /tmp/pd.csv | head -11
rm -rf gpurun_out/pd
