#!/bin/bash
# kernel summary of the secondary workloads: usage prof_models.sh baseline|mlp
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pm; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pm -o pm -- python3 scripts/bench_models.py $1 2>&1 | tail -1 | cut -c1-400
python3 scripts/rocprof_summary.py gpurun_out/pm/pm_results.db /tmp/pm.csv | head -${2:-14}
rm -rf gpurun_out/pm
