#!/bin/bash
# Profile of one of the secondary BASELINE.json configs on the GPU box: kernel trace (+stats), the two HBM-traffic PMC passes and one
# SQ pass of the SAME command, plus a meta record with the source hash of the profiled library (bench.py quotes a profile only when
# that hash equals the hash of the library it loads).
# usage: scripts/profile_config.sh <tag> <baseline|mlp>   -> gpurun_out/<tag>_<config>_{kernel_stats,pmc_traffic,pmc_sq}.csv, _meta.json
tag=${1:-rX}_${2:-baseline}
cfg=${2:-baseline}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CMD="python3 bench.py --config $cfg --no-cpu-baseline"
mkdir -p gpurun_out
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_s
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- $CMD > gpurun_out/${tag}_kt.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/kt/kt_results.db gpurun_out/${tag}_kernel_stats.csv | head -12
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- $CMD > gpurun_out/${tag}_pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- $CMD > gpurun_out/${tag}_pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_s -o s -- $CMD > gpurun_out/${tag}_pmc_s.log 2>&1
python3 scripts/pmc_sq_summary.py gpurun_out/pmc_s gpurun_out/${tag}_pmc_sq.csv | cut -c1-260 | head -8
python3 scripts/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/${tag}_pmc_traffic.csv | head -12
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_s
python3 - "$tag" "$cfg" "$CMD" <<'PY'
import json, sys, time
sys.path.insert(0, ".")
from ihmr_amd import hip
tag, cfg, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
json.dump(dict(srchash=hip.loaded_source_hash(), config=cfg, batches_per_launch=None, command=cmd, unix_time=int(time.time())),
          open(f"gpurun_out/{tag}_meta.json", "w"))
PY
tail -1 gpurun_out/${tag}_kt.log | cut -c1-600
