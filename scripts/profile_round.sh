#!/bin/bash
# Round profile on the GPU box: kernel trace (+stats) and the two HBM-traffic PMC passes of the SAME bench command.
# usage: FUSE=<batches per launch> STEPS=<n> scripts/profile_round.sh <tag>   -> gpurun_out/<tag>_f<FUSE>_kernel_stats.csv, _pmc_traffic.csv, _pmc_sq.csv
# (rocprofv3 gets the program itself after `--`; counters run separately from the trace, one TCC counter per pass.)
tag=${1:-rX}_f${FUSE:-8}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CMD="python3 bench.py --steps ${STEPS:-${FUSE:-8}} --warmup ${FUSE:-8} --streams 1 --fuse ${FUSE:-8} --no-cpu-baseline --no-extras --no-work-counters"
mkdir -p gpurun_out
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt -- $CMD > gpurun_out/${tag}_kt.log 2>&1
python3 scripts/rocprof_summary.py gpurun_out/kt/kt_results.db gpurun_out/${tag}_kernel_stats.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- $CMD > gpurun_out/${tag}_pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- $CMD > gpurun_out/${tag}_pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_s -o s -- $CMD > gpurun_out/${tag}_pmc_s.log 2>&1
python3 scripts/pmc_sq_summary.py gpurun_out/pmc_s gpurun_out/${tag}_pmc_sq.csv | cut -c1-260 | head -12
find gpurun_out/pmc_f -name "*counter_collection.csv" | head -1 | xargs head -3
python3 scripts/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/${tag}_pmc_traffic.csv
rm -rf gpurun_out/kt gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_s
# which code was profiled: bench.py quotes a profile only when this hash equals the hash of the library it loads
python3 - "$tag" "${FUSE:-8}" "$CMD" <<'PY'
import json, sys, time
sys.path.insert(0, ".")
from ihmr_amd import hip
tag, fuse, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
json.dump(dict(srchash=hip.loaded_source_hash(), config="opt", batches_per_launch=fuse, command=cmd, unix_time=int(time.time())),
          open(f"gpurun_out/{tag}_meta.json", "w"))
PY
tail -1 gpurun_out/${tag}_kt.log | cut -c1-400
