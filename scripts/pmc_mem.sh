#!/bin/bash
# vector-memory path counters of the IHMR-OPT kernels at FUSE batches per launch (TA / TCP / TCC busy + hit rates, VMEM latency): usage FUSE=8 scripts/pmc_mem.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=${FUSE:-8}
CMD="python3 bench.py --steps $F --warmup $F --streams 1 --fuse $F --no-cpu-baseline --no-extras --no-work-counters"
rocprofv3 -L 2>/dev/null | grep -oE "Name[^A-Za-z]*(TA_|TCP_|TCC_|TD_|GRBM_)[A-Za-z0-9_]*|^\s*(TA_|TCP_|TCC_|TD_|GRBM_)[A-Za-z0-9_]*" | grep -oE "(TA_|TCP_|TCC_|TD_|GRBM_)[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/mem_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "TA_TA_BUSY TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TCC_BUSY_avr TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/pm
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pm -o p -- $CMD > gpurun_out/pm.log 2>&1
  python3 - "$set" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        if not k.startswith(("sdf_", "opt_tail", "lbs_skin")): continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
if not acc: print("no data for", sys.argv[1]); print(open("gpurun_out/pm.log").read()[-400:])
for k, c in sorted(acc.items()):
    print(f"{k:36s} " + " ".join(f"{n}={v[1] / v[0]:.4g}" for n, v in sorted(c.items())))
PY
done
rm -rf gpurun_out/pm
