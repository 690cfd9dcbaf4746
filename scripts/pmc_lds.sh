#!/bin/bash
# LDS / memory-instruction counters of the IHMR-OPT kernels at FUSE batches per launch: usage FUSE=8 scripts/pmc_lds.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F=${FUSE:-8}
rm -rf gpurun_out/pl
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pl -o p -- python3 bench.py --steps $F --warmup $F --streams 1 --fuse $F --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        if not k.startswith(("sdf_", "opt_", "lbs_")): continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"][1]):
    m = {n: v[1] / v[0] for n, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{k:40s} x{c['SQ_WAVE_CYCLES'][0]:5d} wait/wave {m['SQ_WAIT_ANY']/wc:.2f} waitLDS/wave {m['SQ_WAIT_INST_LDS']/wc:.3f} LDS instr {m['SQ_INSTS_LDS']/1e6:.2f} M bankconf/ldsactive {m['SQ_LDS_BANK_CONFLICT']/max(m['SQ_LDS_IDX_ACTIVE'],1):.2f} ldsactive/wave {m['SQ_LDS_IDX_ACTIVE']/wc:.3f} vmem rd {m['SQ_INSTS_VMEM_RD']/1e6:.2f} M wr {m['SQ_INSTS_VMEM_WR']/1e6:.2f} M")
PY
rm -rf gpurun_out/pl
