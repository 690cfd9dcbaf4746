#!/bin/bash
# SQ counters of the secondary workloads, grouped by kernel AND grid: usage pmc_models.sh baseline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pm; timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pm -o pm -- python3 scripts/bench_models.py $1 > /tmp/pm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size"], r["Workgroup_Size"])
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for k, c in acc.items():
    m = {n: v[1] / v[0] for n, v in c.items()}
    rows.append((m["SQ_BUSY_CYCLES"] * c["SQ_BUSY_CYCLES"][0], k, c["SQ_BUSY_CYCLES"][0], m))
for tot, k, n, m in sorted(rows, reverse=True)[:16]:
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{k[0]:30s} grid={k[1]:>8s} wg={k[2]:>4s} x{n:3d}  mfma_busy/busy={m['SQ_VALU_MFMA_BUSY_CYCLES']/m['SQ_BUSY_CYCLES']:.2f}  wait/wave={m['SQ_WAIT_ANY']/wc:.2f} waitLDS/wave={m['SQ_WAIT_INST_LDS']/wc:.2f} ldsActive/wave={m['SQ_ACTIVE_INST_LDS']/wc:.2f} bankconf/ldsact={m['SQ_LDS_BANK_CONFLICT']/max(m['SQ_ACTIVE_INST_LDS'],1):.2f} busy={m['SQ_BUSY_CYCLES']:.0f} mfma={m['SQ_INSTS_MFMA']:.0f}")
PY
rm -rf gpurun_out/pm
