#!/bin/bash
# Kernel durations of the encoder per LAYER (a convolution launch + the reduce / fix-up launch that finishes it), in call order, from a
# rocprofv3 kernel trace of scripts/encoder_pass.py (6 passes of 64 images; the median over the passes is printed).
# usage (on the GPU box): [IHMR_HIP_LIBRARY=...] scripts/prof_encoder_layers.sh <tag>   -> gpurun_out/<tag>_encoder_layers.txt
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/ktl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -o l -- python3 scripts/encoder_pass.py > gpurun_out/${tag}_ktl.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/ktl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"] or "pool" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
P = 6
n = len(rows) // P
assert n * P == len(rows), (len(rows), P)
layers = []      # (label, [durations per pass])
for i in range(n):
    k = rows[i]["Kernel_Name"].split("(")[0].replace("void ", "")
    wg = [int(rows[i][f"Workgroup_Size_{a}"]) for a in "XYZ"]
    grid = tuple(int(rows[i][f"Grid_Size_{a}"]) // w for a, w in zip("XYZ", wg))
    ds = [(int(rows[p * n + i]["End_Timestamp"]) - int(rows[p * n + i]["Start_Timestamp"])) / 1e3 for p in range(P)]
    med = sorted(ds)[P // 2]
    if ("reduce" in k or "fixup" in k) and layers:
        layers[-1][1] += med; layers[-1][0] += f" + {k.split('<')[0]} {med:.1f}"
    else:
        layers.append([f"{k} grid {grid} {med:.1f}", med])
out = open(f"gpurun_out/{tag}_encoder_layers.txt", "w")
tot = 0.0
for i, (label, t) in enumerate(layers):
    tot += t; print(f"{i:3d} {t:8.1f} us  {label}", file=out)
print(f"total {tot:.1f} us per pass", file=out)
out.close()
print(open(f"gpurun_out/{tag}_encoder_layers.txt").read())
PY
rm -rf gpurun_out/ktl
