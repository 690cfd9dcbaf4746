#!/bin/bash
cd $GRAFT_REPO_ROOT
# ablation of the conv K-step: bit0 no global loads, bit1 no LDS operand reads, bit2 no LDS stores, bit3 no barriers, bit4 no epilogue (results are garbage, timing only)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DIHMR_CONV_EXPERIMENT ihmr_amd/csrc/ihmr_hip.hip -o scripts/lib_convexp.so || exit 1
cp ihmr_amd/libihmr_hip.so /tmp/keep.so; cp scripts/lib_convexp.so ihmr_amd/libihmr_hip.so
for m in 0 1 2 4 8 15 16 31; do echo -n "exp_mask=$m: "; IHMR_CONV_EXP=$m timeout 200 python scripts/bench_models.py baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['encoder_ms_per_batch'],2), 'ms', round(d['encoder_tflops'],1), 'TF')"; done
cp /tmp/keep.so ihmr_amd/libihmr_hip.so
rm -f scripts/lib_convexp.so
