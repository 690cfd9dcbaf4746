"""Duration of the 56 x 56 / 28 x 28 layers of the encoder against the image count N (tiles = N x pixels / 128): a staircase shows how
much of a layer is its last, partly filled generation of workgroups.  usage (GPU box): python scripts/experiments/conv_generations.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ihmr_amd.networks import _Packed, conv_igemm

dev = torch.device("cuda")
layers = (("3x3 64->64 @56", 64, 64, 3, 1, 56, False), ("1x1 256->64 @56", 256, 64, 1, 0, 56, False), ("1x1 64->256 @56 +res", 64, 256, 1, 0, 56, True),
          ("1x1 128->512 @28 +res", 128, 512, 1, 0, 28, True), ("1x1 256->1024 @14 +res", 256, 1024, 1, 0, 14, True))
for name, cin, cout, k, pad, hw, with_res in layers:
    pk = _Packed(torch.randn(cout, cin, k, k, device=dev) * 0.05, torch.zeros(cout, device=dev), stride=1, pad=pad)
    line = []
    for N in (40, 41, 42, 48, 52, 56, 60, 61, 62, 63, 64, 72, 80, 96, 128):
        M = N * hw * hw
        x = torch.randn(M, cin, device=dev); out = torch.empty(M, cout, device=dev)
        res = torch.randn(M, cout, device=dev) if with_res else None
        for _ in range(3):
            conv_igemm(x, pk, N, hw, hw, cin, out=out, ldy=cout, residual=res, ldr=cout if with_res else 0, act=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv_igemm(x, pk, N, hw, hw, cin, out=out, ldy=cout, residual=res, ldr=cout if with_res else 0, act=1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line.append(f"N {N:3d}: {us:6.1f} us = {us / N:5.2f}/img, {2.0 * M * cin * k * k * cout / us / 1e6:5.1f} TF")
    print(name + "\n   " + "\n   ".join(line))
