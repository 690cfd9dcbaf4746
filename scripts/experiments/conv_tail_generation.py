"""How much of a short-K 1 x 1 layer is the last, nearly empty generation of workgroups?  Times ihmr_conv_igemm on 1 x 1 layers of
the c3 shape (Cin -> 4 Cin, residual + ReLU) with M chosen so that the 128 x 128 tiles number exactly 768 (three per CU), 784 (the
14 x 14 layers of a 64-image batch) and 800.  usage (GPU box): python scripts/experiments/conv_tail_generation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ihmr_amd.networks import _Packed, conv_igemm

dev = torch.device("cuda")
for cin, cout, ms in ((256, 1024, tuple(4096 * k for k in range(1, 9)) + (96 * 128 + 128, 98 * 128, 100 * 128)), (128, 512, (384 * 128, 392 * 128, 400 * 128)),
                      (512, 2048, (16 * 128, 24 * 128, 3136, 32 * 128)), (512, 1024, (96 * 128, 98 * 128)), (256, 128, (1536 * 128, 1568 * 128))):
    pk = _Packed(torch.randn(cout, cin, 1, 1, device=dev) * 0.05, torch.zeros(cout, device=dev))
    for M in ms:
        x = torch.randn(M, cin, device=dev); res = torch.randn(M, cout, device=dev); out = torch.empty(M, cout, device=dev)
        for _ in range(3):
            conv_igemm(x, pk, M, 1, 1, cin, out=out, ldy=cout, residual=res, ldr=cout, act=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv_igemm(x, pk, M, 1, 1, cin, out=out, ldy=cout, residual=res, ldr=cout, act=1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tiles = (M + 127) // 128 * (cout // 128)
        print(f"Cin {cin:4d} Cout {cout:4d} M {M:6d}: {tiles:5d} tiles = {tiles / 256:5.2f} per CU  {us:7.1f} us  {2.0 * M * cin * cout / us / 1e6:6.1f} TFLOP/s")
