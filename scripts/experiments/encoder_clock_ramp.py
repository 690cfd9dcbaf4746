"""Encoder pass time against the length of the measurement: the shader clock ramps up under sustained load (2.0-2.1 GHz in the first
milliseconds of work, ~2.35 GHz after a few hundred: scripts/experiments/conv_stamps.py <launches>, mfma_sustained.hip), so a
10-pass measurement reads slower kernels than a long run does.  usage (GPU box): python scripts/experiments/encoder_clock_ramp.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ihmr_amd.networks import InterHandEncoder

B = 64
enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), torch.zeros(B, 122)).cuda()
img = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
for _ in range(3):
    enc(img)
torch.cuda.synchronize()
for n in (10, 10, 50, 200, 800, 10):
    time.sleep(0.5)                                   # let the clock fall back
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        enc(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{n:4d} passes back to back: {dt * 1e3:.3f} ms per pass = {8.2e9 * B / dt / 1e12:.1f} TFLOP/s = {8.2e9 * B / dt / 1e12 / 157.3 * 100:.1f} % of the 2.4 GHz peak")
