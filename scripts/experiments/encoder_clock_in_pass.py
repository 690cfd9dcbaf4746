"""Shader clock INSIDE the Stream-K launches of whole encoder passes run back to back (5, 50, 400 passes): the clock ramps with sustained load
(2.18 -> 2.33 -> 2.34 GHz), the pass time settles at 5.72-5.8 ms.  Needs an experiment build:
   hipcc <HIPCC_FLAGS of ihmr_amd/hip.py> -DCONV_STAMPS ihmr_amd/csrc/ihmr_hip.hip -o build/ab/conv_stamps.so
   IHMR_HIP_LIBRARY=build/ab/conv_stamps.so python3 scripts/experiments/encoder_clock_in_pass.py"""
import ctypes as C, os, sys, types, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ihmr_amd import hip
from ihmr_amd.networks import InterHandEncoder
L = hip.lib(); L.ihmr_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
B = 64
enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), torch.zeros(B, 122)).cuda()
img = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
for _ in range(3): enc(img)
for n in (5, 50, 400):
    time.sleep(0.5)
    L.ihmr_debug_conv_stamps(None, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): enc(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    raw = np.zeros(1024 * 10, np.int64); L.ihmr_debug_conv_stamps(raw.ctypes.data, 0); raw = raw.reshape(1024, 10)[:512].astype(np.float64)
    print(f"{n} passes: {dt*1e3:.3f} ms per pass; shader clock inside the Stream-K launches: {raw[:,8].sum()/(raw[:,9].sum()*10):.3f} GHz")
