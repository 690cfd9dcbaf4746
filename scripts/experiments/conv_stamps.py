"""Where a Stream-K worker's time goes (experiment build):
   hipcc <HIPCC_FLAGS> -DCONV_STAMPS ihmr_amd/csrc/ihmr_hip.hip -o build/ab/conv_stamps.so
   IHMR_HIP_LIBRARY=build/ab/conv_stamps.so python3 scripts/experiments/conv_stamps.py
Runs single Stream-K layers of the ResNet-50 shapes at batch 64 and prints the mean shader-clock time thread 0 of a worker spends per K step in:
load issue, LDS reads + MFMA issue, wait for the older tile + LDS stores, barrier; and per segment in prologue and epilogue."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ihmr_amd import hip
from ihmr_amd.networks import _Packed, conv_igemm

L = hip.lib()
L.ihmr_debug_conv_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda")
for name, N, H, cin, cout, k in (("3x3 256 @14", 64, 14, 256, 256, 3), ("3x3 128 @28", 64, 28, 128, 128, 3), ("3x3 512 @7", 64, 7, 512, 512, 3),
                                 ("1x1 1024->256 @14", 64, 14, 1024, 256, 1), ("1x1 2048->512 @7", 64, 7, 2048, 512, 1)):
    pk = _Packed(torch.randn(cout, cin, k, k, device=dev) * 0.02, torch.zeros(cout, device=dev), pad=k // 2)
    x = torch.randn(N * H * H, cin, device=dev)
    for _ in range(3):
        conv_igemm(x, pk, N, H, H, cin, act=1)
    L.ihmr_debug_conv_stamps(None, 1)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 10      # launches per measurement (the clock is reported for all of them together)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        conv_igemm(x, pk, N, H, H, cin, act=1)
    e1.record(); torch.cuda.synchronize()
    raw = np.zeros(1024 * 10, np.int64)
    L.ihmr_debug_conv_stamps(raw.ctypes.data, 0)
    raw = raw.reshape(1024, 10)[:512].astype(np.float64)
    ghz = raw[:, 8].sum() / (raw[:, 9].sum() * 10.0)          # shader clocks per ns of the 100 MHz wall clock
    steps, segs = raw[:, 6].mean() / R, raw[:, 7].mean() / R
    per_step = raw[:, 1:5].sum(0) / raw[:, 6].sum() / 2403.0
    per_seg = raw[:, [0, 5]].sum(0) / raw[:, 7].sum() / 2403.0
    tot = raw[:, :6].sum(1) / R / 2403.0
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / R:.1f} us per call (main + fix-up); worker: {steps:.1f} K steps in {segs:.2f} segments, {tot.mean():.1f} us (max {tot.max():.1f}); "
          f"per K step: load issue {per_step[0]:.3f}, LDS reads + MFMA issue {per_step[1]:.3f}, wait + LDS stores {per_step[2]:.3f}, barrier {per_step[3]:.3f} = {per_step.sum():.3f} us; "
          f"per segment: prologue {per_seg[0]:.2f}, epilogue {per_seg[1]:.2f} us (all at 2.403 shader clocks per ns); shader clock while the workers ran: {ghz:.3f} GHz")
