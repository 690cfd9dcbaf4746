import ctypes as C, os, sys, types, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ihmr_amd import hip, mano
from ihmr_amd.assets import synthetic_mano
L = hip.lib()
m = mano.MANO(synthetic_mano(True)).cuda()
for N in (128, 256, 1024, 2048):
    g = torch.Generator().manual_seed(0)
    o, p, b = torch.randn(N, 3, generator=g).cuda(), (torch.randn(N, 45, generator=g) * 0.3).cuda(), torch.randn(N, 10, generator=g).cuda()
    for force in (1, 0):
        L.ihmr_debug_force_lbs_skin_vector(force)
        for _ in range(5): m(global_orient=o, hand_pose=p, betas=b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): m(global_orient=o, hand_pose=p, betas=b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(N, "vector" if force else "mfma", round(dt * 1e6, 1), "us per forward (skeleton + skin launches)")
L.ihmr_debug_force_lbs_skin_vector(0)
