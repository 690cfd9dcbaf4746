import ctypes as C, os, sys, types
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
G, B = 8, 64
o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                          cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                          save_mid_freq=10, optimizer="adam", opt_epoch=49, fuse_batches=G)
m = OptimizeModel(o)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i) for i in range(G)]
inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
m.set_input(inp); m.init_optimize(); m.optimize(); torch.cuda.synchronize()
L = hip.lib(); L.ihmr_debug_stamps.argtypes = [C.c_void_p, C.c_int]
m.set_input(inp); m.init_optimize(); m.run_stage(m.strategy[0])
L.ihmr_debug_stamps(None, 1)
m.run_stage(m.strategy[1]); torch.cuda.synchronize()
raw = np.zeros(4096 * 4 * 12 + 4096 * 8, np.int64); L.ihmr_debug_stamps(raw.ctypes.data, 0)
prep = raw[4096 * 4 * 12:].reshape(-1, 8); prep = prep[prep[:, 7] > 0]
ray = prep[:, 4] / prep[:, 7] / 2400.0; P = prep[:, 6] / prep[:, 7]
print("hands", len(prep), "P mean", P.mean(), "p90", np.percentile(P, 90), "max", P.max(), "frac > 4096", (P > 4096).mean())
order = np.argsort(ray)
for q in (0.1, 0.5, 0.9, 0.99, 1.0):
    i = order[min(int(q * len(order)), len(order) - 1)]
    print(f"ray-phase time quantile {q}: {ray[i]:.2f} us, pairs {P[i]:.0f}")
print("corr(ray time, P)", np.corrcoef(ray, P)[0, 1])
c = np.polyfit(P, ray, 1); print("fit: ray_us =", c[0] * 1000, "us per 1000 pairs +", c[1])
