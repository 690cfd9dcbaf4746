"""EXPERIMENT (round 4): per-stage candidate-list slack.  Wall time of each opt_default stage (50 iterations, 8 x 64 samples per launch, graphs)
for several values of the slack, with a build that exposes ihmr_debug_set_list_slack.  usage: IHMR_HIP_LIBRARY=build/ab/lib_slack.so python3 scripts/experiments/slack_sweep.py"""
import ctypes as C, os, sys, time, types
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
G, B = 8, 64
L = hip.lib(); L.ihmr_debug_set_list_slack.argtypes = [C.c_int, C.c_float]
def run(slacks):
    for k, v in enumerate(slacks): L.ihmr_debug_set_list_slack(k, v)
    o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                              cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                              save_mid_freq=10, optimizer="adam", opt_epoch=49, fuse_batches=G)
    m = OptimizeModel(o)
    fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i) for i in range(G)]
    inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
    m.set_input(inp); m.init_optimize(); m.optimize(); torch.cuda.synchronize()
    out = []
    for rep in range(3):
        m.set_input(inp); m.init_optimize(); torch.cuda.synchronize()
        ts = []
        for st in m.strategy:
            t0 = time.perf_counter(); m.run_stage(st); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out.append(ts)
    res = m.get_pred_result()
    return np.min(np.array(out), axis=0), res["pred_pose_params"].copy()
base, ref = run([0, 0, 0, 0])
print("default 0.04:", np.round(base, 3), "sum", round(base.sum(), 3))
for s in (0.01, 0.02, 0.03, 0.06):
    t, r = run([s, s, s, s])
    print(f"slack {s}:", np.round(t, 3), "sum", round(t.sum(), 3), "bit-identical result:", bool(np.array_equal(r, ref)))
