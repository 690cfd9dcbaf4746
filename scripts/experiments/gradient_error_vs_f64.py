#!/usr/bin/env python3
"""Where does the HIP path's extra distance from the float64 trajectory come from (round 5's arbiter: 2.2-2.8 x torch's float32)?
One iteration of every stage from identical state at B = 16: the whole-loss gradient of the HIP path (Adam's first moment / 0.1), of
the float32 oracle (autograd) and of the float64 oracle, per parameter block: |hip - f64| against |oracle32 - f64| (max and mean, relative
to the block's largest gradient); the same for the forward's vertices and joints.   usage (GPU): python3 scripts/experiments/gradient_error_vs_f64.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ihmr_amd.assets import synthetic_mano
from ihmr_amd.hip import PARAM_BLOCKS
from ihmr_amd.strategies import make_opt_strategy
from oracle.opt_ref import OptimizeRef
import test_gpu_parity as T

B = 16
arrays = (synthetic_mano(True), synthetic_mano(False))
for stage_id in range(4):
    orc, model, batch = T._oracle_and_model(arrays, B, 0, 1, seed=99)
    stage = make_opt_strategy(0)[stage_id]
    o64 = OptimizeRef(arrays[0], arrays[1], B, [stage], save_mid_freq=1, record=True, dtype=torch.float64)
    orc.strategy = [stage]
    for o in (orc, o64):
        o.set_input(batch); o.init_optimize(); o.optimize()
    model.set_input(batch); model.init_optimize(); model.run_stage(stage)
    torch.cuda.synchronize()
    m = model.buf["adam_m"].cpu().numpy().astype(np.float64) / 0.1
    for n in sorted(stage["update_params"]):
        g32, g64 = orc.trace[0]["grads"][n].astype(np.float64), o64.trace[0]["grads"][n]
        gh = m[:, PARAM_BLOCKS[n][1]:PARAM_BLOCKS[n][1] + PARAM_BLOCKS[n][2]].reshape(g64.shape)
        sc = np.abs(g64).max()
        eh, eo = np.abs(gh - g64) / sc, np.abs(g32 - g64) / sc
        print(f"stage {stage_id} dL/d{n:26s}: |hip-f64| max {eh.max():.2e} mean {eh.mean():.2e};  |oracle32-f64| max {eo.max():.2e} mean {eo.mean():.2e};  ratio of means {eh.mean() / max(eo.mean(), 1e-30):.2f}")
# forward only
orc, model, batch = T._oracle_and_model(arrays, B, 0, 1, seed=99)
o64 = OptimizeRef(arrays[0], arrays[1], B, [], save_mid_freq=1, dtype=torch.float64)
for o in (orc, o64):
    o.set_input(batch); o.init_optimize(); o.forward(); o.compute_loss(o.default_loss_weights)
model.set_input(batch); model.init_optimize(); model.forward_losses(); torch.cuda.synchronize()
g, r, a = model.get_pred_result(), orc.get_pred_result(), o64.get_pred_result()
for k in ("pred_right_hand_verts", "pred_left_hand_verts", "pred_joints_3d", "collision_loss_origin_scale"):
    eh, eo = np.abs(g[k].astype(np.float64) - a[k]), np.abs(r[k].astype(np.float64) - a[k])
    print(f"forward {k:30s}: |hip-f64| max {eh.max():.2e} mean {eh.mean():.2e};  |oracle32-f64| max {eo.max():.2e} mean {eo.mean():.2e};  ratio of means {eh.mean() / max(eo.mean(), 1e-30):.2f}")
