import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from helpers import ragged_opt_batch
from ihmr_amd.assets import synthetic_mano
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.strategies import make_opt_strategy
ma = (synthetic_mano(True), synthetic_mano(False))
for B, epoch in ((16, 39), (64, 14)):
    _, batch = T._two_hand_verts(ma, B, 3100 + B)
    if B == 16:
        batch = ragged_opt_batch(batch)
    base = make_opt_strategy(epoch)
    extra = []
    for params, like in ((["pred_right_orient"], 1), (["pred_left_pose_params"], 2), (["pred_cam_params", "pred_hand_trans"], 0)):
        st = dict(base[like]); st["update_params"] = params
        extra.append(st)
    for nstage in (1, 4, 7):
        outs = []
        for mode in (0, 2, 1):
            opt = T._make_opt(B, epoch=epoch, save_mid_freq=5)
            opt.sdf_no_static_reuse = mode == 1
            opt.sdf_no_translated_reuse = mode == 2
            m = OptimizeModel(opt)
            m.strategy = (base + extra)[:nstage]
            for rep in range(2):
                m.set_input(batch); m.init_optimize(); m.optimize()
                torch.cuda.synchronize()
            outs.append((m.get_pred_result(), m.buf["snap_loss"].cpu().numpy().copy(), m.buf["adam_m"].cpu().numpy().copy(), m.buf["loss_batch"].cpu().numpy().copy()))
        for name, (x, y) in (("0 vs 2", (outs[0], outs[1])), ("2 vs 1", (outs[1], outs[2]))):
            d = {k: float(np.abs(x[0][k].astype(np.float64) - y[0][k].astype(np.float64)).max()) for k in ("pred_left_hand_verts", "pred_hand_trans", "collision_loss_origin_scale", "collision_loss")}
            print(f"B {B} stages {nstage} {name}: result {d}; snap_loss {np.abs(x[1] - y[1]).max():.3e}, adam_m {np.abs(x[2] - y[2]).max():.3e}, loss_batch {np.abs(x[3] - y[3]).max():.3e}", flush=True)
