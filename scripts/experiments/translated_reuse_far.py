import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
from ihmr_amd.assets import synthetic_mano
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.strategies import make_opt_strategy
ma = (synthetic_mano(True), synthetic_mano(False))
B = 64
_, batch = T._two_hand_verts(ma, B, 4242)
base = make_opt_strategy(199)
print({k: base[0][k] for k in base[0] if k != "loss_weights"})
for lr_scale in (1.0, 10.0, 50.0):
    st = dict(base[0]); st["lr"] = base[0].get("lr", 0.01) * lr_scale if "lr" in base[0] else None
    outs = []
    for off in (False, True):
        opt = T._make_opt(B, epoch=199, save_mid_freq=10)
        opt.sdf_no_translated_reuse = off
        m = OptimizeModel(opt)
        s0 = dict(m.strategy[0])
        for key in ("lr", "learning_rate"):
            if key in s0:
                s0[key] = s0[key] * lr_scale
        m.strategy = [s0]
        m.set_input(batch); m.init_optimize(); m.optimize(); torch.cuda.synchronize()
        r = m.get_pred_result()
        outs.append((r, torch.stack(m.selected_history).cpu().numpy(), m.buf["snap_loss"].cpu().numpy().copy()))
    (a, sa, la), (b, sb, lb) = outs
    moved = float(np.abs(a["pred_hand_trans"] - batch["init_hand_trans"][:, 0, :3].numpy()).max()) if "init_hand_trans" in batch else -1
    d = {k: float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()) for k in ("pred_left_hand_verts", "pred_hand_trans", "collision_loss_origin_scale")}
    print(f"lr x {lr_scale}: left hand moved up to {moved:.4f} m; selections equal {np.array_equal(sa, sb)}; max diff {d}; snap_loss {np.abs(la - lb).max():.3e}", flush=True)
