import sys, os, types, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
from ihmr_amd import two_hand
B, epoch = int(sys.argv[1]), int(sys.argv[2])
def mk():
    return OptimizeModel(types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
        cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=epoch))
m = mk()
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
batch = {k: v.cuda() for k, v in synthetic_opt_batch(B, fwd, seed=1234).items()}
outs = []
for r in range(6):
    mm = m if r < 4 else mk()
    mm.set_input(batch); mm.init_optimize(); mm.optimize(); torch.cuda.synchronize()
    res = mm.get_pred_result()
    outs.append(res)
    print(r, "mean pen", float(np.mean(res["collision_loss_origin_scale"])), "pose sum", float(np.abs(res["pred_pose_params"]).sum()), "sel", [int(s.sum()) for s in mm.selected_history])
for k in ("pred_pose_params", "pred_shape_params", "pred_hand_trans", "collision_loss_origin_scale"):
    d = [float(np.abs(outs[0][k] - o[k]).max()) for o in outs[1:]]
    print(k, "max|diff| vs run 0:", d)
