import sys, os, types
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ihmr_amd import two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
B = 512
opt = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                            cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="",
                            strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=49)
m = OptimizeModel(opt)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
m.set_input(synthetic_opt_batch(B, fwd, seed=1234)); m.init_optimize()
def report(tag):
    m.forward_losses(m.default_loss_weights); torch.cuda.synchronize()
    d = m.buf["coll_origin_scale"].cpu().numpy()
    r, l = (d[:, :778] > 0).sum(1), (d[:, 778:] > 0).sum(1)
    print(tag, "hands with NO penetrating vertex in their grid: right", float((r == 0).mean()), "left", float((l == 0).mean()), "mean penetrating verts", float(r.mean()), float(l.mean()))
report("init")
for i, stage in enumerate(m.strategy):
    m.run_stage(stage); report(f"after stage {i}")
