"""Candidate lists on / off over a long schedule (4 x (epoch + 1) iterations): results must be bit-identical.  usage: long_lists_check.py [B] [epoch]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ihmr_amd import two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 300
outs = []
for off in (False, True):
    opt = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                                cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="",
                                strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=epoch, sdf_no_candidate_lists=off)
    m = OptimizeModel(opt)
    fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    batch = synthetic_opt_batch(B, fwd, seed=4321)
    m.set_input(batch); m.init_optimize()
    if not off: m.sdf_counters_start()
    m.optimize()
    if not off: print({k: v for k, v in m.sdf_counters_stop().items() if k.startswith("voxels") or k == "inside_voxels"})
    torch.cuda.synchronize()
    outs.append(m.get_pred_result())
a, b = outs
bad = [k for k in a if isinstance(a[k], np.ndarray) and a[k].dtype.kind == "f" and not np.array_equal(a[k], b[k])]
print("keys differing:", bad)
print("max collision depth", float(a["collision_loss_origin_scale"].max()))
assert not bad
print("OK: bit-identical over", 4 * (epoch + 1), "iterations")
