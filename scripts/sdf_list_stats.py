"""Work counters of the collision kernels over the four stages of one batch-64 refinement (GPU): how many inside voxels were answered
from their candidate lists, how many went through the full search (new voxels / hands whose lists were rebuilt), sphere tests and
exact distances executed.  usage: python scripts/sdf_list_stats.py [batch]"""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ihmr_amd import two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opt = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                            cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="",
                            strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=49)
m = OptimizeModel(opt)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
m.set_input(synthetic_opt_batch(B, fwd, seed=1234)); m.init_optimize()
for i, stage in enumerate(m.strategy):
    m.sdf_counters_start()
    m.run_stage(stage)
    c = m.sdf_counters_stop()
    print(f"stage {i}: " + ", ".join(f"{k} {v}" for k, v in c.items()))
