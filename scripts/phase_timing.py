import sys, os, ctypes as C, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# use the timing build of the library
import ihmr_amd.hip as hip
hip.LIB_PATH = os.path.join(ROOT, "scripts", "libihmr_hip_timing.so")
import types, torch, numpy as np
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
from ihmr_amd import two_hand
from ihmr_amd.strategies import make_opt_strategy
B=64
opt = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=3)
model = OptimizeModel(opt)
fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
batch = {k: v.cuda() for k, v in synthetic_opt_batch(B, fwd, seed=1234).items()}
model.set_input(batch); model.init_optimize()
for stage_id in (2, 3):
    st = make_opt_strategy(3)[stage_id]
    for _ in range(3): model.run_stage(st)
    out = (C.c_longlong * 256)()
    hip.lib().ihmr_debug_read.argtypes = [C.c_void_p]
    hip.lib().ihmr_debug_read(out)
    d = np.array(out[:64], dtype=np.int64)
    def seg(name, ids):
        print(f"stage{stage_id} {name}: " + "  ".join(f"{(d[b]-d[a])/2400:.1f}us" for a, b in zip(ids[:-1], ids[1:])) + f"   total {(d[ids[-1]]-d[ids[0]])/2400:.1f}us")
    seg("skin [load-skel | shape | pose | weights+skin+store]", [0,1,2,3,4])
    seg("bwd1 [gsum | load | dvp | dA | chain | tail]", [10,11,12,13,14,15,16])
    seg("prep [bbox | norm+mark | triangles+parity | publish-scan | publish | stats]", [20,21,22,27,28,29,26]); dd=np.array(out[:200],dtype=np.int64); print("   per-wave triangle-phase us", [round(float(x)/2400,1) for x in dd[100:116]]); print("   after-table us it0", [round(float(x)/2400,1) for x in dd[120:136]], "it1", [round(float(x)/2400,1) for x in dd[136:152]]); print("   max cols it0", dd[160:176].tolist(), "it1", dd[176:192].tolist())
    seg("parity [loop | publish]", [30,31,32])
    seg("dist [loop]", [40,41]); dd=np.array(out[:100],dtype=np.int64); print("   dist max over waves: total %.1fus load %.1fus vox %.1fus items %d vox %d; per-XCD totals %s" % (dd[80]/2400, dd[81]/2400, dd[82]/2400, dd[83], dd[84], dd[90:98].tolist()))
    seg("sample [loop | reduce]", [50,51,52])
