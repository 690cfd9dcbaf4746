import sys, os, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ihmr_amd.assets import synthetic_mano
from ihmr_amd.optimize_model import OptimizeModel
from test_gpu_parity import _two_hand_verts, _make_opt
B, epoch, freq, graphs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ma = (synthetic_mano(True), synthetic_mano(False))
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 11
_, b1 = _two_hand_verts(ma, B, seed)
opt = _make_opt(B, epoch=epoch, save_mid_freq=freq); opt.use_graphs = bool(graphs)
m = OptimizeModel(opt)
if os.environ.get('SIDE'):
    _st = torch.cuda.Stream(); torch.cuda.set_stream(_st)
m.set_input(b1); m.init_optimize()
for i, st in enumerate(m.strategy):
    m.run_stage(st); torch.cuda.synchronize(); print("stage", i, "ok", flush=True)
m.forward_losses(); torch.cuda.synchronize(); print("done", B, epoch, freq, graphs, flush=True)
mode = os.environ.get("SECOND", "same")
if mode == "nograph": m.use_graphs = False
m.set_input(b1); m.init_optimize()
for i, st in enumerate(m.strategy):
    m.run_stage(st); torch.cuda.synchronize(); print("second: stage", i, "ok", flush=True)
m.forward_losses(); torch.cuda.synchronize(); print("second run ok")
