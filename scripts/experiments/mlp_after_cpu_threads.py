"""Why does bench.py's in-line IHMR-MLP figure (123 k images/s) trail the standalone one (135 k)?  Runs bench.secondary("mlp") (a) fresh,
(b) after a CPU leg that used every core with torch (as bench.py's cpu_baseline does), (c) after the C oracle's OpenMP region ran too.
usage (GPU box): python3 scripts/experiments/mlp_after_cpu_threads.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

def show(tag):
    d = bench.secondary("mlp", with_cpu=False)
    print(f"{tag}: {d['value'] / 1e3:.1f} k images/s, wall {d['ms_per_step']:.3f} ms, GPU {d['gpu_ms_per_batch']:.3f} ms per batch", flush=True)

show("fresh")
cores = os.cpu_count() or 1
torch.set_num_threads(cores)
a = torch.randn(4096, 4096)
t0 = time.perf_counter()
for _ in range(5):
    (a @ a).sum().item()
print(f"CPU leg with {cores} torch threads: {time.perf_counter() - t0:.2f} s", flush=True)
show("after a torch CPU leg on every core")
if len(sys.argv) > 1:
    from oracle import sdf_ref
    from ihmr_amd.assets import synthetic_mano
    m = synthetic_mano(True)
    keys = {k.lower(): k for k in m}
    v = torch.as_tensor(m[keys.get("v_template", "v_template")], dtype=torch.float32)
    v = (v - v.mean(0)) / (v - v.mean(0)).abs().max() * 0.9
    f = torch.as_tensor(m[keys.get("f", keys.get("faces", "f"))].astype("int32"))
    t0 = time.perf_counter()
    for _ in range(3):
        sdf_ref.sdf_grid(v[None].repeat(64, 1, 1), f)
    print(f"C oracle OpenMP leg: {time.perf_counter() - t0:.2f} s", flush=True)
    show("after the C oracle's OpenMP region as well")
    # what bench.py has alive before its secondary configs: several OPT instances on their own streams, graphs captured
    import types
    from ihmr_amd.optimize_model import OptimizeModel
    from ihmr_amd import two_hand
    from ihmr_amd.synthetic import synthetic_opt_batch
    o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=64, inputSize=224, num_joints=42, total_params_dim=122,
                              cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                              save_mid_freq=10, optimizer="adam", opt_epoch=49)
    streams = [torch.cuda.Stream() for _ in range(6)]
    models = []
    for st in streams:
        with torch.cuda.stream(st):
            mm = OptimizeModel(o)
            fwd = lambda p, s_, t: two_hand.forward_from_packed(mm.mano_models["right"], p.cuda(), s_.cuda(), t.cuda())[2]
            inp = {k: v_.cuda() for k, v_ in synthetic_opt_batch(64, fwd, seed=1).items()}
            mm.set_input(inp); mm.init_optimize(); mm.optimize()
        models.append(mm)
    torch.cuda.synchronize()
    show("with six OPT instances and their streams alive")
    del models, mm
    torch.cuda.empty_cache()
    show("after deleting them (streams still alive)")
