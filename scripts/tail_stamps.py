#!/usr/bin/env python3
"""Phase times of opt_tail_kernel per sample from an experiment build:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -DTAIL_STAMPS ihmr_amd/csrc/ihmr_hip.hip -o build/ab/lib_tailstamps.so
   IHMR_HIP_LIBRARY=build/ab/lib_tailstamps.so python3 scripts/tail_stamps.py [fuse]
Runs the four stages at `fuse` x 64 samples per launch and prints, per kernel form, the mean / p90 / max shader-clock time of each
phase of a sample's workgroup (thread 0: sampling + losses, LBS backward, optimizer step, skeletons, skinning)."""
import ctypes as C, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 64
o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                          cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                          save_mid_freq=10, optimizer="adam", opt_epoch=49, fuse_batches=G)
m = OptimizeModel(o)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i) for i in range(G)]
inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
m.set_input(inp); m.init_optimize(); m.optimize()
torch.cuda.synchronize()
L = hip.lib()
L.ihmr_debug_tail_stamps.argtypes = [C.c_void_p, C.c_int]
names = ["sampling + losses", "LBS backward", "optimizer step", "skeletons", "skinning"]
for sid, stage in enumerate(m.strategy):
    m.set_input(inp); m.init_optimize()
    for s in m.strategy[:sid]:
        m.run_stage(s)
    L.ihmr_debug_tail_stamps(None, 1)
    m.run_stage(stage)
    raw = np.zeros(4 * 4096 * 8, np.int64)
    L.ihmr_debug_tail_stamps(raw.ctypes.data, 0)
    samp = raw[3 * 4096 * 8:].reshape(4096, 8)
    raw = raw[:3 * 4096 * 8].reshape(3, 4096, 8)
    sm = samp[samp[:, 7] > 0]
    if len(sm):
        u = sm[:, :6] / sm[:, 7:8] / 2403.0
        print(f"stage {sid} sampler (thread 0, all tail launches): " + "; ".join(f"{n} {u[:, k].mean():.2f}" for k, n in enumerate(
            ["issue first loads", "they land (+ bitmap barrier)", "bitmap words", "phi lands", "values + gradients", "block sum"])))
    for form, label in enumerate(("<false,false>", "<true,false>", "<true,true>")):
        r = raw[form][raw[form][:, 7] > 0]
        if not len(r):
            continue
        us = r[:, :5] / r[:, 7:8] / 2403.0           # s_memtime: 2403 ticks per us (scripts/experiments/clock_ratio.hip)
        print(f"stage {sid} opt_tail_kernel{label}: {len(r)} samples x {int(r[0, 7])} launches; " +
              "; ".join(f"{n} {us[:, k].mean():.2f} (p90 {np.percentile(us[:, k], 90):.2f})" for k, n in enumerate(names) if us[:, k].max() > 0) +
              f"; total {us.sum(1).mean():.2f} (max {us.sum(1).max():.2f}) us")
