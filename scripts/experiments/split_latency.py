#!/usr/bin/env python3
"""One batch of 64 samples refined as S independent sub-batches of 64 / S on S streams (the samples of IHMR-OPT are independent:
optimize_model.py:393-407 sums per-sample losses): wall time of the 200-iteration stage loop for ALL 64 samples, per iteration.
usage: python3 scripts/experiments/split_latency.py [total batch] [reps]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ihmr_amd import two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9


def opt(B):
    return types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                                 cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                                 save_mid_freq=10, optimizer="adam", opt_epoch=49)


probe = OptimizeModel(opt(BT))
fwd = lambda p, s, t: two_hand.forward_from_packed(probe.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
full = {k: v.cuda() for k, v in synthetic_opt_batch(BT, fwd, seed=1234).items()}
for S in (1, 2, 4):
    B = BT // S
    streams = [torch.cuda.Stream() for _ in range(S)]
    models, parts = [], []
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            m = OptimizeModel(opt(B))
            part = {k: v[i * B:(i + 1) * B].contiguous() for k, v in full.items()}
            m.set_input(part); m.init_optimize(); m.optimize()
        models.append(m); parts.append(part)
    torch.cuda.synchronize()
    tot = []
    for _ in range(reps):
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                models[i].set_input(parts[i]); models[i].init_optimize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in range(len(models[0].strategy)):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    models[i].run_stage(models[i].strategy[st])
        torch.cuda.synchronize()
        tot.append(time.perf_counter() - t0)
    print(f"{BT} samples as {S} x {B} on {S} streams: {1e3 * np.median(tot) / 200:.4f} ms per refinement iteration "
          f"(min {1e3 * min(tot) / 200:.4f}); {BT / np.median(tot) / 1e3:.2f} k images/s over the stage loop")
