#!/usr/bin/env python3
"""Per launch and hand: how many inside voxels go to the list search / the full search (an -DSDF_HANDLOG build).  The question behind it
(round 6): if a hand's own prep workgroup also evaluated the hand's voxels -- no separate distance launch at batch 64 -- how long would
the slowest hand of an iteration be?  Prints per stage the distribution over ITERATIONS of max-over-hands of list items (32 voxels)
and full-search items (16 voxels), and how many hands rebuild.
usage: IHMR_HIP_LIBRARY=build/handlog.so python3 scripts/experiments/hand_work_log.py [batch]"""
import ctypes as C
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opt = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                            cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="",
                            strategy="opt_default", save_mid_freq=10, optimizer="adam", opt_epoch=49)
m = OptimizeModel(opt)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
L = hip.lib()
L.ihmr_debug_handlog.restype = C.c_long
L.ihmr_debug_handlog.argtypes = [C.c_void_p, C.c_long]
for seed in (1234, 2234):
    m.set_input(synthetic_opt_batch(B, fwd, seed=seed)); m.init_optimize()
    for si, stage in enumerate(m.strategy):
        assert L.ihmr_debug_handlog(None, 1 << 20) == 0
        m.run_stage(stage)
        torch.cuda.synchronize()
        rec = np.zeros((1 << 20, 4), np.uint32)
        n = L.ihmr_debug_handlog(rec.ctypes.data, 0)
        rec = rec[:n].reshape(-1, 2 * B, 4)            # launches are sequential: 2B records per launch
        na, nb, fl = rec[:, :, 1].astype(int), rec[:, :, 2].astype(int), rec[:, :, 3]
        items_a, items_b = -(-na // 32), -(-nb // 16)
        rebuild = ((fl & 3) == 0)                      # neither reused nor static
        q = lambda x: "/".join(f"{np.percentile(x, p):.0f}" for p in (50, 90, 100))
        # a 16-wave workgroup = 4 groups of 4 waves: rounds of (full items 8 us, list items 4.5 us) if the hand does its own work
        own_us = np.ceil(items_b / 4.0) * 8.0 + np.ceil(items_a / 4.0) * 4.5
        print(f"seed {seed} stage {si}: {rec.shape[0]} launches; per iteration max over hands: list items {q(items_a.max(1))} (p50/p90/max), "
              f"full items {q(items_b.max(1))}, rebuilding hands {q(rebuild.sum(1))}, hands with > 1 full item {q((items_b > 1).sum(1))}; "
              f"mean voxels per hand list {na.mean():.1f} full {nb.mean():.1f}; own-work estimate of the slowest hand {q(own_us.max(1))} us, "
              f"mean hand {own_us.mean():.1f} us; iterations whose slowest hand stays <= 9 us: {100.0 * (own_us.max(1) <= 9.0).mean():.0f} %")
