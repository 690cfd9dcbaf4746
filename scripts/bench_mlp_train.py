#!/usr/bin/env python3
"""Timing of the IHMR-MLP training step (set_input -> retrive_prev_prediction -> forward -> compute_loss ->
optimize_parameters, src/train_mlp.py:93-99) on synthetic data, per stage of mlp_default, batch 128 (configs[2])."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ihmr_amd import two_hand
from ihmr_amd.mlp_model import MLPModel
from ihmr_amd.strategies import make_mlp_strategy
from ihmr_amd.synthetic import synthetic_opt_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
opt = types.SimpleNamespace(isTrain=True, dist=False, process_rank=-1, batchSize=B, inputSize=224, input_nc=3, num_joints=42,
                            total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3,
                            model_root="", checkpoints_dir="./checkpoints", strategy="mlp_default", total_epoch=1)
model = MLPModel(opt)
fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
batch = synthetic_opt_batch(B, fwd, seed=1234, with_feat=True)
batch["init_hand_trans"] = batch["init_hand_trans"][:, 0, :3].contiguous()
batch = {k: v.cuda() for k, v in batch.items()}
strategy = make_mlp_strategy()
model.set_update_info(strategy, B)
with torch.no_grad():
    model.set_input(batch); model.forward(forward_backbone=True); model.compute_loss(); model.save_pred_to_prev()
for sid, stage in enumerate(strategy):
    model.add_new_network(sid)
    def step():
        model.set_input(batch); model.retrive_prev_prediction(); model.forward()
        model.compute_loss(stage["loss_weights"]); model.optimize_parameters()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    first = float(model.loss)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"stage {sid} ({'+'.join(stage['update_params'])}): {dt * 1e3:.2f} ms / step of {B} samples = {B / dt:.0f} samples/s; "
          f"loss {first:.4f} -> {float(model.loss):.4f}")
