#!/usr/bin/env python3
"""Counterpart of the reference's ``src/train_baseline.py`` (main loop :60-108) on synthetic data:
``set_input -> forward -> optimize_parameters`` per batch, learning-rate update and checkpoint per epoch.  One process per GPU
(``python -m torch.distributed.run --nproc-per-node N -m ihmr_amd.run_train_baseline``); the encoder's flat gradient
(26 M floats) is averaged over the ranks in 25 MB buckets that are all-reduced while the backward pass is still running.

    python -m ihmr_amd.run_train_baseline --num_samples 256 --batchSize 64 --total_epoch 2
"""
from __future__ import annotations

import argparse
import json
import time
import types

import torch

from . import dist as D
from . import two_hand
from .baseline_model import InterHandModel
from .synthetic import synthetic_opt_batch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_samples", type=int, default=128, help="samples per rank")
    ap.add_argument("--batchSize", type=int, default=64)
    ap.add_argument("--total_epoch", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--lr_decay_type", type=str, default="none", choices=["none", "cosine"])
    ap.add_argument("--use_collision_loss", action="store_true")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--save", action="store_true")
    args = ap.parse_args(argv)
    rank, world = D.init_dist()
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()   # (a one-rank group under torch.distributed.run too)
    if world == 1:
        torch.cuda.set_device(0)
    B = args.batchSize
    opt = types.SimpleNamespace(isTrain=True, dist=grouped, process_rank=rank if grouped else -1, batchSize=B, inputSize=224, input_nc=3,
                                num_joints=42, total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20,
                                trans_params_dim=3, model_root="", mean_param_file="mean_mano_params.pkl", checkpoints_dir="./checkpoints",
                                lr=args.lr, lr_decay_type=args.lr_decay_type, total_epoch=args.total_epoch,
                                use_collision_loss=args.use_collision_loss)
    torch.manual_seed(args.seed)                              # same initial weights on every rank (they are broadcast anyway)
    model = InterHandModel(opt)
    fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    data = []
    for i in range(max(1, args.num_samples // B)):
        b = synthetic_opt_batch(B, fwd, seed=args.seed + 1000 * rank + i, first_index=i * B, with_image=True)
        data.append({k: v.cuda() for k, v in b.items()})
    log = []
    for epoch in range(1, args.total_epoch + 1):
        torch.cuda.synchronize()
        t0, first = time.time(), None
        for b in data:
            model.set_input(b)
            model.forward_train()
            model.optimize_parameters()
            if first is None:
                first = model.get_current_errors()["total_loss"]
        last = model.get_current_errors()["total_loss"]
        torch.cuda.synchronize()
        dt = time.time() - t0
        model.update_learning_rate(epoch)
        if args.save and rank == 0:
            model.save("latest", epoch)
        log.append(dict(epoch=epoch, steps=len(data), ms_per_step=1e3 * dt / len(data), images_per_s=world * B * len(data) / dt,
                        loss_first=first, loss_last=last))
        if rank == 0:
            print(json.dumps(log[-1]))
    if torch.distributed.is_available() and torch.distributed.is_initialized():     # (also a one-rank group under torch.distributed.run)
        torch.distributed.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
