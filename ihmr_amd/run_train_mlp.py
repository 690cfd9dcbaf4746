#!/usr/bin/env python3
"""Counterpart of the reference's ``src/train_mlp.py`` (main loop :56-133) on synthetic data: the backbone prediction
fills the "prev" tables, then every stage of the strategy adds a sub-network, trains it for ``epoch`` passes
(``set_input -> retrive_prev_prediction -> forward -> compute_loss(stage weights) -> optimize_parameters``), decays the
learning rate, and ends with the selection pass that updates the "prev" tables.  One process per GPU
(``python -m torch.distributed.run --nproc-per-node N -m ihmr_amd.run_train_mlp``): every rank trains on its own shard,
the sub-network's gradients are averaged with one all-reduce per step (:func:`ihmr_amd.dist.all_reduce_gradients`).

    python -m ihmr_amd.run_train_mlp --num_samples 512 --batchSize 128 --epochs 2
"""
from __future__ import annotations

import argparse
import json
import time
import types

import torch

from . import dist as D
from . import two_hand
from .mlp_model import MLPModel
from .strategies import make_mlp_strategy
from .synthetic import synthetic_opt_batch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_samples", type=int, default=256, help="samples per rank")
    ap.add_argument("--batchSize", type=int, default=128)
    ap.add_argument("--epochs", type=int, default=0, help="passes per stage (0 = the strategy's own `epoch`)")
    ap.add_argument("--stages", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--save", action="store_true", help="write <checkpoints_dir>/latest_net_mlp_stage_XX.pth after every stage")
    args = ap.parse_args(argv)

    rank, world = D.init_dist()
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()   # (a one-rank group under torch.distributed.run too)
    if world == 1:
        torch.cuda.set_device(0)
    B = args.batchSize
    torch.manual_seed(args.seed)                              # same initial sub-network weights on every rank and in every run (they are broadcast anyway)
    opt = types.SimpleNamespace(isTrain=True, dist=grouped, process_rank=rank if grouped else -1, batchSize=B, inputSize=224,
                                input_nc=3, num_joints=42, total_params_dim=122, cam_params_dim=3, pose_params_dim=96,
                                shape_params_dim=20, trans_params_dim=3, model_root="", checkpoints_dir="./checkpoints",
                                strategy="mlp_default", total_epoch=1)
    model = MLPModel(opt)
    fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    n_batches = max(1, args.num_samples // B)
    data = []
    for i in range(n_batches):                                  # this rank's shard of the synthetic "dataset", resident in HBM
        b = synthetic_opt_batch(B, fwd, seed=args.seed + 1000 * rank + i, first_index=i * B, with_feat=True)
        b["init_hand_trans"] = b["init_hand_trans"][:, 0, :3].contiguous()
        data.append({k: v.cuda() for k, v in b.items()})
    strategy = make_mlp_strategy()[:args.stages]
    model.set_update_info(strategy, n_batches * B)
    with torch.no_grad():                                       # train_mlp.py:60-66
        for b in data:
            model.set_input(b); model.forward(forward_backbone=True); model.compute_loss(); model.save_pred_to_prev()
    log = []
    for sid, stage in enumerate(strategy):
        model.add_new_network(sid)
        total_epoch = args.epochs or stage["epoch"]
        model.opt.total_epoch = total_epoch
        torch.cuda.synchronize()
        t0, steps = time.time(), 0
        first = last = None                                     # mean loss over the batches of the first / the last epoch (the same batches)
        for epoch in range(1, total_epoch + 1):                 # train_mlp.py:79-121
            ep = []
            for b in data:
                model.set_input(b)
                model.retrive_prev_prediction()
                model.forward()
                model.compute_loss(stage["loss_weights"])
                model.optimize_parameters()
                steps += 1
                ep.append(model.loss)                           # (read back after the epoch: no host sync inside the step)
            last = float(sum(float(x) for x in ep) / len(ep))
            if first is None:
                first = last
            model.update_learning_rate(epoch, sid)
        torch.cuda.synchronize()
        dt = time.time() - t0
        kept = 0
        with torch.no_grad():                                   # train_mlp.py:124-133: selection pass -> "prev" tables
            for b in data:
                model.set_input(b)
                model.retrive_prev_prediction()
                model.forward()
                model.compute_loss()
                kept += int(model.select_better_params(sid).sum())
                model.save_pred_to_prev()
        if args.save and rank == 0:
            model.save("latest", sid)
        log.append(dict(stage=sid, params="+".join(stage["update_params"]), steps=steps, ms_per_step=1e3 * dt / max(steps, 1),
                        samples_per_s=world * B * steps / dt, loss_first=first, loss_last=last, kept=kept, of=n_batches * B))
        if rank == 0:
            print(json.dumps(log[-1]))
    if torch.distributed.is_available() and torch.distributed.is_initialized():     # (also a one-rank group under torch.distributed.run)
        torch.distributed.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
