#!/usr/bin/env python3
"""Counterpart of the reference's ``src/optimize.py`` (main loop :61-71, result gathering :78-102) on synthetic
data: one process per GPU (``python -m torch.distributed.run --nproc-per-node N -m ihmr_amd.run_optimize``),
samples sharded contiguously over the ranks, every rank refines its batches with :class:`OptimizeModel`, and the
four reported metrics are combined with ONE all-reduce of the metric sums (instead of the reference's
pickle-file gather + barrier).

    python -m ihmr_amd.run_optimize --num_samples 256 --batchSize 64 --opt_epoch 49
"""
from __future__ import annotations

import argparse
import json
import os
import time
import types

import numpy as np
import torch

from . import dist as D
from . import two_hand
from .evaluator import Evaluator
from .optimize_model import OptimizeModel
from .synthetic import synthetic_opt_batch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_samples", type=int, default=128)
    ap.add_argument("--batchSize", type=int, default=64)
    ap.add_argument("--opt_epoch", type=int, default=49, help="iterations per stage - 1 (reference default 300)")
    ap.add_argument("--save_mid_freq", type=int, default=10)
    ap.add_argument("--strategy", type=str, default="opt_default")
    ap.add_argument("--optimizer", type=str, default="adam", choices=["adam", "sgd"], help="options/opt_options.py: --optimizer")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--host_eval", action="store_true",
                    help="evaluate with the host (numpy) Evaluator on get_pred_result() exports, as the reference does; "
                         "default: metrics on the device (ihmr_eval_metrics), no export")
    ap.add_argument("--streams", type=int, default=1,
                    help="model instances driven side by side on their own HIP streams (2 x --fuse_batches 8 saturates one MI355X; "
                         "more than 4 need GPU_MAX_HW_QUEUES raised in the environment)")
    ap.add_argument("--fuse_batches", type=int, default=1,
                    help="consecutive batches carried by one launch sequence (per-sample results identical to separate batches)")
    args = ap.parse_args(argv)

    rank, world = D.init_dist()
    if world == 1:
        torch.cuda.set_device(0)
    opt = types.SimpleNamespace(isTrain=False, dist=world > 1, process_rank=rank if world > 1 else -1, batchSize=args.batchSize,
                                inputSize=224, num_joints=42, total_params_dim=122, cam_params_dim=3, pose_params_dim=96,
                                shape_params_dim=20, trans_params_dim=3, model_root="", strategy=args.strategy,
                                save_mid_freq=args.save_mid_freq, optimizer=args.optimizer, opt_epoch=args.opt_epoch)
    model = OptimizeModel(opt)
    G, S = max(1, args.fuse_batches), max(1, args.streams)
    # one model instance (own buffers, workspace, graphs) per stream; `model` doubles as the first single-batch instance
    fused = [model if (G == 1 and i == 0) else OptimizeModel(types.SimpleNamespace(**vars(opt), fuse_batches=G)) for i in range(S)]
    singles = [model] + [None] * (S - 1)                                      # remainder of fewer than G batches: batch by batch
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [main_stream]
    evaluator = Evaluator(model.mano_models)
    fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]

    idx, is_pad = D.shard_indices(args.num_samples, args.batchSize, rank, world)
    Bsz = args.batchSize
    n_batches = len(idx) // Bsz
    t0 = time.time()
    done = 0
    while done < n_batches:
        jobs = []                                       # up to S launch sequences in flight
        for i in range(S):
            if done >= n_batches:
                break
            g = G if n_batches - done >= G else 1
            s0 = done * Bsz
            sel, pad = idx[s0:s0 + g * Bsz], is_pad[s0:s0 + g * Bsz]
            # the synthetic "dataset": sample i is generated from seed + i's batch; padding entries repeat sample 0's batch row
            parts = [synthetic_opt_batch(Bsz, fwd, seed=args.seed + int(sel[q * Bsz]), first_index=int(sel[q * Bsz])) for q in range(g)]
            data = parts[0] if g == 1 else {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
            if g == G:
                mdl = fused[i]
            else:
                if singles[i] is None:
                    singles[i] = OptimizeModel(opt)
                mdl = singles[i]
            jobs.append((mdl, streams[i], data, sel, pad))
            done += g
        for mdl, st, data, _, _ in jobs:
            st.wait_stream(main_stream)                 # the synthetic data was produced on the caller's stream
            with torch.cuda.stream(st):
                mdl.set_input(data)
                mdl.init_optimize()
        for stage in model.strategy:                    # stage by stage over the jobs: every stream stays fed
            for mdl, st, *_ in jobs:
                with torch.cuda.stream(st):
                    mdl.run_stage(stage)
        for mdl, st, _, sel, pad in jobs:
            with torch.cuda.stream(st):
                mdl.forward_losses(mdl.default_loss_weights)
                if args.host_eval:
                    pred = mdl.get_pred_result()
                    n0 = len(evaluator.pred_results)
                    evaluator.update(sel, pred)
                    new = evaluator.pred_results[n0:]
                    evaluator.pred_results = evaluator.pred_results[:n0] + [p for p, k in zip(new, ~pad) if k]   # drop padding duplicates
                else:
                    evaluator.update_device(mdl.pred_joints_3d, mdl.buf["gt_joints_3d"], mdl.collision_loss_origin_scale,
                                            keep=torch.from_numpy(~pad))
        for _, st, *_ in jobs:
            main_stream.wait_stream(st)                 # the next round's data generation reuses the instances' MANO handles
    torch.cuda.synchronize()
    sums = D.reduce_metrics(evaluator.metric_sums())
    elapsed = time.time() - t0
    if rank == 0:
        m = Evaluator.metrics_from_sums(sums)
        for k in ("mpjpe_3d", "inter_mpjpe_3d", "collision_ave", "collision_max"):
            print(f"{k} : {m[k]:.3f} (optimize)")
        print(json.dumps(dict(num_samples=args.num_samples, world=world, seconds=elapsed, **m)))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return Evaluator.metrics_from_sums(sums)


if __name__ == "__main__":
    main()
