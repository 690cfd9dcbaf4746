"""Seeded synthetic batches with the reference's batch-dict schema (no dataset is available).

Schema = what ``src/data/opt_dataset.py:176-196`` emits per item, with the DataLoader's leading batch
dimension: ``joints_2d (B,42,3)``, ``joints_3d (B,42,4)``, ``mano_pose (B,96)``, ``mano_betas (B,20)``,
``mano_params_weight (B,2)``, ``hand_trans (B,1,4)``, ``hand_type_array (B,2)``, ``hand_type_valid (B,1)``,
``scale_ratio (B,)``, ``index (B,) int64``, ``init_cam (B,3)``, ``init_shape_params (B,20)``,
``init_pose_params (B,96)``, ``init_hand_trans (B,1,4)``, ``init_joints_2d (B,42,3)``,
``init_joints_3d (B,42,4)``, ``init_hand_trans_j (B,1,4)``; the Baseline/MLP schemas add
``img (B,3,224,224)``, ``do_flip``, ``img_feat (B,1024)`` (``baseline_dataset.py:212-226``,
``mlp_dataset.py:185-208``).

The target joints are the two-hand forward of a perturbed parameter set; the forward is injected as
``forward_fn(pose96, shape20, trans3) -> joints (B,42,3)`` so that tests can drive it with the CPU
oracle and the benchmark with the HIP path -- this module itself holds no model arithmetic.

Distribution (batch seed ``1234 + rank``, SURVEY.md 8(d), adjusted so that the hands really
inter-penetrate): the left hand is turned by ~pi about z so both hands point the same way and the
wrist-to-wrist translation is a few centimetres.
"""
from __future__ import annotations

import math
from typing import Callable, Dict

import numpy as np
import torch


def synthetic_opt_batch(batch_size: int, forward_fn: Callable, seed: int = 1234, first_index: int = 0,
                        with_image: bool = False, with_feat: bool = False, interlock: bool = False) -> Dict[str, torch.Tensor]:
    """``interlock`` (for the finger asset, ``assets.synthetic_mano(kind="fingers")``): instead of two hands lying on each other, the
    left hand keeps its native direction (the mirrored model points to -x) and comes from the front, its wrist ~26 cm ahead of the
    right wrist and half a finger pitch to the side: the fingers of one hand sit between -- and, with the random pose noise, in --
    the fingers of the other.  Same random draws as the default batch (which is unchanged), other means."""
    rng = np.random.RandomState(seed)
    B = batch_size
    f32 = np.float32
    init_pose = rng.normal(0.0, 0.2, size=(B, 96)).astype(f32)
    # left global orient (cols 48:51): ~pi about z, so that the mirrored left hand overlaps the right
    init_pose[:, 48:51] = (np.array([0.0, 0.0, 0.97 * math.pi]) + rng.normal(0.0, 0.1, size=(B, 3))).astype(f32)
    init_pose[:, 0:3] = rng.normal(0.0, 0.1, size=(B, 3)).astype(f32)
    init_shape = rng.normal(0.0, 0.5, size=(B, 20)).astype(f32)
    init_cam = (np.array([5.0, 0.0, 0.0]) + rng.normal(0.0, 0.05, size=(B, 3))).astype(f32)
    init_trans = (rng.uniform(-1.0, 1.0, size=(B, 3)) * np.array([0.02, 0.02, 0.012])
                  + np.array([0.0, 0.0, 0.034])).astype(f32)
    if interlock:
        init_pose[:, 48:51] -= np.array([0.0, 0.0, 0.97 * math.pi], f32)          # no half turn: the left hand points at the right one
        init_trans = (init_trans - np.array([0.0, 0.0, 0.034], f32)) * np.array([1.0, 0.5, 0.5], f32) + np.array([0.26, 0.009, 0.002], f32)

    pose_star = init_pose + rng.normal(0.0, 0.1, size=(B, 96)).astype(f32)
    shape_star = init_shape + rng.normal(0.0, 0.2, size=(B, 20)).astype(f32)
    trans_star = init_trans + rng.normal(0.0, 0.008, size=(B, 3)).astype(f32)
    noise3 = rng.normal(0.0, 0.002, size=(B, 42, 3)).astype(f32)
    noise_gt = rng.normal(0.0, 0.001, size=(B, 42, 3)).astype(f32)

    with torch.no_grad():
        j_star = forward_fn(torch.from_numpy(pose_star), torch.from_numpy(shape_star), torch.from_numpy(trans_star))
    j_star = np.asarray(j_star.detach().cpu().numpy(), dtype=f32)
    assert j_star.shape == (B, 42, 3)

    ones = np.ones((B, 42, 1), f32)
    init_j3d = j_star + noise3
    cam = init_cam[:, None, :]
    proj = lambda j: (j[:, :, :2] + cam[:, :, 1:3]) * cam[:, :, 0:1]
    gt_j3d = j_star + noise_gt
    one1 = np.ones((B, 1, 1), f32)

    batch = dict(
        joints_2d=np.concatenate([proj(gt_j3d), ones], 2),
        joints_3d=np.concatenate([gt_j3d, ones], 2),
        mano_pose=pose_star, mano_betas=shape_star,
        mano_params_weight=np.ones((B, 2), f32),
        hand_trans=np.concatenate([(gt_j3d[:, 21] - gt_j3d[:, 0])[:, None, :], one1], 2),
        hand_type_array=np.ones((B, 2), f32), hand_type_valid=np.ones((B, 1), f32),
        scale_ratio=np.ones((B,), f32), index=np.arange(first_index, first_index + B, dtype=np.int64),
        init_cam=init_cam, init_shape_params=init_shape, init_pose_params=init_pose,
        init_hand_trans=np.concatenate([init_trans[:, None, :], one1], 2),
        init_joints_2d=np.concatenate([proj(init_j3d), ones], 2),
        init_joints_3d=np.concatenate([init_j3d, ones], 2),
        init_hand_trans_j=np.concatenate([(init_j3d[:, 21] - init_j3d[:, 0])[:, None, :], one1], 2),
    )
    if with_image:
        batch["img"] = rng.uniform(-1.0, 1.0, size=(B, 3, 224, 224)).astype(f32)
        batch["do_flip"] = np.zeros((B,), f32)
    if with_feat:
        batch["img_feat"] = np.maximum(rng.normal(0.0, 0.5, size=(B, 1024)), 0).astype(f32)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
