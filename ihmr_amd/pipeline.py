"""Chaining the three models on the device, and the prediction file the reference chains them through.

The reference runs IHMR-Baseline over a dataset, stores its predictions in a pickle keyed by image path
(``hand26m_pred_path``; read back by ``data/data_utils.py:42-70`` with the keys ``pred_cam_params, pred_shape_params,
pred_pose_params, pred_hand_trans, joints_2d, joints_3d, img_feat``) and builds the IHMR-MLP / IHMR-OPT inputs from
it in the dataset classes (``data/opt_dataset.py:134-196``, ``data/mlp_dataset.py:160-208``).  Here the same three steps
are functions on DEVICE tensors, so a Baseline batch can feed the refinement without a round trip through the
host or the disk; the file form is kept, with the reference's schema, for interchange.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Sequence

import numpy as np
import torch

from . import ry_utils

PRED_KEYS = ("pred_cam_params", "pred_shape_params", "pred_pose_params", "pred_hand_trans", "joints_2d", "joints_3d", "img_feat")


def predictions_from_baseline(model) -> Dict[str, torch.Tensor]:
    """After ``InterHandModel.test()``: the per-sample record of the prediction file, as device tensors."""
    return OrderedDict(
        pred_cam_params=model.pred_cam_params.contiguous(), pred_shape_params=model.pred_shape_params.contiguous(),
        pred_pose_params=model.pred_pose_params.contiguous(), pred_hand_trans=model.pred_hand_trans.reshape(-1, 3).contiguous(),
        joints_2d=model.pred_joints_2d.contiguous(), joints_3d=model.pred_joints_3d.contiguous(),
        img_feat=model.encoder.feat.contiguous())


def save_pred_file(path: str, img_paths: Sequence[str], preds: Dict[str, torch.Tensor]) -> None:
    """``{img_path: {key: float32 array}}`` -- what ``load_anno_pred_data`` indexes (data_utils.py:51-66)."""
    host = {k: preds[k].detach().cpu().numpy().astype(np.float32) for k in PRED_KEYS}
    ry_utils.save_pkl(path, {p: {k: host[k][i] for k in PRED_KEYS} for i, p in enumerate(img_paths)})


def load_pred_file(path: str, img_paths: Sequence[str], device=None) -> Dict[str, torch.Tensor]:
    data = ry_utils.load_pkl(path)
    out = OrderedDict()
    for k in PRED_KEYS:
        t = torch.from_numpy(np.stack([np.asarray(data[p][k], dtype=np.float32) for p in img_paths]))
        out[k] = t if device is None else t.to(device)
    return out


def refinement_batch(preds: Dict[str, torch.Tensor], anno: Dict[str, torch.Tensor], for_mlp: bool = False) -> Dict[str, torch.Tensor]:
    """The batch dict ``OptimizeModel.set_input`` / ``MLPModel.set_input`` take, from predictions + annotations
    (``joints_2d (B,42,3)``, ``joints_3d (B,42,4)``, ``mano_pose``, ``mano_betas``, ``mano_params_weight``, ``hand_trans (B,1,4)``,
    ``hand_type_array``, ``index``), as ``opt_dataset.py:134-196`` builds it: unit scores appended to the predicted joints,
    ``init_hand_trans_j`` = predicted joint 21 - joint 0 with weight 1.  (The reference also sends ``init_joints_2d``
    through the crop / resize of the image; the caller applies that transform when it has one.)"""
    B = preds["pred_cam_params"].shape[0]
    dev = preds["pred_cam_params"].device
    one = torch.ones(B, 42, 1, device=dev)
    j3 = preds["joints_3d"]
    w1 = torch.ones(B, 1, device=dev)
    trans = preds["pred_hand_trans"].reshape(B, 3)
    batch = OrderedDict((k, v.to(dev)) for k, v in anno.items())
    batch.update(
        init_cam=preds["pred_cam_params"], init_shape_params=preds["pred_shape_params"], init_pose_params=preds["pred_pose_params"],
        init_hand_trans=trans if for_mlp else torch.cat([trans, w1], dim=1).reshape(B, 1, 4),     # (B,3) in the MLP dataset, mlp_dataset.py:179
        init_joints_2d=torch.cat([preds["joints_2d"], one], dim=2), init_joints_3d=torch.cat([j3, one], dim=2),
        init_hand_trans_j=torch.cat([j3[:, 21] - j3[:, 0], w1], dim=1).reshape(B, 1, 4))
    if for_mlp:
        batch["img_feat"] = preds["img_feat"]
    return batch
