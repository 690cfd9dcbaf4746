"""ORACLE (test infrastructure) -- CPU restatement of the two-hand collision module the reference
imports as ``from sdf import SDFLoss`` (reference ``src/models/loss_utils.py:13``), constructs as
``SDFLoss(faces_right, faces_left, robustifier=...)`` (``:38``) and calls as
``self.sdf_loss(hand_verts, return_per_vert_loss=True, return_origin_scale_loss=True)`` on
``hand_verts`` of shape (B, 2, 778, 3) with index 0 = right hand (``:177-182``).

PARITY UNPINNED: the package is the third-party CUDA extension github.com/penincillin/SDF_ihmr at an
unspecified commit (``docs/install.md:37``); it is absent here and nothing in the reference pins its
outputs.  The algorithm restated below is that of its published parent (JiangWenPL/multiperson,
``sdf/sdf_loss.py``), specialised to two hands per sample.  Every constant is a decision of this
build and is listed in DESIGN.md:

  per hand h of sample b (all under no_grad, as in the parent):
      box = [min, max] over the 778 vertices;  centre = (min+max)/2
      scale = (1 + 0.2) * 0.5 * max_axis(max - min)
      phi_h = 32^3 voxel grid of the mesh normalised by (centre, scale)    [oracle/sdf_grid.c]
  for h in (0, 1):   q = (verts[b, 1-h] - centre_h) / scale_h              (gradient -> verts[b,1-h])
      val[b, h*778 + v] = trilinear grid_sample(phi_h, q_v), zeros padding, align_corners=False
                          (the default of the reference's pinned torch 1.6.0)
      robustifier rho (train only; None at test, ``loss_utils.py:36``): x -> (x/rho)^2/((x/rho)^2+1)
  loss[b]            = sum_v val[b, v] / 2**2        (parent: ``cur_loss.sum() / valid_people ** 2``)
  origin_scale[b, v] = val[b, v] * scale_h           (metres; the evaluator multiplies by 1000,
                                                      ``src/utils/evaluator.py:169,179``)
  halves: [0:778] = values sampled in the RIGHT hand's grid (at the left-hand vertices),
          [778:1556] = values sampled in the LEFT hand's grid (at the right-hand vertices);
          the evaluator swaps the halves on a flip (``evaluator.py:118-120``).

Returns ``(loss (B,), per_vert (B,1556), origin_scale (B,1556))`` when both flags are set (the only
combination the reference uses, which reshapes the first to (B,1), ``loss_utils.py:183``).
"""
from __future__ import annotations

import ctypes
import os
import os.path as osp
import subprocess

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

GRID = 32
SCALE_FACTOR = 0.2
NUM_HANDS = 2

_HERE = osp.dirname(osp.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = osp.join(_HERE, "liboracle_sdf.so")
        src = osp.join(_HERE, "sdf_grid.c")
        if not osp.isfile(so) or osp.getmtime(so) < osp.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_sdf.so"], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.ihmr_oracle_sdf_grid.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.ihmr_oracle_sdf_grid.restype = None
        lib.ihmr_oracle_point_tri_dist2.argtypes = [ctypes.c_void_p] * 4
        lib.ihmr_oracle_point_tri_dist2.restype = ctypes.c_float
        lib.ihmr_oracle_ray_hit_px.argtypes = [ctypes.c_void_p] * 4
        lib.ihmr_oracle_ray_hit_px.restype = ctypes.c_int
        _LIB = lib
    return _LIB


_LIB64 = None


def _lib64():
    """The float64 build of the same C source (oracle/sdf_grid_f64.c): the arbiter of the parity tests."""
    global _LIB64
    if _LIB64 is None:
        so = osp.join(_HERE, "liboracle_sdf_f64.so")
        srcs = [osp.join(_HERE, "sdf_grid.c"), osp.join(_HERE, "sdf_grid_f64.c")]
        if not osp.isfile(so) or osp.getmtime(so) < max(osp.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_sdf_f64.so"], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.ihmr_oracle_sdf_grid_f64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.ihmr_oracle_sdf_grid_f64.restype = None
        _LIB64 = lib
    return _LIB64


def sdf_grid(verts_n: torch.Tensor, faces: torch.Tensor, grid: int = GRID) -> torch.Tensor:
    """verts_n (H,V,3) normalised to [-1,1]; faces (F,3) int32 -> phi (H,G,G,G).  float32 vertices: the float32 grid (the spec the HIP
    kernels match bit for bit); float64 vertices: the same algorithm in double precision (the arbiter)."""
    f64 = verts_n.dtype == torch.float64
    dt = np.float64 if f64 else np.float32
    v = np.ascontiguousarray(verts_n.detach().cpu().numpy(), dtype=dt)
    f = np.ascontiguousarray(faces.cpu().numpy(), dtype=np.int32)
    H, V, _ = v.shape
    phi = np.empty((H, grid, grid, grid), dtype=dt)
    if f64:
        _lib64().ihmr_oracle_sdf_grid_f64(v.ctypes.data, f.ctypes.data, H, V, f.shape[0], grid, phi.ctypes.data)
    else:
        _lib().ihmr_oracle_sdf_grid(v.ctypes.data, f.ctypes.data, H, V, f.shape[0], grid, phi.ctypes.data)
    return torch.from_numpy(phi)


def hand_boxes(hand_verts: torch.Tensor, scale_factor: float = SCALE_FACTOR):
    """centre (B,2,1,3), scale (B,2,1,1) -- detached, float32, fixed evaluation order."""
    with torch.no_grad():
        bmin = hand_verts.min(dim=2)[0]
        bmax = hand_verts.max(dim=2)[0]
        centre = ((bmin + bmax) * 0.5)[:, :, None, :]
        scale = ((1.0 + scale_factor) * 0.5) * (bmax - bmin).max(dim=-1)[0][:, :, None, None]
    return centre, scale


class SDFLossRef(nn.Module):
    def __init__(self, faces_right, faces_left, robustifier=None, grid_size=GRID, align_corners=False, loss_divisor=float(NUM_HANDS ** 2),
                 swap_xz=False):
        """``align_corners`` / ``loss_divisor``: the conventions nothing in the reference pins (defaults: torch 1.6.0's
        grid_sample default; the parent project's ``/ valid_people ** 2``); ``swap_xz``: the grid tensor handed to grid_sample is
        phi[x][y][z] instead of phi[z][y][x], i.e. a query's x addresses the field's z axis; switchable to mirror ``ihmr_sdf_options``."""
        super().__init__()
        self.align_corners, self.loss_divisor, self.swap_xz = bool(align_corners), float(loss_divisor), bool(swap_xz)
        self.register_buffer("faces_right", torch.tensor(np.asarray(faces_right).astype(np.int32)))
        self.register_buffer("faces_left", torch.tensor(np.asarray(faces_left).astype(np.int32)))
        self.grid_size = grid_size
        self.robustifier = robustifier

    def forward(self, hand_verts, scale_factor=SCALE_FACTOR, return_per_vert_loss=False,
                return_origin_scale_loss=False):
        B = hand_verts.shape[0]
        assert hand_verts.shape[1] == NUM_HANDS
        centre, scale = hand_boxes(hand_verts, scale_factor)
        with torch.no_grad():
            vn = (hand_verts - centre) / scale
            phi = [sdf_grid(vn[:, 0].contiguous(), self.faces_right, self.grid_size),
                   sdf_grid(vn[:, 1].contiguous(), self.faces_left, self.grid_size)]
        per_vert, origin = [], []
        for h in (0, 1):
            q = (hand_verts[:, 1 - h] - centre[:, h]) / scale[:, h]  # (B,778,3)
            if self.swap_xz:
                q = q.flip(-1)
            val = F.grid_sample(phi[h][:, None].to(hand_verts.dtype), q.view(B, -1, 1, 1, 3), mode="bilinear",
                                padding_mode="zeros", align_corners=self.align_corners).view(B, -1)
            if self.robustifier:
                frac = (val / self.robustifier) ** 2
                val = frac / (frac + 1)
            per_vert.append(val)
            origin.append(val * scale[:, h, 0])
        per_vert = torch.cat(per_vert, dim=1)
        origin = torch.cat(origin, dim=1)
        losses = per_vert.sum(dim=1) / self.loss_divisor
        if return_per_vert_loss and return_origin_scale_loss:
            return losses, per_vert, origin
        if return_per_vert_loss:
            return losses, per_vert
        if return_origin_scale_loss:
            return losses, origin
        return losses


# names of the reference's import (``from sdf import SDFLoss, SDFLoss_Single``)
SDFLoss = SDFLossRef


class SDFLoss_Single(nn.Module):  # imported by the reference, never used (loss_utils.py:13)
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("SDFLoss_Single is never called on the reference's hot path")
