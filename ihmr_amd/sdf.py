"""Seam B -- drop-in for ``from sdf import SDFLoss`` (reference ``src/models/loss_utils.py:13``):
``SDFLoss(faces_right, faces_left, robustifier=None)`` -> ``nn.Module`` with ``.cuda()`` and
``__call__(hand_verts (B,2,778,3), return_per_vert_loss=True, return_origin_scale_loss=True)`` ->
``(losses (B,), per_vert (B,1556), losses_origin_scale (B,1556))`` (the reference reshapes the first to
(B,1), ``loss_utils.py:181-183``), differentiable w.r.t. ``hand_verts``.

All arithmetic is in the HIP kernels behind ``ihmr_sdf_collision`` (sparse voxel SDF + trilinear
sampling, semantics in DESIGN.md "SDF arithmetic spec"); the bounding boxes are detached, so the
gradient reaches only the sampled (other-hand) vertices, as in the upstream module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip


class _SdfFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hand_verts, module):
        hip.require_gpu()
        hv = hand_verts.contiguous().float()
        B = hv.shape[0]
        dev = hv.device
        loss = torch.empty(B, device=dev)
        per_vert = torch.empty(B, 1556, device=dev)
        origin = torch.empty(B, 1556, device=dev)
        dval = torch.empty(B, 1556, 3, device=dev)
        ws = module._workspace(B, dev)
        import ctypes as C
        options = hip.SdfOptions(int(bool(module.align_corners)), float(module.loss_divisor), int(bool(module.swap_xz)))
        hip.check(hip.lib().ihmr_sdf_collision_ex(hip.ptr(module.faces_right), hip.ptr(module.faces_left), hip.ptr(hv), B,
                                                  float(module.robustifier or 0.0), C.byref(options), hip.ptr(loss), hip.ptr(per_vert),
                                                  hip.ptr(origin), hip.ptr(dval), hip.ptr(ws), hip.stream_ptr()),
                  "ihmr_sdf_collision_ex")
        ctx.loss_divisor = float(module.loss_divisor)
        # box scale per entry (origin = per_vert * scale): recover it for the backward of `origin`
        ctx.save_for_backward(dval, per_vert, origin)
        return loss, per_vert, origin

    @staticmethod
    def backward(ctx, g_loss, g_per_vert, g_origin):
        dval, per_vert, origin = ctx.saved_tensors
        B = dval.shape[0]
        coef = torch.zeros(B, 1556, device=dval.device)
        if g_loss is not None:
            coef = coef + g_loss.reshape(B, 1) / ctx.loss_divisor
        if g_per_vert is not None:
            coef = coef + g_per_vert
        if g_origin is not None:
            scale = torch.where(per_vert != 0, origin / torch.where(per_vert != 0, per_vert, torch.ones_like(per_vert)),
                                torch.zeros_like(per_vert))
            coef = coef + g_origin * scale
        g = (coef[..., None] * dval).view(B, 2, 778, 3)
        # entry (h, v) is sampled at vertex v of hand 1-h
        return torch.flip(g, dims=[1]).contiguous(), None


class SDFLoss(nn.Module):
    def __init__(self, faces_right, faces_left, robustifier=None, grid_size=32, align_corners=False, loss_divisor=4.0, swap_xz=False):
        """``align_corners`` / ``loss_divisor`` / ``swap_xz`` (axis order of the grid as ``grid_sample`` sees it): the three conventions of the (absent, unpinned) upstream module that a maintainer
        holding the real package can switch to pin this seam (``include/ihmr_hip.h: ihmr_sdf_options``)."""
        super().__init__()
        assert grid_size == 32, "the kernels are built for the 32^3 grid of the upstream module"
        # one value for the options struct AND the autograd backward (the C side reads <= 0 as "default"; the backward divides by it)
        loss_divisor = float(loss_divisor)
        if not loss_divisor > 0.0:
            raise ValueError(f"SDFLoss(loss_divisor={loss_divisor}): must be > 0 (4 = num_hands^2 of the parent project, 1 = plain sum)")
        self.align_corners, self.loss_divisor, self.swap_xz = bool(align_corners), loss_divisor, bool(swap_xz)
        self.register_buffer("faces_right", torch.tensor(np.asarray(faces_right).astype(np.int32)))
        self.register_buffer("faces_left", torch.tensor(np.asarray(faces_left).astype(np.int32)))
        self.robustifier = robustifier
        self._ws = None

    def _workspace(self, B, dev):
        need = hip.lib().ihmr_sdf_workspace_bytes(B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def forward(self, hand_verts, scale_factor=0.2, return_per_vert_loss=False, return_origin_scale_loss=False):
        assert abs(scale_factor - 0.2) < 1e-12, "box margin is fixed at the upstream default 0.2"
        if self.faces_right.device != hand_verts.device:
            self.to(hand_verts.device)
        loss, per_vert, origin = _SdfFunction.apply(hand_verts, self)
        if return_per_vert_loss and return_origin_scale_loss:
            return loss, per_vert, origin
        if return_per_vert_loss:
            return loss, per_vert
        if return_origin_scale_loss:
            return loss, origin
        return loss


class SDFLoss_Single(nn.Module):  # imported but never used by the reference (loss_utils.py:13)
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("SDFLoss_Single is not on the IHMR hot path")
