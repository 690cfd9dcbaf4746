"""Batched image preprocessing on the GPU: what the reference's DataLoader workers do per image on
the CPU before the encoder sees it --

    DataProcessor.padding_and_resize     src/data/data_preprocess.py:45-60   (cv2.resize, uint8 BGR, zero padding)
    DataProcessor.random_flip(do_flip)   src/data/data_preprocess.py:63-72   (left-only samples, baseline_dataset.py:71-74)
    DataProcessor.normalize_joints_2d    src/data/data_preprocess.py:162-169
    ToTensor + Normalize(0.5, 0.5)       src/data/baseline_dataset.py:41-44,202

as ONE kernel launch for a whole batch of differently sized images (``ihmr_preprocess_images``): the raw BGR bytes go
to the device in one copy, the (B,3,224,224) float tensor the encoder reads is produced in HBM and never exists on
the host.  No CPU fallback: without the library or a GPU the calls raise.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import hip


class DataProcessor:
    """Test-time half of the reference's ``DataProcessor`` (``data_preprocess.py:17``), batched."""

    def __init__(self, opt=None, final_size: Optional[int] = None):
        self.final_size = int(final_size if final_size is not None else getattr(opt, "inputSize", 224))

    @staticmethod
    def pack(images: Sequence[np.ndarray]):
        """(H,W,3) uint8 arrays -> one pinned byte buffer, byte offsets (B) int64, sizes (B,2) int32."""
        sizes = np.zeros((len(images), 2), np.int32)
        offsets = np.zeros((len(images),), np.int64)
        total = 0
        for i, im in enumerate(images):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("images must be (H, W, 3) uint8 arrays, as cv2.imread returns them")
            sizes[i] = im.shape[:2]
            offsets[i] = total
            total += im.size
        buf = torch.empty((max(total, 1),), dtype=torch.uint8).pin_memory()
        flat = buf.numpy()
        for i, im in enumerate(images):
            flat[offsets[i]:offsets[i] + im.size] = np.ascontiguousarray(im).reshape(-1)
        return buf, torch.from_numpy(offsets), torch.from_numpy(sizes)

    def check_sizes(self, sizes: np.ndarray):
        """cv2.resize raises on an empty destination; so does this (data_preprocess.py:47-57 size arithmetic)."""
        S = self.final_size
        for h, w in np.asarray(sizes).reshape(-1, 2).tolist():
            if h <= 0 or w <= 0:
                raise ValueError(f"empty image ({h} x {w})")
            nw, nh = (int(S / h * w), S) if h > w else (S, int(S / w * h))
            if nw < 1 or nh < 1:
                raise ValueError(f"image {h} x {w} is too thin: the resized side would be empty")

    def preprocess_packed(self, pixels: torch.Tensor, offsets: torch.Tensor, sizes: torch.Tensor,
                          joints_2d: Optional[torch.Tensor] = None, do_flip: Optional[torch.Tensor] = None,
                          return_uint8: bool = False) -> Dict[str, torch.Tensor]:
        """Device-resident inputs (bytes, offsets int64, sizes int32 (B,2), joints (B,42,3) float32, do_flip (B) uint8)."""
        hip.require_gpu()
        B, S = sizes.shape[0], self.final_size
        img = torch.empty((B, 3, S, S), dtype=torch.float32, device="cuda")
        u8 = torch.empty((B, S, S, 3), dtype=torch.uint8, device="cuda") if return_uint8 else None
        jout = torch.empty_like(joints_2d) if joints_2d is not None else None
        hip.check(hip.lib().ihmr_preprocess_images(hip.ptr(pixels), hip.ptr(offsets), hip.ptr(sizes), hip.ptr(do_flip), B, S,
                                                   hip.ptr(img), hip.ptr(u8), hip.ptr(joints_2d), hip.ptr(jout),
                                                   hip.stream_ptr()), "ihmr_preprocess_images")
        out = dict(img=img)
        if u8 is not None:
            out["img_uint8"] = u8
        if jout is not None:
            out["joints_2d"] = jout
        return out

    def __call__(self, images: Sequence[np.ndarray], joints_2d=None, hand_type_array=None, return_uint8: bool = False):
        """Host images -> the batch-dict fields the Baseline model reads (``img``, ``joints_2d``, ``do_flip``,
        ``ori_img_size``; baseline_dataset.py:123-124,212-226).  ``hand_type_array`` (B,2): left-only samples
        (``[0, 1]``) are mirrored, exactly the test-time branch of ``preprocess_data`` (baseline_dataset.py:71-74)."""
        hip.require_gpu()
        buf, offsets, sizes = self.pack(images)
        self.check_sizes(sizes.numpy())
        flip = None
        if hand_type_array is not None:
            h = np.asarray(hand_type_array, np.float32).reshape(-1, 2)
            flip = torch.from_numpy(((h[:, 0] < 0.5) & (h[:, 1] > 0.5)).astype(np.uint8))
        j = None
        if joints_2d is not None:
            j = torch.as_tensor(np.asarray(joints_2d, np.float32)).reshape(-1, 42, 3).contiguous().cuda(non_blocking=True)
        out = self.preprocess_packed(buf.cuda(non_blocking=True), offsets.cuda(non_blocking=True), sizes.cuda(non_blocking=True),
                                     j, flip.cuda(non_blocking=True) if flip is not None else None, return_uint8)
        out["do_flip"] = (flip if flip is not None else torch.zeros(len(images), dtype=torch.uint8)).to(torch.float32)
        out["ori_img_size"] = sizes.max(dim=1)[0]
        return out
