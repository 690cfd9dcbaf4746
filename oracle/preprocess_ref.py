"""ORACLE (test infrastructure) -- CPU restatement of the reference's image preprocessing for the encoder:

    DataProcessor.padding_and_resize      src/data/data_preprocess.py:45-60
    DataProcessor.normalize_joints_2d     src/data/data_preprocess.py:162-169
    transforms.ToTensor + Normalize(0.5)  src/data/baseline_dataset.py:41-44,202 (same in mlp_dataset.py:33-36,166
                                          and opt_dataset.py:32-35)

applied to the BGR uint8 array ``cv2.imread`` returns (baseline_dataset.py:123) -- no channel swap anywhere.

PARITY UNPINNED for the resize: ``cv2.resize(img, (new_width, new_height))`` is third-party
(``opencv-python==4.2.0.32``, docs/ihmr.yml:112), absent here and not installable.  ``resize_linear_u8`` restates the
published algorithm of OpenCV 4.2.0 ``modules/imgproc/src/resize.cpp`` for CV_8UC3 / INTER_LINEAR (the default
interpolation) as this build understands it:

  * ``cv::resize``: equal sizes -> plain copy; ``scale = 1.0 / ((double)dst / src)`` per axis; when both scales are
    exactly 2 INTER_LINEAR is replaced by the 2x2 box mean ``(a + b + c + d + 2) >> 2`` (the "area fast" switch);
    the IPP path is not taken for 8-bit linear ("doesn't match OpenCV exactly"), so the generic fixed-point
    code below is what runs;
  * coefficients (``resizeGeneric``): ``f = (float)((d + 0.5) * scale - 0.5)``, ``s = floor(f)``, ``f -= s``;
    horizontally ``s < 0 -> (s, f) = (0, 0)`` and ``s >= width - 1 -> (s, f) = (width - 1, 0)``; vertically the two
    source rows are clamped to [0, height - 1] and the weights are kept;
    weights = ``saturate_cast<short>((1 - f) * 2048)``, ``saturate_cast<short>(f * 2048)`` (float32 arithmetic,
    round half to even);
  * horizontal pass (``HResizeLinear``, int32): ``S[s] * a0 + S[s + 1] * a1`` (``S[s] * 2048`` at the right edge);
  * vertical pass (``VResizeLinear<uchar, int, short>``):
    ``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2`` -> uint8.

What IS pinned: ToTensor / Normalize follow torch's float32 arithmetic (``x / 255``, ``- 0.5``, ``/ 0.5``) and are checked
against torch itself; ``padding_and_resize``'s size arithmetic and the joint scaling are the reference's own lines,
checked against the reference's function with this resize injected as ``cv2.resize`` (tests/golden/make_golden.py,
``preprocess.npz``).
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS   # INTER_RESIZE_COEF_SCALE


def _sat_short(x: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float): round half to even, clamp to int16."""
    return np.clip(np.rint(x.astype(np.float32)), -32768, 32767).astype(np.int32)


def _axis_coeffs(dst: int, src: int, clamp_fraction: bool):
    scale = 1.0 / (float(dst) / float(src))                  # double, as hal::resize derives it from inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)         # (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_fraction:                                       # x axis: fx = 0 outside, sx clamped
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0.0
        s[hi] = src - 1
    w0 = _sat_short((np.float32(1.0) - f) * np.float32(COEF_SCALE))
    w1 = _sat_short(f * np.float32(COEF_SCALE))
    return scale, s, w0, w1


def resize_linear_u8(img: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """``cv2.resize(img, (new_width, new_height))`` for an (H, W, C) uint8 array (see the module docstring)."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    if (new_width, new_height) == (W, H):
        return img.copy()
    sx, xs, a0, a1 = _axis_coeffs(new_width, W, True)
    sy, ys, b0, b1 = _axis_coeffs(new_height, H, False)
    eps = np.finfo(np.float64).eps
    if abs(sx - 2) < eps and abs(sy - 2) < eps:              # INTER_LINEAR -> INTER_AREA (fast) for an exact 2x decimation
        s = img.astype(np.int32)
        r0, r1 = s[0:2 * new_height:2], s[1:2 * new_height:2]
        return ((r0[:, 0:2 * new_width:2] + r0[:, 1:2 * new_width:2] + r1[:, 0:2 * new_width:2]
                 + r1[:, 1:2 * new_width:2] + 2) >> 2).astype(np.uint8)
    s = img.astype(np.int32)
    x1 = np.minimum(xs + 1, W - 1)                           # weight a1 is 0 wherever xs + 1 would be outside
    rows = s[:, xs, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]     # (H, new_w, C) int32
    y0 = np.clip(ys, 0, H - 1)
    y1 = np.clip(ys + 1, 0, H - 1)
    top = (b0[:, None, None] * (rows[y0] >> 4)) >> 16
    bot = (b1[:, None, None] * (rows[y1] >> 4)) >> 16
    return ((top + bot + 2) >> 2).astype(np.uint8)


def padding_and_resize(img: np.ndarray, joints_2d: np.ndarray, final_size: int = 224):
    """data_preprocess.py:45-60 -- longer side -> final_size, zero padding at the bottom / right, joints scaled."""
    height, width = img.shape[:2]
    if height > width:
        ratio = final_size / height
        new_height = final_size
        new_width = int(ratio * width)
    else:
        ratio = final_size / width
        new_width = final_size
        new_height = int(ratio * height)
    new_img = np.zeros((final_size, final_size, 3), dtype=np.uint8)
    new_img[:new_height, :new_width, :] = resize_linear_u8(img, new_width, new_height)
    joints_2d = joints_2d.copy()
    joints_2d[:, :2] *= ratio
    return new_img, joints_2d


def normalize_joints_2d(joints_2d: np.ndarray, final_size: int = 224) -> np.ndarray:
    """data_preprocess.py:162-169."""
    out = np.copy(joints_2d)
    out[:, 0] = (joints_2d[:, 0] / final_size) * 2.0 - 1.0
    out[:, 1] = (joints_2d[:, 1] / final_size) * 2.0 - 1.0
    return out


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """transforms.ToTensor() + Normalize((0.5,)*3, (0.5,)*3): (H, W, C) uint8 -> (C, H, W) float32."""
    x = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def flip_image_joints(img: np.ndarray, joints_2d: np.ndarray):
    """Image / 2-D joint part of ``random_flip(..., do_flip=True)`` (data_preprocess.py:63-72), the test-time
    treatment of left-only samples (baseline_dataset.py:71-74)."""
    img_new = np.fliplr(img).copy()
    j = np.zeros((42, 3), dtype=np.float32)
    j[:21, :] = joints_2d[21:, :]
    j[21:, :] = joints_2d[:21, :]
    j[:, 0] = img.shape[1] - j[:, 0]
    return img_new, j


def preprocess(img: np.ndarray, joints_2d: np.ndarray, final_size: int = 224, do_flip: bool = False):
    """The chain of baseline_dataset.py:69-74,106,202 for test-time data:
    -> (img (3,S,S) float32, joints_2d (42,3) float32, padded uint8 image (S,S,3))."""
    new_img, j = padding_and_resize(img, joints_2d.astype(np.float32), final_size)
    if do_flip:
        new_img, j = flip_image_joints(new_img, j)
    return to_tensor_normalize(new_img), normalize_joints_2d(j, final_size).astype(np.float32), new_img
