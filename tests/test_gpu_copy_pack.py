"""GPU tests of the packed-transfer entry points (csrc/copy_pack.h): `ihmr_copy_segments` -- one launch for what MLPModel.set_input /
get_pred_result move tensor by tensor in the reference (models/mlp_model.py:120-170, 702-719) -- and `ihmr_root_align_joints` (the
export's root alignment, mlp_model.py:530-531 + loss_utils.py:90-98).  Copies are exact: every comparison is bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_copy_segments_moves_flat_and_strided_operands_bit_exactly():
    from ihmr_amd import hip
    g = torch.Generator().manual_seed(3)
    B = 37
    packed = torch.zeros(B, 211, device="cuda")
    srcs = [torch.randn(B, w, generator=g).cuda() for w in (3, 45, 10, 1, 96)]
    pairs, col = [], 0
    for s in srcs:                                   # contiguous sources into column slices of one packed matrix
        pairs.append((s, packed[:, col:col + s.shape[1]]))
        col += s.shape[1] + 2                        # (gaps stay zero)
    hip.copy_segments(pairs)
    torch.cuda.synchronize()
    col = 0
    for s in srcs:
        assert torch.equal(packed[:, col:col + s.shape[1]], s)
        assert float(packed[:, col + s.shape[1]:col + s.shape[1] + 2].abs().sum()) == 0.0
        col += s.shape[1] + 2
    # column slices back out into contiguous tensors, a flat 1-D copy, and 8-byte elements (two dwords each)
    outs = [torch.empty_like(s) for s in srcs]
    flat_src, flat_dst = torch.randn(1000, generator=g).cuda(), torch.empty(1000, device="cuda")
    i64_src = torch.randint(-2 ** 62, 2 ** 62, (B, 5), generator=g).cuda()
    i64_mat = torch.zeros(B, 9, dtype=torch.int64, device="cuda")
    col, pairs = 0, [(flat_src, flat_dst), (i64_src, i64_mat[:, 2:7])]
    for s, o in zip(srcs, outs):
        pairs.append((packed[:, col:col + s.shape[1]], o))
        col += s.shape[1] + 2
    hip.copy_segments(pairs)
    torch.cuda.synchronize()
    assert torch.equal(flat_dst, flat_src) and torch.equal(i64_mat[:, 2:7], i64_src)
    assert int(i64_mat[:, :2].abs().sum()) == 0 and int(i64_mat[:, 7:].abs().sum()) == 0
    for s, o in zip(srcs, outs):
        assert torch.equal(o, s)


def test_copy_segments_splits_tables_longer_than_the_kernel_argument():
    from ihmr_amd import hip
    n = 2 * hip.COPY_MAX_SEGS + 5                    # three launches
    src = [torch.full((7 + i,), float(i), device="cuda") for i in range(n)]
    dst = [torch.empty_like(s) for s in src]
    hip.copy_segments(list(zip(src, dst)))
    torch.cuda.synchronize()
    for s, d in zip(src, dst):
        assert torch.equal(s, d)


def test_copy_segments_refuses_malformed_tables():
    from ihmr_amd import hip
    L = hip.lib()
    a, b = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    ok = hip.CopySeg(a.data_ptr(), b.data_ptr(), 1, 64, 64, 64)
    one = (hip.CopySeg * 1)(ok)
    assert L.ihmr_copy_segments(one, 1, None) == 0
    assert L.ihmr_copy_segments(one, 0, None) != 0 and L.ihmr_copy_segments(None, 1, None) != 0
    too_many = (hip.CopySeg * (hip.COPY_MAX_SEGS + 1))(*([ok] * (hip.COPY_MAX_SEGS + 1)))
    assert L.ihmr_copy_segments(too_many, hip.COPY_MAX_SEGS + 1, None) != 0
    for bad in (hip.CopySeg(a.data_ptr() + 2, b.data_ptr(), 1, 8, 8, 8),        # not dword-aligned
                hip.CopySeg(a.data_ptr(), b.data_ptr(), 2, 16, 8, 16),          # a row wider than its leading dimension
                hip.CopySeg(a.data_ptr(), None, 1, 8, 8, 8),                    # no destination
                hip.CopySeg(a.data_ptr(), b.data_ptr(), 0, 8, 8, 8)):           # no rows
        assert L.ihmr_copy_segments((hip.CopySeg * 1)(bad), 1, None) != 0
    torch.cuda.synchronize()


def test_root_align_joints_follows_the_reference_rule():
    """loss_utils.py:90-98: root = joint 0 when its weight > 0.5, joint 21 when it is < 1e-7, no alignment otherwise; weights copied."""
    from ihmr_amd import hip
    rng = np.random.default_rng(11)
    B = 9
    j = rng.standard_normal((B, 42, 4)).astype(np.float32)
    j[:, :, 3] = rng.uniform(0.0, 1.0, (B, 42)).astype(np.float32)
    j[0::3, 0, 3] = 1.0; j[1::3, 0, 3] = 0.0; j[2::3, 0, 3] = 0.3
    src = torch.tensor(j).cuda()
    out = torch.empty_like(src)
    hip.check(hip.lib().ihmr_root_align_joints(hip.ptr(src), hip.ptr(out), B, hip.stream_ptr()), "ihmr_root_align_joints")
    ref = j.copy()
    for b in range(B):
        w0 = j[b, 0, 3]
        root = 0 if w0 > 0.5 else (21 if w0 < 1e-7 else None)
        if root is not None:
            ref[b, :, :3] = j[b, :, :3] - j[b, root, :3]
    assert np.array_equal(out.cpu().numpy(), ref)
    assert hip.lib().ihmr_root_align_joints(None, hip.ptr(out), B, None) != 0 and hip.lib().ihmr_root_align_joints(hip.ptr(src), hip.ptr(out), 0, None) != 0
