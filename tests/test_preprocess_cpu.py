"""CPU tests of the image-preprocessing oracle (oracle/preprocess_ref.py): the golden fixture produced by the
reference's own ``DataProcessor`` functions around the injected resize (tests/golden/make_golden.py:gen_preprocess),
and known-answer tests of the restated 8-bit linear resize (cv2 is absent: parity of the resize itself is unpinned)."""
import os.path as osp

import numpy as np
import torch

from oracle import preprocess_ref as P

GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), "golden", "preprocess.npz")


def test_oracle_matches_reference_golden():
    g = np.load(GOLDEN)
    for i in range(int(g["n"])):
        S, flip = [int(x) for x in g[f"size{i}"]]
        f32, j, u8 = P.preprocess(g[f"img{i}"], g[f"joints{i}"], S, bool(flip))
        assert np.array_equal(u8, g[f"u8_{i}"]), i
        assert np.array_equal(f32, g[f"f32_{i}"]), i            # bit-exact float32
        assert np.array_equal(j, g[f"jout{i}"]), i


def test_to_tensor_normalize_is_torch_arithmetic():
    u8 = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    t = torch.from_numpy(np.ascontiguousarray(u8.transpose(2, 0, 1))).float().div(255).sub_(0.5).div_(0.5)
    assert np.array_equal(P.to_tensor_normalize(u8), t.numpy())
    assert P.to_tensor_normalize(u8).min() == -1.0 and P.to_tensor_normalize(u8).max() == 1.0


def test_resize_known_answers():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(12, 18, 3)).astype(np.uint8)
    # same size: copy
    assert np.array_equal(P.resize_linear_u8(img, 18, 12), img)
    # exact 2x decimation: 2x2 box mean, rounded half up
    s = img.astype(np.int32)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(P.resize_linear_u8(img, 9, 6), box.astype(np.uint8))
    # constant image stays constant at any size (weights sum to 2048, (2048 * (c*2048 >> 4) >> 16) = 4c)
    for c in (0, 1, 127, 255):
        const = np.full((7, 11, 3), c, np.uint8)
        for (w, h) in ((5, 3), (22, 14), (11, 30), (1, 1)):
            assert (P.resize_linear_u8(const, w, h) == c).all()
    # 1x2 -> 1x4 upscale of [0, 255]: sample positions -0.25, 0.25, 0.75, 1.25 (pixel centres), edges replicate
    row = np.array([[[0] * 3, [255] * 3]], np.uint8)
    assert P.resize_linear_u8(row, 4, 1)[0, :, 0].tolist() == [0, 64, 191, 255]
    # the same along y
    assert P.resize_linear_u8(row.transpose(1, 0, 2), 1, 4)[:, 0, 0].tolist() == [0, 64, 191, 255]
    # 3 -> 2 downscale: centres at 0.25 and 1.75 of [0, 100, 200]
    r3 = np.array([[[0] * 3, [100] * 3, [200] * 3]], np.uint8)
    assert P.resize_linear_u8(r3, 2, 1)[0, :, 0].tolist() == [25, 175]


def test_padding_and_resize_geometry():
    rng = np.random.RandomState(1)
    img = rng.randint(1, 256, size=(50, 30, 3)).astype(np.uint8)       # no zero pixel: the padding is recognisable
    j = np.ones((42, 3), np.float32)
    out, j2 = P.padding_and_resize(img, j, 20)
    assert out.shape == (20, 20, 3) and (out[:, 12:] == 0).all() and (out[:, :12] > 0).all()   # 30 * 20/50 = 12
    assert np.allclose(j2[:, :2], 0.4) and (j2[:, 2] == 1).all()
    img_f, j_f = P.flip_image_joints(out, np.arange(126, dtype=np.float32).reshape(42, 3))
    assert (img_f[:, :8] == 0).all() and np.array_equal(img_f[:, 8:], out[:, :12][:, ::-1])
    assert j_f[0, 0] == 20 - 63.0 and j_f[0, 2] == 65.0 and j_f[21, 1] == 1.0
