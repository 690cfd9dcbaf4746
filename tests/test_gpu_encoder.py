"""GPU parity of the encoder kernels (fp32 implicit GEMM on MFMA) against the CPU oracle / plain PyTorch fp32."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _report(name, got, ref, atol, rtol):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    print(f"[parity] {name}: max|err|={err.max():.3e} max|ref|={np.abs(ref).max():.3e}")
    assert np.all(err <= atol + rtol * np.abs(ref)), f"{name}: max err {err.max():.3e}"


@pytest.mark.parametrize("cfg", [
    dict(N=2, H=14, W=14, Cin=64, Cout=64, k=3, s=1, p=1),      # 3x3, M = 392 (tail tile), narrow tile
    dict(N=3, H=15, W=13, Cin=32, Cout=160, k=3, s=2, p=1),     # odd sizes, stride 2, Cout not a tile multiple
    dict(N=2, H=28, W=28, Cin=128, Cout=256, k=1, s=2, p=0),    # strided 1x1 (downsample)
    dict(N=2, H=32, W=32, Cin=3, Cout=64, k=7, s=2, p=3),       # stem-like, Cin = 3 (generic gather path)
    dict(N=3, H=30, W=34, Cin=4, Cout=64, k=7, s=2, p=3),       # the stem on a 4-channel image (CONV_C4: one pixel of one tap per 16-byte load)
    dict(N=1, H=9, W=9, Cin=4, Cout=40, k=5, s=1, p=2),         # CONV_C4, 64-row tile, taps wrapping every step (kw = 5), Cout not a tile multiple
    dict(N=1, H=6, W=5, Cin=2064, Cout=32, k=3, s=1, p=1),      # Cin beyond the zero page of the fast gather: generic path, padding still zero
    dict(N=1, H=24, W=24, Cin=32, Cout=320, k=1, s=1, p=0),     # 5 x 3 = 15 tiles: the XCD-aware tile order with a remainder (15 = 8 + 7), last row and column tiles partial
    dict(N=5, H=20, W=20, Cin=48, Cout=192, k=3, s=1, p=1),     # 16 x 2 = 32 tiles (last row tile partial, 1.5 column tiles): every XCD two row tiles, their column tiles adjacent
])
def test_conv_igemm_matches_torch(cfg):
    from ihmr_amd.networks import _Packed, conv_igemm
    g = torch.Generator().manual_seed(1)
    x = torch.randn(cfg["N"], cfg["Cin"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["Cout"], cfg["Cin"], cfg["k"], cfg["k"], generator=g) / np.sqrt(cfg["Cin"] * cfg["k"] ** 2)
    b = torch.randn(cfg["Cout"], generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=cfg["s"], padding=cfg["p"]))
    pk = _Packed(w.cuda(), b.cuda(), stride=cfg["s"], pad=cfg["p"])
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, Ho, Wo = conv_igemm(xn, pk, cfg["N"], cfg["H"], cfg["W"], ldx=cfg["Cin"], act=1)
    got = y.view(cfg["N"], Ho, Wo, cfg["Cout"]).permute(0, 3, 1, 2).cpu()
    # fp32 products summed in a different order: K <= 1152 terms of O(1/sqrt(K)) -> 1e-5 absolute is ample
    _report(f"conv {cfg}", got, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cfg", [
    dict(N=33, H=14, W=14, Cin=256, Cout=256, k=3, s=1, p=1, res=True),    # 102 tiles (last M tile partial) x 144 K steps: every tile in pieces
    dict(N=96, H=10, W=10, Cin=1024, Cout=1024, k=1, s=1, p=0, res=True),  # 600 tiles x 64 K steps: workers own whole tiles AND pieces
    dict(N=64, H=7, W=7, Cin=512, Cout=512, k=3, s=1, p=1, res=False),     # the 7 x 7 layer of a 64-image batch: 100 tiles x 288 K steps
    dict(N=40, H=28, W=28, Cin=128, Cout=128, k=3, s=2, p=1, res=False),   # stride 2, one column of tiles (62 -> below the limit: split-K path)
])
def test_conv_streamk_matches_torch(cfg):
    """The Stream-K form of the 128 x 128 tile (csrc/encoder.h: conv_streamk_kernel + conv_streamk_fixup_kernel) on shapes that
    select it (ihmr_conv_igemm: >= 64 K steps, 64..768 tiles), with residual + ReLU in both epilogues, and bit-identical on a re-run
    (the partial sums are added in a fixed order)."""
    from ihmr_amd.networks import _Packed, conv_igemm
    g = torch.Generator().manual_seed(5)
    x = torch.randn(cfg["N"], cfg["Cin"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["Cout"], cfg["Cin"], cfg["k"], cfg["k"], generator=g) / np.sqrt(cfg["Cin"] * cfg["k"] ** 2)
    b = torch.randn(cfg["Cout"], generator=g)
    ref = torch.nn.functional.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=cfg["s"], padding=cfg["p"])   # rocm fp32 conv as the reference
    pk = _Packed(w.cuda(), b.cuda(), stride=cfg["s"], pad=cfg["p"])
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = None
    if cfg["res"]:
        res = torch.randn(cfg["N"] * Ho * Wo, cfg["Cout"], generator=g).cuda()
        ref = ref + res.view(cfg["N"], Ho, Wo, cfg["Cout"]).permute(0, 3, 1, 2)
    ref = torch.relu(ref).cpu()
    y, _, _ = conv_igemm(xn, pk, cfg["N"], cfg["H"], cfg["W"], ldx=cfg["Cin"], residual=res, ldr=cfg["Cout"], act=1)
    y2, _, _ = conv_igemm(xn, pk, cfg["N"], cfg["H"], cfg["W"], ldx=cfg["Cin"], residual=res, ldr=cfg["Cout"], act=1)
    assert torch.equal(y, y2)
    got = y.view(cfg["N"], Ho, Wo, cfg["Cout"]).permute(0, 3, 1, 2).cpu()
    _report(f"stream-K conv {cfg}", got, ref, atol=3e-5, rtol=1e-5)


def test_encoder_matches_oracle():
    from helpers import seeded_state_dict
    from ihmr_amd.networks import InterHandEncoder
    from oracle.encoder_ref import InterHandEncoderRef
    rng = np.random.RandomState(3)
    mean_params = torch.tensor(rng.normal(0, 0.2, (1, 122)), dtype=torch.float32)
    mean_params[0, 0] = 5.0
    B = 2
    ref = InterHandEncoderRef(mean_params.repeat(B, 1))
    sd = seeded_state_dict(ref, 100)
    ref.load_state_dict(sd)
    ref.eval()
    enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), mean_params.repeat(B, 1))
    assert list(enc.state_dict().keys()) == list(ref.state_dict().keys())
    enc.load_state_dict(sd)
    img = torch.tensor(np.random.RandomState(7).uniform(-1, 1, (B, 3, 224, 224)), dtype=torch.float32)
    torch.set_num_threads(8)
    with torch.no_grad():
        p_ref, h_ref = ref(img)
        f_ref = ref.main_encoder(img)
    enc = enc.cuda()
    p, h = enc(img.cuda())
    torch.cuda.synchronize()
    # 53 fp32 layers with different summation order and folded BN: 1e-4 relative to the feature scale
    _report("main_feat", enc.main_feat.cpu(), f_ref, atol=1e-4 * float(f_ref.abs().max()), rtol=1e-4)
    _report("pred_params", p.cpu(), p_ref, atol=1e-4, rtol=1e-4)
    _report("hand_class", h.cpu(), h_ref, atol=1e-5, rtol=0)


def test_encoder_matches_reference_golden():
    """Same seeded weights / image as tests/golden/encoder.npz (outputs of the reference's own InterHandEncoder)."""
    import os
    from helpers import seeded_state_dict
    from ihmr_amd.networks import InterHandEncoder
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder.npz")))
    mean_params = torch.tensor(g["mean_params"])
    enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), mean_params.repeat(2, 1))
    enc.load_state_dict(seeded_state_dict(enc, 100))
    img = torch.tensor(np.random.RandomState(7).uniform(-1, 1, (2, 3, 224, 224)), dtype=torch.float32)
    p, h = enc.cuda()(img.cuda())
    _report("golden params", p.cpu(), g["params"], atol=1e-4, rtol=1e-4)
    _report("golden hand_class", h.cpu(), g["hand_class"], atol=1e-5, rtol=0)
    _report("golden main_feat", enc.main_feat.cpu(), g["main_feat"], atol=1e-4 * float(np.abs(g["main_feat"]).max()), rtol=1e-4)


def test_mlp_head_matches_reference_golden():
    import os
    from helpers import seeded_state_dict
    from ihmr_amd.networks import InterHandSubNetwork
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "mlp_head.npz")))
    for k in (3, 90):
        net = InterHandSubNetwork(None, 1146, k)
        net.load_state_dict(seeded_state_dict(net, 500 + k))
        y = net.cuda()(torch.tensor(g[f"x_{k}"]).cuda())
        _report(f"mlp head {k}", y.cpu(), g[f"y_{k}"], atol=1e-6, rtol=1e-4)
