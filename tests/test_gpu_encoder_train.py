"""GPU numerics of the encoder's training-mode kernels (ihmr_bn_train_*, ihmr_conv_wgrad, the input gradient through
ihmr_conv_igemm, pooling backward) against plain PyTorch fp32 on the CPU (torch.nn.functional + autograd of the same op)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(name, got, ref, rtol):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err, scale = float((got - ref).abs().max()), float(ref.abs().max())
    print(f"[parity] {name}: max|err|={err:.3e} max|ref|={scale:.3e}")
    assert err <= rtol * scale + 1e-7, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def _nhwc(t):            # (N,C,H,W) -> [N*H*W, C] on the GPU
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()


def _nchw(m, N, H, W):   # [N*H*W, C] -> (N,C,H,W) on the CPU
    return m.cpu().reshape(N, H, W, -1).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("N,C,H,W,res,relu", [(4, 64, 14, 14, False, True), (2, 256, 9, 7, True, True), (3, 8, 5, 5, False, False),
                                              (8, 64, 56, 56, True, True)])
def test_batchnorm_train_forward_backward(N, C, H, W, res, relu):
    from ihmr_amd import encoder_train as T
    g = torch.Generator().manual_seed(N * 1000 + C)
    z = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    r = torch.randn(N, C, H, W, generator=g) if res else None
    dy = torch.randn(N, C, H, W, generator=g)
    y_ref = F.batch_norm(z, None, None, gamma, beta, training=True, eps=1e-5)
    if res:
        y_ref = y_ref + r
    if relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(dy)
    y, saved = T.bn_train_forward(_nhwc(z.detach()), gamma.detach().cuda(), beta.detach().cuda(), _nhwc(r) if res else None, relu)
    _close("bn forward", _nchw(y, N, H, W), y_ref, 2e-6)
    _close("bn batch mean", saved[0], z.detach().mean(dim=(0, 2, 3)), 2e-6)
    _close("bn batch var (biased)", saved[1], z.detach().var(dim=(0, 2, 3), unbiased=False), 5e-6)
    gm = _nhwc(dy)
    if relu:
        T.relu_backward_(gm, y)
    dz, dgamma, dbeta = T.bn_train_backward(_nhwc(z.detach()), gm, saved, gamma.detach().cuda())
    torch.cuda.synchronize()
    _close("bn dz", _nchw(dz, N, H, W), z.grad, 2e-5)
    _close("bn dgamma", dgamma, gamma.grad, 2e-5)
    _close("bn dbeta", dbeta, beta.grad, 2e-5)


CONVS = [  # N, Cin, H, W, Cout, k, stride, pad
    (4, 64, 16, 16, 64, 1, 1, 0), (2, 64, 14, 14, 64, 3, 1, 1), (2, 128, 16, 16, 128, 3, 2, 1), (2, 256, 8, 8, 512, 1, 2, 0),
    (2, 4, 32, 32, 64, 7, 2, 3), (3, 512, 7, 7, 2048, 1, 1, 0), (2, 512, 7, 7, 512, 3, 1, 1), (8, 64, 56, 56, 256, 1, 1, 0),
    (2, 16, 6, 10, 24, 3, 1, 1),
]


@pytest.mark.parametrize("N,Cin,H,W,Cout,k,stride,pad", CONVS)
def test_conv_weight_and_input_gradients(N, Cin, H, W, Cout, k, stride, pad):
    from ihmr_amd import encoder_train as T
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, stride, pad)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    Ho, Wo = y_ref.shape[2:]
    xg, dyg = _nhwc(x.detach()), _nhwc(dy)
    if dyg.shape[1] % 4:
        pytest.skip("dY rows must be 16-byte aligned")
    y, ho, wo = T.conv_forward(xg, T.pack_forward_weight(w.detach()).cuda(), N, H, W, Cin, Cout, k, stride, pad)
    assert (ho, wo) == (Ho, Wo)
    _close("conv forward", _nchw(y, N, Ho, Wo), y_ref, 1e-5)
    dw = T.conv_wgrad(xg, dyg, N, H, W, Cin, Cout, k, stride, pad)
    torch.cuda.synchronize()
    _close("conv dW", T.unpack_wgrad(dw, w.shape), w.grad, 2e-5)
    assert float(dw[k * k * Cin:].abs().max() if dw.shape[0] > k * k * Cin else 0.0) == 0.0 and float(dw[:, Cout:].abs().sum()) == 0.0
    if Cout % 16 == 0 and (stride == 1 or (H % 2 == 0 and W % 2 == 0)):
        dx = T.conv_dgrad(dyg, T.pack_dgrad_weight(w.detach()).cuda(), N, H, W, Cin, Cout, k, stride, pad)
        torch.cuda.synchronize()
        _close("conv dX", _nchw(dx, N, H, W), x.grad, 2e-5)


def test_pooling_backward():
    from ihmr_amd import encoder_train as T
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 3, 64, 12, 10
    x = F.relu(torch.randn(N, C, H, W, generator=g)).requires_grad_(True)       # many exact ties at 0, as after a ReLU
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = T.maxpool_backward(_nhwc(x.detach()), _nhwc(dy), N, H, W, C)
    torch.cuda.synchronize()
    _close("max-pool backward", _nchw(dx, N, H, W), x.grad, 1e-6)
    N, C = 4, 2048
    x = torch.randn(N, C, 7, 7, generator=g).requires_grad_(True)
    y = F.relu(F.avg_pool2d(x, 7).flatten(1))
    dy = torch.randn(N, C, generator=g)
    y.backward(dy)
    dx = T.avgpool_relu_backward(y.detach().cuda().contiguous(), dy.cuda().contiguous(), N, 49, C)
    torch.cuda.synchronize()
    _close("avg-pool + ReLU backward", _nchw(dx, N, 7, 7), x.grad, 1e-6)
