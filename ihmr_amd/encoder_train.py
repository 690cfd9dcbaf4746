"""Training-mode building blocks of the image encoder on the HIP path (groundwork for
``src/train_baseline.py``): BatchNorm2d with batch statistics, the weight / input gradients of a convolution, and the
pooling backward passes, as thin wrappers over the C ABI (``ihmr_bn_train_*``, ``ihmr_conv_wgrad``, ``ihmr_conv_igemm`` with
the flipped filter, ``ihmr_dilate2``, ``ihmr_maxpool3x3s2_backward``, ``ihmr_avgpool_relu_backward``).  Activations are
NHWC matrices ``[N*H*W, C]`` as everywhere in :mod:`ihmr_amd.networks`.  No CPU fallback.

The reference: ``models/resnet.py:58-94`` (Bottleneck: conv-bn-relu x2, conv-bn, + identity / downsample, relu),
``:138-156`` (stem, max-pool, four stages, AvgPool2d(7), ReLU, fc1, ReLU), trained through ``loss.backward()`` at
``models/baseline_model.py:341``.
"""
from __future__ import annotations

import torch

from . import hip
from .networks import _ceil, _splitk_workspace

BN_EPS = 1e-5


def _ldw(n):
    return _ceil(n, 128) if n > 64 else 64


def pack_forward_weight(weight: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> [ceil16(kh*kw*Cin)][ldw] K-major, k = (fh, fw, cin): what ``ihmr_conv_igemm`` reads."""
    cout, cin, kh, kw = weight.shape
    K = kh * kw * cin
    full = weight.new_zeros(_ceil(K, 16), _ldw(cout))
    full[:K, :cout] = weight.permute(2, 3, 1, 0).reshape(K, cout)
    return full.contiguous()


def pack_dgrad_weight(weight: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> [ceil16(kh*kw*Cout)][ldw(Cin)]: the flipped, transposed filter of the input gradient,
    k = (kh-1-fh, kw-1-fw, cout)."""
    cout, cin, kh, kw = weight.shape
    K = kh * kw * cout
    full = weight.new_zeros(_ceil(K, 16), _ldw(cin))
    full[:K, :cin] = weight.flip(2, 3).permute(2, 3, 0, 1).reshape(K, cin)
    return full.contiguous()


def unpack_wgrad(dw_packed: torch.Tensor, shape) -> torch.Tensor:
    """[K][ldw] gradient in the forward layout -> torch's (Cout, Cin, kh, kw)."""
    cout, cin, kh, kw = shape
    return dw_packed[:kh * kw * cin, :cout].reshape(kh, kw, cin, cout).permute(3, 2, 0, 1).contiguous()


def conv_forward(x, w_packed, N, H, W, Cin, Cout, k, stride, pad, out=None):
    """Plain convolution (no bias, no activation): x [N*H*W, Cin] -> [N*Ho*Wo, Cout]."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty(N * Ho * Wo, Cout, device=x.device)
    ws = _splitk_workspace(x.device)
    hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(x), hip.ptr(w_packed), None, None, hip.ptr(out), N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad,
                                        x.shape[1], w_packed.shape[1], out.shape[1], 0, 0, ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()),
              "ihmr_conv_igemm")
    return out, Ho, Wo


def conv_dgrad(dy, w_dgrad, N, H, W, Cin, Cout, k, stride, pad):
    """dy [N*Ho*Wo, Cout] -> dx [N*H*W, Cin] (H, W even for the stride-2 layers, as everywhere in ResNet-50 at 224 x 224)."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    src, Hs, Ws = dy, Ho, Wo
    if stride == 2:
        assert H == 2 * Ho and W == 2 * Wo
        src = torch.empty(N * H * W, Cout, device=dy.device)
        hip.check(hip.lib().ihmr_dilate2(hip.ptr(dy), hip.ptr(src), N, Ho, Wo, Cout, hip.stream_ptr()), "ihmr_dilate2")
        Hs, Ws = H, W
    else:
        assert stride == 1
    dx, H2, W2 = conv_forward(src, w_dgrad, N, Hs, Ws, Cout, Cin, k, 1, k - 1 - pad)
    assert (H2, W2) == (H, W)
    return dx


_WG_WS = {}


def conv_wgrad(x, dy, N, H, W, Cin, Cout, k, stride, pad, out=None):
    """x [N*H*W, Cin], dy [N*Ho*Wo, Cout] -> dW in the forward packed layout [ceil16(k*k*Cin)][ldw]."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    K = k * k * Cin
    if out is None:
        out = torch.zeros(_ceil(K, 16), _ldw(Cout), device=x.device)
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    if key not in _WG_WS:
        _WG_WS[key] = torch.empty(64 * 1024 * 1024, device=x.device)            # 256 MB of pixel-range partial sums
    ws = _WG_WS[key]
    hip.check(hip.lib().ihmr_conv_wgrad(hip.ptr(x), hip.ptr(dy), hip.ptr(out), N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, x.shape[1],
                                        dy.shape[1], out.shape[1], ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()), "ihmr_conv_wgrad")
    return out


def bn_train_forward(z, gamma, beta, residual=None, relu=True):
    """nn.BatchNorm2d (training) [+ residual] [+ ReLU] on z [M, C] -> (y, saved = (mean, var, invstd))."""
    M, C = z.shape
    y = torch.empty_like(z)
    mean, var, invstd = (torch.empty(C, device=z.device) for _ in range(3))
    ws = torch.empty(hip.lib().ihmr_bn_workspace_bytes(C) // 4, device=z.device)
    hip.check(hip.lib().ihmr_bn_train_forward(hip.ptr(z), M, C, hip.ptr(gamma), hip.ptr(beta), hip.ptr(residual), int(relu), BN_EPS, hip.ptr(y),
                                              hip.ptr(mean), hip.ptr(var), hip.ptr(invstd), hip.ptr(ws), hip.stream_ptr()), "ihmr_bn_train_forward")
    return y, (mean, var, invstd)


def bn_train_backward(z, g, saved, gamma):
    """g = gradient w.r.t. the BatchNorm output (after the ReLU mask) -> (dz, dgamma, dbeta)."""
    M, C = z.shape
    mean, _, invstd = saved
    dz = torch.empty_like(z)
    dgamma, dbeta = torch.empty(C, device=z.device), torch.empty(C, device=z.device)
    ws = torch.empty(hip.lib().ihmr_bn_workspace_bytes(C) // 4, device=z.device)
    hip.check(hip.lib().ihmr_bn_train_backward(hip.ptr(z), hip.ptr(g), M, C, hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(dz),
                                               hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws), hip.stream_ptr()), "ihmr_bn_train_backward")
    return dz, dgamma, dbeta


def relu_backward_(g, y):
    """g[r][c] = y[r][c] > 0 ? g[r][c] : 0 in place."""
    hip.check(hip.lib().ihmr_relu_backward(hip.ptr(g), hip.ptr(y), g.shape[0], g.shape[1], g.shape[1], y.shape[1], hip.stream_ptr()),
              "ihmr_relu_backward")
    return g


def maxpool_backward(x, dy, N, H, W, C):
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dx = torch.empty(N * H * W, C, device=x.device)
    hip.check(hip.lib().ihmr_maxpool3x3s2_backward(hip.ptr(x), hip.ptr(dy), hip.ptr(dx), N, H, W, C, Ho, Wo, hip.stream_ptr()),
              "ihmr_maxpool3x3s2_backward")
    return dx


def avgpool_relu_backward(y, dy, N, HW, C):
    dx = torch.empty(N * HW, C, device=y.device)
    hip.check(hip.lib().ihmr_avgpool_relu_backward(hip.ptr(y), hip.ptr(dy), hip.ptr(dx), N, HW, C, y.shape[1], hip.stream_ptr()),
              "ihmr_avgpool_relu_backward")
    return dx
