"""Host mirror of the reference's ``src/models/networks.py`` (+ ``resnet.py``) for the inference path:
``InterHandEncoder`` (ResNet-50 trunk -> fc1 -> feat_encoder -> 3 IEF iterations of ``regressor_ih`` ->
sigmoid ``hand_classifier``) and ``InterHandSubNetwork`` (the IHMR-MLP refinement head).

The ``nn`` layers below are used ONLY as parameter containers so that ``state_dict()`` has exactly the
reference's keys (``main_encoder.conv1.weight`` ... ``regressor_ih.0.bias``) and a reference checkpoint loads
with ``load_state_dict`` unchanged (``base_model.py:45-61``).  Their ``forward`` is never called: every
convolution / linear layer runs as an fp32 implicit GEMM on the matrix cores (``ihmr_conv_igemm``), with
BatchNorm (eval mode) folded into the packed weights and ReLU / residual / sigmoid fused into the epilogue.
The reference forces ``pretrained=True`` (a URL download, ``networks.py:40``); here weights come from
``load_state_dict`` or stay at their random initialisation.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip

BN_EPS = 1e-5


def _ceil(x, m):
    return (x + m - 1) // m * m


class _Packed:
    """Device-resident K-major weight [Kpad][ldw] (+ bias) of one conv / linear layer."""

    def __init__(self, weight4d: torch.Tensor, bias: torch.Tensor, stride=1, pad=0, k_extra=0):
        cout, cin, kh, kw = weight4d.shape
        K = kh * kw * (cin + k_extra)
        wk = weight4d.permute(2, 3, 1, 0)                                  # [kh][kw][cin][cout]
        if k_extra:
            wk = torch.cat([wk, wk.new_zeros(kh, kw, k_extra, cout)], dim=2)
        wk = wk.reshape(K, cout)
        ldw = _ceil(cout, 128) if cout > 64 else 64
        full = wk.new_zeros(_ceil(K, 16), ldw)
        full[:K, :cout] = wk
        self.w = full.contiguous()
        self.b = bias.contiguous()
        self.cout, self.cin, self.kh, self.kw, self.stride, self.pad, self.ldw = cout, cin + k_extra, kh, kw, stride, pad, ldw


def _fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    return conv.weight.detach() * scale[:, None, None, None], bn.bias.detach() - bn.running_mean.detach() * scale


_WS = {}


def _splitk_workspace(device):
    """128 MB of scratch per (device, stream) for the partial sums of ihmr_conv_igemm (split-K layers: ksplit x M x Cout floats;
    Stream-K layers: two 64 KB tile slots per worker, 512 workers).  Per STREAM on purpose: convolutions of two streams run
    concurrently (two instances in flight) and each needs its own scratch; a hipGraph capture runs on its own stream and so owns
    one more buffer, which lives as long as the captured graph replays into it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _WS:
        _WS[key] = torch.empty(32 * 1024 * 1024, device=device, dtype=torch.float32)
    return _WS[key]


def conv_igemm(x, pk: _Packed, N, H, W, ldx, out=None, ldy=None, residual=None, ldr=0, act=0):
    """x: device tensor holding NHWC activations (pixel stride ldx).  Returns (y, Ho, Wo)."""
    Ho = (H + 2 * pk.pad - pk.kh) // pk.stride + 1
    Wo = (W + 2 * pk.pad - pk.kw) // pk.stride + 1
    if out is None:
        out = torch.empty(N * Ho * Wo, pk.cout, device=x.device, dtype=torch.float32)
        ldy = pk.cout
    ws = _splitk_workspace(x.device)
    hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(x), hip.ptr(pk.w), hip.ptr(pk.b), None if residual is None else residual.data_ptr(),
                                        out.data_ptr(), N, H, W, pk.cin, Ho, Wo, pk.cout, pk.kh, pk.kw, pk.stride, pk.pad,
                                        ldx, pk.ldw, ldy, ldr, act, ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()), "ihmr_conv_igemm")
    return out, Ho, Wo


class _Bottleneck(nn.Module):  # parameter container, resnet.py:58-94
    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        self.stride = stride


class _ResNet50(nn.Module):  # parameter container, resnet.py:97-136 (fc1 instead of fc; num_classes ignored)
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
            layers = []
            for b in range(blocks):
                layers.append(_Bottleneck(cin, planes, stride if b == 0 else 1, project=(b == 0)))
                cin = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.fc1 = nn.Linear(2048, 1024)
        for m in self.modules():  # resnet.py:114-119
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class InterHandEncoder(nn.Module):
    def __init__(self, opt, mean_params):
        super().__init__()
        self.total_params_dim = getattr(opt, "total_params_dim", 122)
        self.mean_params = mean_params.clone().float()
        self.main_encoder = _ResNet50()
        self.feat_encoder = nn.Sequential(nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU())
        self.regressor_ih = nn.Sequential(nn.Linear(1024 + self.total_params_dim, self.total_params_dim))
        self.hand_classifier = nn.Sequential(nn.Linear(1024, 2))
        self._packed = None
        self.main_feat = None

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _pack(self, dev):
        P = {}
        me = self.main_encoder
        t = lambda x: x.to(dev)
        w, b = _fold_bn(me.conv1, me.bn1)
        P["stem"] = _Packed(t(w), t(b), stride=2, pad=3, k_extra=1)   # the image is padded to 4 channels: 16-byte gathers (csrc/encoder.h: CONV_C4)
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(me, f"layer{li}")):
                for ci, (conv, bn, st, pd) in enumerate([(blk.conv1, blk.bn1, 1, 0), (blk.conv2, blk.bn2, blk.stride, 1),
                                                        (blk.conv3, blk.bn3, 1, 0)], start=1):
                    w, b = _fold_bn(conv, bn)
                    P[f"l{li}.{bi}.c{ci}"] = _Packed(t(w), t(b), stride=st, pad=pd)
                if blk.downsample is not None:
                    w, b = _fold_bn(blk.downsample[0], blk.downsample[1])
                    P[f"l{li}.{bi}.ds"] = _Packed(t(w), t(b), stride=blk.stride, pad=0)
        lin = lambda m, k_extra=0: _Packed(t(m.weight.detach())[:, :, None, None], t(m.bias.detach()), k_extra=k_extra)
        P["fc1"] = lin(me.fc1)
        P["feat"] = lin(self.feat_encoder[1])
        P["reg"] = lin(self.regressor_ih[0], k_extra=_ceil(1024 + self.total_params_dim, 16) - (1024 + self.total_params_dim))
        P["cls"] = lin(self.hand_classifier[0])
        self._packed = P

    @torch.no_grad()
    def forward(self, main_input):
        hip.require_gpu()
        dev = main_input.device
        if self._packed is None or self._packed["stem"].w.device != dev:
            self._pack(dev)
        P = self._packed
        B, C, H, W = main_input.shape
        # NHWC, 3 -> 4 channels (layout plumbing only); channel 3 stays zero.  One staging buffer per launch stream: two forwards of one
        # module on two streams (two instances in flight share the weights) must not write the same buffer
        cache = self.__dict__.setdefault("_nhwc4", {})
        skey = torch.cuda.current_stream(dev).cuda_stream
        x = cache.get(skey)
        if x is None or x.shape[:3] != (B, H, W) or x.device != dev:
            x = cache[skey] = torch.zeros(B, H, W, 4, device=dev)
        x[..., :3].copy_(main_input.permute(0, 2, 3, 1))
        # stem: conv 7x7/2 + BN + ReLU, max-pool 3x3/2
        y, H, W = conv_igemm(x, P["stem"], B, H, W, ldx=4, act=1)
        Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        xp = torch.empty(B * Hp * Wp, 64, device=dev)
        hip.check(hip.lib().ihmr_maxpool3x3s2(hip.ptr(y), hip.ptr(xp), B, H, W, 64, Hp, Wp, hip.stream_ptr()), "ihmr_maxpool3x3s2")
        x, H, W, cin = xp, Hp, Wp, 64
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self.main_encoder, f"layer{li}")):
                k = f"l{li}.{bi}"
                y1, H1, W1 = conv_igemm(x, P[k + ".c1"], B, H, W, ldx=cin, act=1)
                y2, H2, W2 = conv_igemm(y1, P[k + ".c2"], B, H1, W1, ldx=P[k + ".c1"].cout, act=1)
                if blk.downsample is not None:
                    res, _, _ = conv_igemm(x, P[k + ".ds"], B, H, W, ldx=cin, act=0)
                else:
                    res = x
                cout = P[k + ".c3"].cout
                x, H, W = conv_igemm(y2, P[k + ".c3"], B, H2, W2, ldx=P[k + ".c2"].cout, residual=res, ldr=cout, act=1)
                cin = cout
        # AvgPool2d(7) + ReLU, fc1 + ReLU  (resnet.py:149-154)
        pooled = torch.empty(B, 2048, device=dev)
        hip.check(hip.lib().ihmr_avgpool_relu(hip.ptr(x), hip.ptr(pooled), B, H * W, 2048, 2048, hip.stream_ptr()), "ihmr_avgpool_relu")
        main_feat, _, _ = conv_igemm(pooled, P["fc1"], B, 1, 1, ldx=2048, act=1)
        self.main_feat = main_feat
        # feat_encoder = ReLU (no-op on a ReLU output), Linear, ReLU; written into the IEF input buffers [feat | params | 0]
        Kp = P["reg"].cin
        nparam = self.total_params_dim
        bufs = [torch.zeros(B, Kp, device=dev) for _ in range(2)]
        conv_igemm(main_feat, P["feat"], B, 1, 1, ldx=1024, out=bufs[0], ldy=Kp, act=1)
        bufs[1][:, :1024].copy_(bufs[0][:, :1024])
        self.feat = bufs[1][:, :1024]            # networks.py:68 -- the 1024-d image feature the MLP stages consume as `img_feat`
        mp = getattr(self, "_mean_dev", None)     # device copy made once (a host-to-device copy cannot be captured in a graph)
        if mp is None or mp.device != dev or getattr(self, "_mean_src", None) is not self.mean_params:
            mp = self._mean_dev = self.mean_params.to(dev)
            self._mean_src = self.mean_params
        bufs[0][:, 1024:1024 + nparam].copy_(mp if mp.shape[0] == B else mp[:1].expand(B, -1))
        cur = 0
        for _ in range(3):  # networks.py:71-75: params += Linear([feat | params])
            src, dst = bufs[cur], bufs[1 - cur]
            conv_igemm(src, P["reg"], B, 1, 1, ldx=Kp, out=dst[:, 1024:], ldy=Kp, residual=src[:, 1024:], ldr=Kp, act=0)
            cur = 1 - cur
        pred_params = bufs[cur][:, 1024:1024 + nparam].contiguous()
        hand_class, _, _ = conv_igemm(bufs[cur], P["cls"], B, 1, 1, ldx=Kp, act=2)
        return pred_params, hand_class


class InterHandSubNetwork(nn.Module):
    """networks.py:83-105: Linear(in,512) ReLU Linear(512,256) ReLU Linear(256,128) ReLU Linear(128,k), xavier gain 0.01."""

    def __init__(self, opt, input_dim, update_param_dim):
        super().__init__()
        fcs = [nn.Linear(input_dim, 512), nn.Linear(512, 256), nn.Linear(256, 128), nn.Linear(128, update_param_dim)]
        for fc in fcs:
            nn.init.xavier_uniform_(fc.weight, gain=0.01)
        self.regressor = nn.Sequential(fcs[0], nn.ReLU(inplace=True), fcs[1], nn.ReLU(inplace=True), fcs[2], nn.ReLU(inplace=True), fcs[3])
        self.input_dim, self.update_param_dim = input_dim, update_param_dim
        self._packed = None

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def packed(self, dev):
        """The four layers as K-major device weights (``_Packed``), built on first use and after every ``load_state_dict``."""
        if self._packed is None or self._packed[0].w.device != dev:
            kx = _ceil(self.input_dim, 16) - self.input_dim
            lins = [self.regressor[i] for i in (0, 2, 4, 6)]
            self._packed = [_Packed(l.weight.detach().to(dev)[:, :, None, None], l.bias.detach().to(dev), k_extra=(kx if i == 0 else 0))
                            for i, l in enumerate(lins)]
        return self._packed

    @torch.no_grad()
    def forward(self, inputs):
        hip.require_gpu()
        dev = inputs.device
        self.packed(dev)
        B = inputs.shape[0]
        Kp = self._packed[0].cin
        x = torch.zeros(B, Kp, device=dev)
        x[:, :self.input_dim].copy_(inputs)
        ld = Kp
        for i, pk in enumerate(self._packed):
            x, _, _ = conv_igemm(x, pk, B, 1, 1, ldx=ld, act=1 if i < 3 else 0)
            ld = pk.cout
        return x
