"""Training-mode building blocks of the image encoder on the HIP path (groundwork for
``src/train_baseline.py``): BatchNorm2d with batch statistics, the weight / input gradients of a convolution, and the
pooling backward passes, as thin wrappers over the C ABI (``ihmr_bn_train_*``, ``ihmr_conv_wgrad``, ``ihmr_conv_igemm`` with
the flipped filter, ``ihmr_dilate2``, ``ihmr_maxpool3x3s2_backward``, ``ihmr_avgpool_relu_backward``).  Activations are
NHWC matrices ``[N*H*W, C]`` as everywhere in :mod:`ihmr_amd.networks`.  No CPU fallback.  :class:`EncoderTrainer` (below)
assembles them into the whole ``InterHandEncoder`` in train mode: forward, backward to every parameter, Adam, checkpointing.

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


def conv_forward(x, w_packed, N, H, W, Cin, Cout, k, stride, pad, out=None, residual=None):
    """Plain convolution (no bias, no activation): x [N*H*W, Cin] -> [N*Ho*Wo, Cout] (+ ``residual`` [N*Ho*Wo, Cout], added in the
    kernel's epilogue: the sum of two gradient branches without an extra pass over the tensor)."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty(N * Ho * Wo, Cout, device=x.device)
    ws = _splitk_workspace(x.device)
    hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(x), hip.ptr(w_packed), None, hip.ptr(residual), hip.ptr(out), N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad,
                                        x.shape[1], w_packed.shape[1], out.shape[1], 0 if residual is None else residual.shape[1], 0,
                                        ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()),
              "ihmr_conv_igemm")
    return out, Ho, Wo


def pack_dgrad_phase_weights(weight: torch.Tensor):
    """(Cout, Cin, 3, 3) filter of a stride-2, padding-1 convolution -> the four sub-filters of its input gradient, one per
    output parity (pi, pj): an even output row gets filter row 1 alone, an odd one rows 2 and 0 (dY rows io and io + 1); each
    [ceil16(kh'*kw'*Cout)][ldw(Cin)] K-major with k = (fh', fw', cout), what ``ihmr_conv_igemm`` reads."""
    cout, cin = weight.shape[:2]
    taps = ([1], [2, 0])
    out = []
    for pi in (0, 1):
        for pj in (0, 1):
            sub = weight[:, :, taps[pi], :][:, :, :, taps[pj]]                       # (cout, cin, kh', kw')
            K = sub.shape[2] * sub.shape[3] * cout
            full = weight.new_zeros(_ceil(K, 16), _ldw(cin))
            full[:K, :cin] = sub.permute(2, 3, 0, 1).reshape(K, cin)
            out.append(full.contiguous())
    return out


def conv_dgrad_s2_3x3(dy, phase_weights, N, H, W, Cin, Cout):
    """Input gradient of a 3x3 / stride 2 / padding 1 convolution as four stride-1 convolutions of dY (one per output parity)
    + one interleave: the work of the forward convolution, not four times it (no zero-inserted dY)."""
    Ho, Wo = H // 2, W // 2
    ws = _splitk_workspace(dy.device)
    phases = []
    for i, w in enumerate(phase_weights):
        kh, kw = 1 + i // 2, 1 + i % 2
        out = torch.empty(N * Ho * Wo, Cin, device=dy.device)
        hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(dy), hip.ptr(w), None, None, hip.ptr(out), N, Ho, Wo, Cout, Ho, Wo, Cin, kh, kw, 1, 0,
                                            dy.shape[1], w.shape[1], Cin, 0, 0, ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()), "ihmr_conv_igemm")
        phases.append(out)
    dx = torch.empty(N * H * W, Cin, device=dy.device)
    hip.check(hip.lib().ihmr_interleave2(*(hip.ptr(p) for p in phases), hip.ptr(dx), N, Ho, Wo, Cin, hip.stream_ptr()), "ihmr_interleave2")
    return dx


def conv_dgrad(dy, w_dgrad, N, H, W, Cin, Cout, k, stride, pad, residual=None):
    """dy [N*Ho*Wo, Cout] -> dx [N*H*W, Cin] (H, W even for the stride-2 layers, as everywhere in ResNet-50 at 224 x 224).
    ``residual`` (stride-1 layers): another gradient w.r.t. the same input, added in the epilogue."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    src, Hs, Ws = dy, Ho, Wo
    if stride == 2 and k == 1 and pad == 0:
        # strided 1x1 (the downsample branch): only the even pixels receive a gradient -- the GEMM runs on the Ho x Wo pixels
        # and its result is spread out, a quarter of the work of a convolution over the zero-inserted dY
        assert H == 2 * Ho and W == 2 * Wo
        small, _, _ = conv_forward(dy, w_dgrad, N, Ho, Wo, Cout, Cin, 1, 1, 0)
        dx = torch.empty(N * H * W, Cin, device=dy.device)
        hip.check(hip.lib().ihmr_dilate2(hip.ptr(small), hip.ptr(dx), N, Ho, Wo, Cin, hip.stream_ptr()), "ihmr_dilate2")
        return dx
    if stride == 2:
        assert H == 2 * Ho and W == 2 * Wo
        src = torch.empty(N * H * W, Cout, device=dy.device)
        hip.check(hip.lib().ihmr_dilate2(hip.ptr(dy), hip.ptr(src), N, Ho, Wo, Cout, hip.stream_ptr()), "ihmr_dilate2")
        Hs, Ws = H, W
    else:
        assert stride == 1
    dx, H2, W2 = conv_forward(src, w_dgrad, N, Hs, Ws, Cout, Cin, k, 1, k - 1 - pad, residual=residual if stride == 1 else None)
    assert (H2, W2) == (H, W)
    return dx if (residual is None or stride == 1) else dx + residual


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


def bn_train_forward(z, gamma, beta, residual=None, relu=True, running=None, momentum=0.1):
    """nn.BatchNorm2d (training) [+ residual] [+ ReLU] on z [M, C] -> (y, saved = (mean, var, invstd)); ``running`` =
    (running_mean, running_var) updated in place."""
    M, C = z.shape
    y = torch.empty_like(z)
    mean, var, invstd = (torch.empty(C, device=z.device) for _ in range(3))
    ws = torch.empty(hip.lib().ihmr_bn_workspace_bytes(C) // 4, device=z.device)
    rm, rv = running if running is not None else (None, None)
    hip.check(hip.lib().ihmr_bn_train_forward(hip.ptr(z), M, C, hip.ptr(gamma), hip.ptr(beta), hip.ptr(residual), int(relu), BN_EPS, hip.ptr(y),
                                              hip.ptr(mean), hip.ptr(var), hip.ptr(invstd), hip.ptr(rm), hip.ptr(rv), momentum, hip.ptr(ws),
                                              hip.stream_ptr()), "ihmr_bn_train_forward")
    return y, (mean, var, invstd)


def bn_train_backward(z, g, saved, gamma, dgamma=None, dbeta=None, relu_y=None):
    """g = gradient w.r.t. the unit's output -> (dz, dgamma, dbeta); dgamma / dbeta optionally into the caller's buffers.
    ``relu_y``: the unit's output when a ReLU follows and ``g`` is not masked yet (the mask is applied on the fly)."""
    M, C = z.shape
    mean, _, invstd = saved
    dz = torch.empty_like(z)
    dgamma = torch.empty(C, device=z.device) if dgamma is None else dgamma
    dbeta = torch.empty(C, device=z.device) if dbeta is None else dbeta
    ws = torch.empty(hip.lib().ihmr_bn_workspace_bytes(C) // 4, device=z.device)
    hip.check(hip.lib().ihmr_bn_train_backward(hip.ptr(z), hip.ptr(g), M, C, hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(relu_y), hip.ptr(dz),
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


# ======================================================================================= the whole encoder in train mode
class _Flat:
    """Named views into ONE flat parameter buffer (+ gradient, + Adam moments of the same layout)."""

    def __init__(self):
        self.specs, self.n = [], 0

    def add(self, name, shape):
        size = int(torch.Size(shape).numel())
        self.specs.append((name, tuple(shape), self.n, size))
        self.n += (size + 3) // 4 * 4                       # keep every view 16-byte aligned
        return name

    def allocate(self, device):
        self.params = torch.zeros(self.n, device=device)
        self.grads = torch.zeros(self.n, device=device)
        self.exp_avg = torch.zeros(self.n, device=device)
        self.exp_avg_sq = torch.zeros(self.n, device=device)
        self.p = {n: self.params[o:o + s].view(shape) for n, shape, o, s in self.specs}
        self.g = {n: self.grads[o:o + s].view(shape) for n, shape, o, s in self.specs}


class _Linear:
    """y = act(x W^T + b) with W kept as the K-major packed matrix [kpad][ldw] inside the flat buffers."""

    def __init__(self, flat, name, in_f, out_f):
        self.name, self.in_f, self.out_f = name, in_f, out_f
        self.kpad, self.ldw = _ceil(in_f, 16), _ldw(out_f)
        flat.add(name + ".weight", (self.kpad, self.ldw))
        flat.add(name + ".bias", (self.ldw,))

    def bind(self, flat, B, device):
        self.w, self.b = flat.p[self.name + ".weight"], flat.p[self.name + ".bias"]
        self.gw, self.gb = flat.g[self.name + ".weight"], flat.g[self.name + ".bias"]
        self.B, Bp = B, _ceil(B, 16)
        self.xT = torch.zeros(self.kpad, Bp, device=device)
        self.wT = torch.zeros(_ceil(self.out_f, 16), _ldw(self.in_f), device=device)
        self.zero = torch.zeros(max(self.ldw, _ldw(self.in_f)), device=device)

    def load(self, lin):
        self.w.zero_(); self.b.zero_()
        self.w[:self.in_f, :self.out_f].copy_(lin.weight.detach().t())
        self.b[:self.out_f].copy_(lin.bias.detach())

    def store(self, lin):
        lin.weight.data.copy_(self.w[:self.in_f, :self.out_f].t())
        lin.bias.data.copy_(self.b[:self.out_f])

    def refresh(self):
        hip.check(hip.lib().ihmr_transpose(hip.ptr(self.w), hip.ptr(self.wT), self.in_f, self.out_f, self.ldw, self.wT.shape[1], hip.stream_ptr()),
                  "ihmr_transpose")

    def _gemm(self, x, ldx, M, K, w, ldw, N, bias, y, ldy, act=0, residual=None, ldr=0):
        ws = _splitk_workspace(x.device)
        hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(x), hip.ptr(w), hip.ptr(bias), None if residual is None else residual.data_ptr(), y.data_ptr(),
                                            M, 1, 1, K, 1, 1, N, 1, 1, 1, 0, ldx, ldw, ldy, ldr, act, ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()),
                  "ihmr_conv_igemm")

    def forward(self, x, out, act=0, residual=None):
        """x [B][ldx >= kpad, zero padded]; out / residual: (tensor view, row stride)."""
        y, ldy = out
        r, ldr = residual if residual is not None else (None, 0)
        self._gemm(x, x.stride(0), self.B, self.kpad, self.w, self.ldw, self.out_f, self.b, y, ldy, act, r, ldr)

    def backward(self, x, dy, accumulate=False, need_dx=True):
        """x [B][>= kpad] (the forward input), dy [ceil16(B)][ldw] zero padded -> gw / gb (added when `accumulate`), returns dx
        [ceil16(B)][ldw(in_f)] or None."""
        L, st, B = hip.lib(), hip.stream_ptr, self.B
        hip.check(L.ihmr_transpose(hip.ptr(x), hip.ptr(self.xT), B, self.kpad, x.stride(0), self.xT.shape[1], st()), "ihmr_transpose")
        gw = torch.empty_like(self.gw) if accumulate else self.gw
        gb = torch.empty_like(self.gb) if accumulate else self.gb
        if accumulate:
            gw.zero_(); gb.zero_()
        self._gemm(self.xT, self.xT.shape[1], self.kpad, B, dy, dy.shape[1], self.out_f, self.zero, gw, self.ldw)
        hip.check(L.ihmr_colsum(hip.ptr(dy), hip.ptr(gb), B, self.out_f, dy.shape[1], st()), "ihmr_colsum")
        if accumulate:
            self.gw.add_(gw); self.gb.add_(gb)
        if not need_dx:
            return None
        dx = torch.zeros(_ceil(B, 16), _ldw(self.in_f), device=x.device)
        self._gemm(dy, dy.shape[1], B, self.out_f, self.wT, self.wT.shape[1], self.in_f, self.zero, dx, dx.shape[1])
        return dx


class EncoderTrainer:
    """``InterHandEncoder`` (models/networks.py:30-80) in TRAIN mode on the HIP path: forward with batch-statistics
    BatchNorm that keeps what the backward needs, backward from (d loss / d params (B,122), d loss / d hand_type (B,2)) to the
    gradient of every parameter, ``torch.optim.Adam`` semantics on one flat buffer, data parallelism by bucketed all-reduces of
    the flat gradient that run under the backward pass (:class:`ihmr_amd.dist.OverlappedGradientReducer`)."""

    def __init__(self, encoder, batch_size, lr, device):
        hip.require_gpu()
        self.enc, self.B, self.lr, self.dev, self.step = encoder, batch_size, float(lr), device, 0
        me = encoder.main_encoder
        flat = self.flat = _Flat()
        self.units = []                                   # conv + bn units in forward order
        def unit(name, conv, bn, cin):
            cout, _, k, _ = conv.weight.shape
            u = dict(name=name, conv=conv, bn=bn, cin=cin, cout=cout, k=k, stride=conv.stride[0], pad=conv.padding[0])
            flat.add(name + ".w", (_ceil(k * k * cin, 16), _ldw(cout)))
            flat.add(name + ".gamma", (cout,)); flat.add(name + ".beta", (cout,))
            self.units.append(u)
            return u
        self.stem = unit("stem", me.conv1, me.bn1, 4)     # the 3-channel image is padded to 4 channels
        self.blocks = []
        cin = 64
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(me, f"layer{li}")):
                k = f"l{li}.{bi}"
                b = dict(c1=unit(k + ".c1", blk.conv1, blk.bn1, cin), c2=unit(k + ".c2", blk.conv2, blk.bn2, blk.conv1.weight.shape[0]),
                         c3=unit(k + ".c3", blk.conv3, blk.bn3, blk.conv2.weight.shape[0]), ds=None)
                if blk.downsample is not None:
                    b["ds"] = unit(k + ".ds", blk.downsample[0], blk.downsample[1], cin)
                self.blocks.append(b)
                cin = blk.conv3.weight.shape[0]
        self.nparam = encoder.total_params_dim
        self.fc1 = _Linear(flat, "fc1", 2048, 1024)
        self.feat = _Linear(flat, "feat", 1024, 1024)
        self.reg = _Linear(flat, "reg", 1024 + self.nparam, self.nparam)
        self.cls = _Linear(flat, "cls", 1024, 2)
        flat.allocate(device)
        for l in (self.fc1, self.feat, self.reg, self.cls):
            l.bind(flat, batch_size, device)
        self.load_from_module()
        # gradient spans in the order the backward pass completes them: the four heads together, the 16 bottlenecks last to
        # first, the stem (flat layout = forward order, so each span ends where the previous one starts)
        off = {n: o for n, _, o, _ in flat.specs}
        starts = [off["fc1.weight"]] + [off[b["c1"]["name"] + ".w"] for b in reversed(self.blocks)] + [0]
        ends = [flat.n] + starts[:-1]
        from .dist import OverlappedGradientReducer
        self.reducer = OverlappedGradientReducer(flat.grads, list(zip(starts, ends)))

    # ---- parameters <-> module
    def _lin_modules(self):
        return ((self.fc1, self.enc.main_encoder.fc1), (self.feat, self.enc.feat_encoder[1]), (self.reg, self.enc.regressor_ih[0]),
                (self.cls, self.enc.hand_classifier[0]))

    def load_from_module(self):
        for u in self.units:
            w = u["conv"].weight.detach().to(self.dev)
            if u is self.stem:
                w = torch.cat([w, w.new_zeros(w.shape[0], 1, w.shape[2], w.shape[3])], dim=1)
            self.flat.p[u["name"] + ".w"].copy_(pack_forward_weight(w))
            self.flat.p[u["name"] + ".gamma"].copy_(u["bn"].weight.detach())
            self.flat.p[u["name"] + ".beta"].copy_(u["bn"].bias.detach())
            u["run_mean"] = u["bn"].running_mean.detach().clone().to(self.dev)
            u["run_var"] = u["bn"].running_var.detach().clone().to(self.dev)
            u["tracked"] = int(u["bn"].num_batches_tracked)
        for l, m in self._lin_modules():
            l.load(m.to(self.dev))
        self._refresh_derived()

    @torch.no_grad()
    def sync_to_module(self):
        for u in self.units:
            cout, cin, k = u["cout"], u["cin"], u["k"]
            w = unpack_wgrad(self.flat.p[u["name"] + ".w"], (cout, cin, k, k))
            u["conv"].weight.data.copy_(w[:, :3] if u is self.stem else w)
            u["bn"].weight.data.copy_(self.flat.p[u["name"] + ".gamma"]); u["bn"].bias.data.copy_(self.flat.p[u["name"] + ".beta"])
            u["bn"].running_mean.data.copy_(u["run_mean"]); u["bn"].running_var.data.copy_(u["run_var"])
            u["bn"].num_batches_tracked.fill_(u["tracked"])
        for l, m in self._lin_modules():
            l.store(m)
        self.enc._packed = None

    def _refresh_derived(self):
        """Operands derived from the master weights: the flipped, transposed filters of the input gradients and the
        transposed Linear weights (layout plumbing after every optimizer step)."""
        for u in self.units:
            if u is self.stem:
                continue                                   # no gradient w.r.t. the image
            cout, cin, k = u["cout"], u["cin"], u["k"]
            w = self.flat.p[u["name"] + ".w"]
            if k == 3 and u["stride"] == 2 and u["pad"] == 1:
                u["w_phase"] = pack_dgrad_phase_weights(unpack_wgrad(w, (cout, cin, 3, 3)))
                continue
            if "w_dgrad" not in u:
                u["w_dgrad"] = torch.zeros(_ceil(k * k * cout, 16), _ldw(cin), device=self.dev)
            hip.check(hip.lib().ihmr_pack_dgrad_weight(hip.ptr(w), hip.ptr(u["w_dgrad"]), k, k, cin, cout, w.shape[1], u["w_dgrad"].shape[1],
                                                       hip.stream_ptr()), "ihmr_pack_dgrad_weight")
        for l in (self.fc1, self.feat, self.reg, self.cls):
            l.refresh()

    # ---- forward
    def _unit_forward(self, u, x, N, H, W, residual=None, relu=True):
        w = self.flat.p[u["name"] + ".w"]
        z, Ho, Wo = conv_forward(x, w, N, H, W, u["cin"], u["cout"], u["k"], u["stride"], u["pad"])
        y, saved = bn_train_forward(z, self.flat.p[u["name"] + ".gamma"], self.flat.p[u["name"] + ".beta"], residual, relu,
                                    running=(u["run_mean"], u["run_var"]))      # nn.BatchNorm2d's running statistics, momentum 0.1
        u["tracked"] += 1
        u["save"] = dict(x=x, z=z, y=y, saved=saved, N=N, H=H, W=W, relu=relu)
        return y, Ho, Wo

    def _block_forward(self, b, x, N, H, W):
        """Bottleneck (resnet.py:76-94): conv-bn-relu, conv-bn-relu, conv-bn, + identity / downsample, relu."""
        y1, H1, W1 = self._unit_forward(b["c1"], x, N, H, W)
        y2, H2, W2 = self._unit_forward(b["c2"], y1, N, H1, W1)
        res = x
        if b["ds"] is not None:
            res, _, _ = self._unit_forward(b["ds"], x, N, H, W, relu=False)
        return self._unit_forward(b["c3"], y2, N, H2, W2, residual=res, relu=True)

    def _block_backward(self, b, g):
        """g = gradient w.r.t. the block output (modified in place) -> gradient w.r.t. the block input."""
        relu_backward_(g, b["c3"]["save"]["y"])                                  # the ReLU after the residual add
        g3 = self._unit_backward(b["c3"], g)
        g2 = self._unit_backward(b["c2"], g3, masked=False)
        # the gradient of the skip branch (identity or downsample) is added in the epilogue of c1's input-gradient convolution
        # (every c1 is 1 x 1 / stride 1): g1 + skip without a pass of its own over the tensor, the same bits
        skip = self._unit_backward(b["ds"], g) if b["ds"] is not None else g
        return self._unit_backward(b["c1"], g2, masked=False, residual=skip)

    def forward(self, img):
        B, dev = self.B, self.dev
        assert img.shape[0] == B and img.shape[1] == 3
        H = W = img.shape[2]
        x = torch.zeros(B, H, W, 4, device=dev)
        x[..., :3].copy_(img.permute(0, 2, 3, 1))
        x = x.reshape(B * H * W, 4)
        y, H, W = self._unit_forward(self.stem, x, B, H, W)
        Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        xp = torch.empty(B * Hp * Wp, 64, device=dev)
        hip.check(hip.lib().ihmr_maxpool3x3s2(hip.ptr(y), hip.ptr(xp), B, H, W, 64, Hp, Wp, hip.stream_ptr()), "ihmr_maxpool3x3s2")
        self._pool = dict(x=y, H=H, W=W)
        x, H, W = xp, Hp, Wp
        for b in self.blocks:
            x, H, W = self._block_forward(b, x, B, H, W)
        self._last = dict(x=x, HW=H * W)
        pooled = torch.empty(B, 2048, device=dev)
        hip.check(hip.lib().ihmr_avgpool_relu(hip.ptr(x), hip.ptr(pooled), B, H * W, 2048, 2048, hip.stream_ptr()), "ihmr_avgpool_relu")
        self.pooled = pooled
        self.main_feat = torch.empty(B, 1024, device=dev)
        self.fc1.forward(pooled, (self.main_feat, 1024), act=1)
        Kp = self.reg.kpad
        self.ief = [torch.zeros(B, Kp, device=dev) for _ in range(4)]           # [feat | params_i | 0] for i = 0..3
        self.feat.forward(self.main_feat, (self.ief[0], Kp), act=1)
        mp = self.enc.mean_params.to(dev)
        self.ief[0][:, 1024:1024 + self.nparam].copy_(mp if mp.shape[0] == B else mp[:1].expand(B, -1))
        for i in range(3):                                                       # networks.py:71-75
            self.ief[i + 1][:, :1024].copy_(self.ief[0][:, :1024])
            self.reg.forward(self.ief[i], (self.ief[i + 1][:, 1024:], Kp), act=0, residual=(self.ief[i][:, 1024:], Kp))
        self.pred_params = self.ief[3][:, 1024:1024 + self.nparam].contiguous()
        self.hand_type = torch.empty(B, 2, device=dev)
        self.cls.forward(self.ief[0], (self.hand_type, 2), act=2)
        return self.pred_params, self.hand_type

    # ---- backward
    def _unit_backward(self, u, g, need_dx=True, masked=True, residual=None):
        """g = gradient w.r.t. the unit's output -> gradient w.r.t. its input.  masked=False: the unit's ReLU mask has not been
        applied to g yet; the BatchNorm backward kernels apply it on the fly."""
        s = u["save"]
        dz, _, _ = bn_train_backward(s["z"], g, s["saved"], self.flat.p[u["name"] + ".gamma"], self.flat.g[u["name"] + ".gamma"],
                                     self.flat.g[u["name"] + ".beta"], relu_y=None if masked or not s["relu"] else s["y"])
        conv_wgrad(s["x"], dz, s["N"], s["H"], s["W"], u["cin"], u["cout"], u["k"], u["stride"], u["pad"], out=self.flat.g[u["name"] + ".w"])
        if not need_dx:
            return None
        if "w_phase" in u:
            return conv_dgrad_s2_3x3(dz, u["w_phase"], s["N"], s["H"], s["W"], u["cin"], u["cout"])
        return conv_dgrad(dz, u["w_dgrad"], s["N"], s["H"], s["W"], u["cin"], u["cout"], u["k"], u["stride"], u["pad"], residual=residual)

    def backward(self, d_params, d_hand):
        B, dev, Kp, P = self.B, self.dev, self.reg.kpad, self.nparam
        Bp = _ceil(B, 16)
        pad = lambda t, ld: torch.cat([torch.cat([t, t.new_zeros(B, ld - t.shape[1])], 1), t.new_zeros(Bp - B, ld)], 0).contiguous()
        # hand classifier: s = sigmoid(u)
        du = d_hand * self.hand_type * (1.0 - self.hand_type)
        dfeat = self.cls.backward(self.ief[0], pad(du, self.cls.ldw))[:B, :1024].clone()
        # IEF iterations, last first; the regressor's weights are shared: gradients add up
        g = d_params.clone()
        for i in (2, 1, 0):
            dx = self.reg.backward(self.ief[i], pad(g, self.reg.ldw), accumulate=(i != 2))
            dfeat += dx[:B, :1024]
            g = g + dx[:B, 1024:1024 + P]
        # feat_encoder: ReLU, Linear, ReLU
        gf = pad(dfeat, self.feat.ldw)
        relu_backward_(gf[:B], self.ief[0])
        dmain = self.feat.backward(self.main_feat, gf)
        relu_backward_(dmain[:B], self.main_feat)
        dpool = self.fc1.backward(self.pooled, dmain)
        self.reducer.ready(0)                                                     # the heads' gradients are final
        g = avgpool_relu_backward(self.pooled, dpool[:B, :2048].contiguous(), B, self._last["HW"], 2048)
        for k, b in enumerate(reversed(self.blocks)):
            g = self._block_backward(b, g)
            self.reducer.ready(1 + k)
        gp = maxpool_backward(self._pool["x"], g, B, self._pool["H"], self._pool["W"], 64)
        self._unit_backward(self.stem, gp, need_dx=False, masked=False)
        self.reducer.ready(1 + len(self.blocks))

    # ---- optimizer (torch.optim.Adam semantics; DistributedDataParallel = one all-reduce of the flat gradient)
    def optimizer_step(self, world_size: int = 1):
        scale = self.reducer.finish()                   # the bucketed all-reduces were started during backward()
        self.step += 1
        f = self.flat
        hip.check(hip.lib().ihmr_adam_step(hip.ptr(f.params), hip.ptr(f.grads), hip.ptr(f.exp_avg), hip.ptr(f.exp_avg_sq), f.n, scale, self.lr,
                                           0.9, 0.999, 1e-8, self.step, hip.stream_ptr()), "ihmr_adam_step")
        self._refresh_derived()

    # ---- flat buffers <-> tensors under the reference's parameter names, torch layouts
    def _module_names(self, u):
        me, n = "main_encoder.", u["name"]
        if n == "stem":
            return me + "conv1.weight", me + "bn1"
        li, bi, c = n.split(".")
        base = f"{me}layer{li[1:]}.{bi}."
        return (base + "downsample.0.weight", base + "downsample.1") if c == "ds" else (base + f"conv{c[1]}.weight", base + f"bn{c[1]}")

    _HEADS = (("fc1", "main_encoder.fc1"), ("feat", "feat_encoder.1"), ("reg", "regressor_ih.0"), ("cls", "hand_classifier.0"))

    def _views(self, buf):
        return {n: buf[o:o + sz].view(shape) for n, shape, o, sz in self.flat.specs}

    def _to_named(self, buf):
        v, out = self._views(buf), {}
        for u in self.units:
            wn, bn = self._module_names(u)
            w = unpack_wgrad(v[u["name"] + ".w"], (u["cout"], u["cin"], u["k"], u["k"]))
            out[wn] = w[:, :3].contiguous() if u is self.stem else w
            out[bn + ".weight"], out[bn + ".bias"] = v[u["name"] + ".gamma"].clone(), v[u["name"] + ".beta"].clone()
        for attr, key in self._HEADS:
            l = getattr(self, attr)
            out[key + ".weight"] = v[attr + ".weight"][:l.in_f, :l.out_f].t().contiguous()
            out[key + ".bias"] = v[attr + ".bias"][:l.out_f].clone()
        return out

    def _from_named(self, buf, named):
        v = self._views(buf)
        buf.zero_()
        for u in self.units:
            wn, bn = self._module_names(u)
            w = named[wn].to(self.dev)
            if u is self.stem:
                w = torch.cat([w, w.new_zeros(w.shape[0], 1, w.shape[2], w.shape[3])], dim=1)
            v[u["name"] + ".w"].copy_(pack_forward_weight(w))
            v[u["name"] + ".gamma"].copy_(named[bn + ".weight"]); v[u["name"] + ".beta"].copy_(named[bn + ".bias"])
        for attr, key in self._HEADS:
            l = getattr(self, attr)
            v[attr + ".weight"][:l.in_f, :l.out_f].copy_(named[key + ".weight"].to(self.dev).t())
            v[attr + ".bias"][:l.out_f].copy_(named[key + ".bias"])

    def named_gradients(self):
        """Gradients under the reference's state_dict keys, torch layouts."""
        return self._to_named(self.flat.grads)

    # ---- checkpoint / resume: the state of ``torch.optim.Adam(encoder.parameters())`` in torch's own format
    # (what the reference stores under 'optimizer' in ``<label>_info.pth``, base_model.py:36-43, baseline_model.py:491-495)
    def optimizer_state_dict(self):
        names = [k for k, _ in self.enc.named_parameters()]
        m, v = self._to_named(self.flat.exp_avg), self._to_named(self.flat.exp_avg_sq)
        state = {i: dict(step=torch.tensor(float(self.step)), exp_avg=m[k].cpu(), exp_avg_sq=v[k].cpu()) for i, k in enumerate(names)} if self.step else {}
        group = dict(lr=self.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(names))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, sd):
        names = [k for k, _ in self.enc.named_parameters()]
        self.lr = float(sd["param_groups"][0]["lr"])
        if not sd["state"]:
            self.step = 0
            self.flat.exp_avg.zero_(); self.flat.exp_avg_sq.zero_()
            return
        self.step = int(float(sd["state"][0]["step"]))
        self._from_named(self.flat.exp_avg, {k: sd["state"][i]["exp_avg"] for i, k in enumerate(names)})
        self._from_named(self.flat.exp_avg_sq, {k: sd["state"][i]["exp_avg_sq"] for i, k in enumerate(names)})
