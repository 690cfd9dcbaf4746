"""GPU numerics of the encoder's training-mode kernels (ihmr_bn_train_*, ihmr_conv_wgrad, the input gradient through
ihmr_conv_igemm, pooling backward) against plain PyTorch fp32 on the CPU (torch.nn.functional + autograd of the same op)."""
import types

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
                                              (8, 64, 56, 56, True, True), (5, 192, 11, 13, False, True),      # C / 4 = 48: not a power of two
                                              (2, 2048, 7, 7, True, False)])                                      # two column blocks of 256 float4s
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
    if relu and N % 2:
        T.relu_backward_(gm, y)                      # explicit masking pass ...
        dz, dgamma, dbeta = T.bn_train_backward(_nhwc(z.detach()), gm, saved, gamma.detach().cuda())
    else:                                            # ... or the mask applied inside the backward kernels
        dz, dgamma, dbeta = T.bn_train_backward(_nhwc(z.detach()), gm, saved, gamma.detach().cuda(), relu_y=y if relu else None)
    torch.cuda.synchronize()
    _close("bn dz", _nchw(dz, N, H, W), z.grad, 2e-5)
    _close("bn dgamma", dgamma, gamma.grad, 2e-5)
    _close("bn dbeta", dbeta, beta.grad, 2e-5)


CONVS = [  # N, Cin, H, W, Cout, k, stride, pad
    (4, 64, 16, 16, 64, 1, 1, 0), (2, 64, 14, 14, 64, 3, 1, 1), (2, 128, 16, 16, 128, 3, 2, 1), (2, 256, 8, 8, 512, 1, 2, 0),
    (2, 4, 32, 32, 64, 7, 2, 3), (3, 512, 7, 7, 2048, 1, 1, 0), (2, 512, 7, 7, 512, 3, 1, 1), (8, 64, 56, 56, 256, 1, 1, 0),
    (2, 16, 6, 10, 24, 3, 1, 1), (3, 64, 12, 20, 32, 3, 2, 1),
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
        if k == 3 and stride == 2 and pad == 1:          # the same gradient from the four parity phases (no zero insertion)
            dx2 = T.conv_dgrad_s2_3x3(dyg, [p.cuda() for p in T.pack_dgrad_phase_weights(w.detach())], N, H, W, Cin, Cout)
            torch.cuda.synchronize()
            _close("conv dX by parity phases", _nchw(dx2, N, H, W), x.grad, 2e-5)


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


def test_whole_encoder_train_forward_backward_matches_oracle_autograd():
    """EncoderTrainer (train-mode forward with batch statistics, backward to every parameter, one Adam step) against torch
    autograd through the CPU oracle's encoder (oracle/encoder_ref.py, pinned to the reference's InterHandEncoder by
    tests/golden/encoder.npz) in train() mode, on the same seeded weights and images."""
    from helpers import seeded_state_dict
    from ihmr_amd.encoder_train import EncoderTrainer
    from ihmr_amd.networks import InterHandEncoder
    from oracle.encoder_ref import InterHandEncoderRef
    rng = np.random.RandomState(3)
    mean_params = torch.tensor(rng.normal(0, 0.2, (1, 122)), dtype=torch.float32)
    mean_params[0, 0] = 5.0
    B = 4
    ref = InterHandEncoderRef(mean_params.repeat(B, 1))
    sd = seeded_state_dict(ref, 100)
    ref.load_state_dict(sd)
    ref.train()
    torch.set_num_threads(16)
    img = torch.tensor(rng.uniform(-1, 1, (B, 3, 224, 224)), dtype=torch.float32)
    A = torch.tensor(rng.normal(0, 1, (B, 122)), dtype=torch.float32)
    Bm = torch.tensor(rng.normal(0, 1, (B, 2)), dtype=torch.float32)
    p_ref, h_ref = ref(img)
    (p_ref * A).sum().add((h_ref * Bm).sum()).backward()

    enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), mean_params.repeat(B, 1))
    enc.load_state_dict(sd)
    tr = EncoderTrainer(enc.cuda(), B, 1e-4, torch.device("cuda"))
    p, h = tr.forward(img.cuda())
    tr.backward(A.cuda(), Bm.cuda())
    torch.cuda.synchronize()
    _close("train-mode params", p, p_ref, 2e-4)
    _close("train-mode hand type", h, h_ref, 2e-4)
    grads = tr.named_gradients()
    ref_grads = {k: v.grad for k, v in ref.named_parameters()}
    assert set(grads) == set(ref_grads), set(ref_grads) ^ set(grads)
    worst, rels = 0.0, {}
    for k, g in ref_grads.items():
        got = grads[k].cpu().double()
        rels[k] = float((got - g.double()).norm() / (g.double().norm() + 1e-30))
        worst = max(worst, rels[k])
    for k in [k for k in rels if 'layer4.2' in k or 'fc1' in k or 'feat' in k or 'regressor' in k or 'classifier' in k] + list(rels)[::8]:
        print(f"[parity] grad {k}: relative L2 error {rels[k]:.3e}")
    # Two fp32 implementations of a 50-layer ReLU network do not agree on every ReLU mask: a pre-activation within rounding
    # of zero is "on" in one and "off" in the other, one such flip among the ~1e5..1e6 elements of a layer changes the
    # gradients upstream by ~1e-3 relative, and the flips of the layers passed on the way add up -- 7e-4 right below the
    # head, 1.3e-2 three layers down, 1.6-1.9e-2 for the rest of the trunk, and 1e-5 for the head itself (no mask above
    # it).  The flip-free comparison at 1e-5 is test_bottleneck_blocks_match_torch below.
    for k, rel in rels.items():
        assert rel < (4e-2 if "main_encoder" in k and "fc1" not in k else 1e-4), f"{k}: relative gradient error {rel:.3e}"
    print(f"[parity] whole-encoder gradients: worst relative L2 error over {len(ref_grads)} parameters = {worst:.3e}")
    # Is the 1e-2 in the trunk a property of fp32 (ReLU masks flipping on pre-activations within rounding of zero) or of this
    # implementation?  The same network in float64 is the arbiter: torch's own fp32 gradients are compared with it exactly
    # as the HIP gradients are.  If the deviation were a defect of the HIP backward pass it would show up as an error
    # against fp64 well above torch-fp32's own; it does not -- both fp32 implementations sit at the same distance from the
    # float64 gradients, parameter by parameter.
    ref64 = InterHandEncoderRef(mean_params.repeat(B, 1).double()).double()
    ref64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    ref64.train()
    p64, h64 = ref64(img.double())
    (p64 * A.double()).sum().add((h64 * Bm.double()).sum()).backward()
    g64 = {k: v.grad for k, v in ref64.named_parameters()}
    e_hip, e_t32 = {}, {}
    for k in ref_grads:
        n = float(g64[k].norm()) + 1e-300
        e_hip[k] = float((grads[k].cpu().double() - g64[k]).norm()) / n
        e_t32[k] = float((ref_grads[k].double() - g64[k]).norm()) / n
    trunk = [k for k in ref_grads if "main_encoder" in k and "fc1" not in k]
    med = lambda d, ks: float(np.median([d[k] for k in ks]))
    print(f"[parity] gradients vs float64: trunk median HIP {med(e_hip, trunk):.3e} torch-fp32 {med(e_t32, trunk):.3e}; "
          f"worst HIP {max(e_hip.values()):.3e} torch-fp32 {max(e_t32.values()):.3e}")
    _close("train-mode params vs float64", p, p64.float(), 2e-4)
    for k in ref_grads:
        # parameter by parameter no worse than 3x torch-fp32's own distance from float64 (+ 1e-4: the head, where both are ~1e-6)
        assert e_hip[k] <= 3.0 * e_t32[k] + 1e-4, f"{k}: HIP vs float64 {e_hip[k]:.3e}, torch fp32 vs float64 {e_t32[k]:.3e}"
    assert med(e_hip, trunk) <= 1.5 * med(e_t32, trunk) + 1e-5
    # the same quantities as the REFERENCE's own encoder produced them (tests/golden/encoder_train.npz, same seeds)
    import os
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder_train.npz")))
    _close("train-mode params vs reference", p, torch.tensor(gold["params"]), 2e-4)
    for k in ref_grads:
        n_ref, n_got = float(gold[f"gradnorm/{k}"]), float(grads[k].double().norm())
        assert abs(n_got - n_ref) <= (4e-2 if "main_encoder" in k and "fc1" not in k else 1e-4) * n_ref + 1e-12, (k, n_got, n_ref)
    # running statistics of the first and the last BatchNorm
    tr.sync_to_module()
    _close("stem running_mean", enc.main_encoder.bn1.running_mean, ref.main_encoder.bn1.running_mean, 1e-4)
    _close("last bn running_var", enc.main_encoder.layer4[2].bn3.running_var, ref.main_encoder.layer4[2].bn3.running_var, 1e-3)
    # one Adam step on every parameter vs torch.optim.Adam
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt.step()
    tr.optimizer_step()
    tr.sync_to_module()
    new = enc.state_dict()
    for k, v in ref.state_dict().items():
        if "running" in k or "num_batches" in k:
            continue
        d = (new[k].cpu() - v).abs()
        g = ref_grads[k].abs()
        well = g > 0.2 * g.max()                        # Adam's first step is -lr * sign(g) wherever the gradient is not ~0:
        assert float(d[well].max()) < 2e-5, (k, float(d[well].max()))     # entries whose sign cannot be in doubt must agree
        assert float(d.max()) <= 2.1e-4, (k, float(d.max()))


@pytest.mark.parametrize("layer,index,H", [(1, 0, 12), (2, 0, 16), (3, 2, 6), (4, 2, 7)])
def test_bottleneck_blocks_match_torch(layer, index, H):
    """One bottleneck of the trainer (projection / stride-2 / identity variants) forward + backward on its own, against torch
    autograd through the oracle's block on the same input: at this size no ReLU mask sits within rounding of zero, so the
    agreement is at fp32 rounding level."""
    from helpers import seeded_state_dict
    from ihmr_amd.encoder_train import EncoderTrainer
    from ihmr_amd.networks import InterHandEncoder
    from oracle.encoder_ref import InterHandEncoderRef
    mean_params = torch.zeros(1, 122)
    ref = InterHandEncoderRef(mean_params)
    sd = seeded_state_dict(ref, 300 + layer)
    ref.load_state_dict(sd)
    ref.train()
    enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), mean_params)
    enc.load_state_dict(sd)
    N = 2
    tr = EncoderTrainer(enc.cuda(), N, 1e-4, torch.device("cuda"))
    blk = getattr(ref.main_encoder, f"layer{layer}")[index]
    b = tr.blocks[sum([3, 4, 6, 3][:layer - 1]) + index]
    cin = blk.conv1.weight.shape[1]
    g = torch.Generator().manual_seed(layer * 10 + index)
    x = F.relu(torch.randn(N, cin, H, H, generator=g)).requires_grad_(True)
    y_ref = blk(x)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    Ho = y_ref.shape[2]
    y, ho, wo = tr._block_forward(b, _nhwc(x.detach()), N, H, H)
    assert (ho, wo) == (Ho, Ho)
    dx = tr._block_backward(b, _nhwc(dy))
    torch.cuda.synchronize()
    _close("block output", _nchw(y, N, Ho, Ho), y_ref, 1e-5)
    _close("block dX", _nchw(dx, N, H, H), x.grad, 5e-5)
    grads = tr.named_gradients()
    prefix = f"main_encoder.layer{layer}.{index}."
    for k, p in blk.named_parameters():
        _close(f"block grad {k}", grads[prefix + k], p.grad, 5e-5)


def test_stem_and_maxpool_match_torch():
    from helpers import seeded_state_dict
    from ihmr_amd import encoder_train as T
    from ihmr_amd.networks import InterHandEncoder
    from oracle.encoder_ref import InterHandEncoderRef
    mean_params = torch.zeros(1, 122)
    ref = InterHandEncoderRef(mean_params)
    sd = seeded_state_dict(ref, 310)
    ref.load_state_dict(sd)
    ref.train()
    enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), mean_params)
    enc.load_state_dict(sd)
    N, H = 2, 32
    tr = T.EncoderTrainer(enc.cuda(), N, 1e-4, torch.device("cuda"))
    g = torch.Generator().manual_seed(9)
    img = torch.rand(N, 3, H, H, generator=g) * 2 - 1
    me = ref.main_encoder
    a = torch.relu(me.bn1(me.conv1(img)))
    y_ref = F.max_pool2d(a, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    x = torch.zeros(N, H, H, 4, device="cuda")
    x[..., :3].copy_(img.cuda().permute(0, 2, 3, 1))
    y, Hs, Ws = tr._unit_forward(tr.stem, x.reshape(N * H * H, 4), N, H, H)
    gp = T.maxpool_backward(y, _nhwc(dy), N, Hs, Ws, 64)
    tr._unit_backward(tr.stem, gp, need_dx=False, masked=False)
    torch.cuda.synchronize()
    grads = tr.named_gradients()
    _close("stem conv1.weight grad", grads["main_encoder.conv1.weight"], me.conv1.weight.grad, 5e-5)
    _close("stem bn1.weight grad", grads["main_encoder.bn1.weight"], me.bn1.weight.grad, 5e-5)
    _close("stem bn1.bias grad", grads["main_encoder.bn1.bias"], me.bn1.bias.grad, 5e-5)


def test_baseline_training_loop_reduces_the_loss():
    """ihmr_amd.run_train_baseline (the train_baseline.py loop on synthetic data): one batch of 8 seen 12 times at lr 1e-4 --
    the loss falls, and the encoder in eval mode afterwards runs on the trained weights."""
    from ihmr_amd import run_train_baseline
    log = run_train_baseline.main(["--num_samples", "8", "--batchSize", "8", "--total_epoch", "12", "--lr", "1e-4"])
    assert len(log) == 12
    assert log[-1]["loss_last"] < 0.8 * log[0]["loss_first"], (log[0], log[-1])


def test_baseline_checkpoint_resume_is_bit_identical(tmp_path):
    """save() after two training steps, a fresh model + load_checkpoint(), a third step: weights, BatchNorm statistics and Adam
    moments equal those of three uninterrupted steps bit for bit (all kernels have a fixed summation order), and the optimizer
    state is torch.optim.Adam's own format (loads into a torch optimizer over the reference-named parameters)."""
    from ihmr_amd import two_hand
    from ihmr_amd.baseline_model import InterHandModel
    from ihmr_amd.synthetic import synthetic_opt_batch
    B = 4
    def make():
        torch.manual_seed(11)
        return InterHandModel(types.SimpleNamespace(
            isTrain=True, dist=False, process_rank=-1, batchSize=B, inputSize=224, input_nc=3, num_joints=42, total_params_dim=122,
            cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", mean_param_file="mean_mano_params.pkl",
            checkpoints_dir=str(tmp_path), lr=1e-4, lr_decay_type="none", total_epoch=3))
    a = make()
    fwd = lambda p, s, t: two_hand.forward_from_packed(a.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    batch = {k: v.cuda() for k, v in synthetic_opt_batch(B, fwd, seed=77, with_image=True).items()}
    def step(m):
        m.set_input(batch); m.forward_train(); m.optimize_parameters()
    step(a); step(a)
    a.save("latest", 2)
    step(a)
    b = make()
    assert b.load_checkpoint("latest") == 2
    step(b)
    torch.cuda.synchronize()
    for name in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a.trainer.flat, name), getattr(b.trainer.flat, name)), name
    assert a.trainer.step == b.trainer.step == 3
    assert torch.equal(a.trainer.stem["run_var"], b.trainer.stem["run_var"])
    info = torch.load(str(tmp_path / "latest_info.pth"), map_location="cpu", weights_only=False)
    opt = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in a.encoder.parameters()], lr=1e-4)
    opt.load_state_dict(info["optimizer"])                 # torch accepts it: same parameter count, shapes checked on the next step
    assert len(opt.state_dict()["state"]) == len(list(a.encoder.parameters()))
