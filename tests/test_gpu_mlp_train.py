"""GPU parity of the IHMR-MLP training step (ihmr_mlp_train_grad + the head's backward GEMMs + ihmr_adam_step, through the
C ABI) against the reference's own training step (tests/golden/mlp_train.npz) and the CPU oracle."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _opt(B, **kw):
    d = dict(isTrain=True, dist=False, process_rank=-1, batchSize=B, inputSize=224, input_nc=3, num_joints=42,
             total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3,
             model_root="", mean_param_file="mean_mano_params.pkl", checkpoints_dir="./checkpoints", strategy="mlp_default",
             total_epoch=1)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _close(name, got, ref, atol, rtol=0.0):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    print(f"[parity] {name}: max|err|={err.max():.3e} max|ref|={np.abs(ref).max():.3e}")
    assert np.all(err <= atol + rtol * np.abs(ref)), f"{name}: max err {err.max():.3e}"


def _close_after_adam(name, got, ref, grad_ref, lr, atol):
    """Weights after ONE Adam step: w - lr * g / (|g| + 1e-8).  Where the gradient entry is tiny the step is ill-conditioned
    (the last bits of g, i.e. the summation order of the GEMM, decide between -lr and +lr): those entries may differ by up to
    2 lr, everywhere else the tolerance is `atol` -- a wrong gradient sign or step size on a well-conditioned entry fails."""
    got, ref, g = np.asarray(got, np.float64), np.asarray(ref, np.float64), np.abs(np.asarray(grad_ref, np.float64))
    err = np.abs(got - ref)
    well = g > 1e-5 * max(g.max(), 1e-30)
    print(f"[parity] {name}: max|err|={err[well].max() if well.any() else 0.0:.3e} on {int(well.sum())} well-conditioned entries, "
          f"{err[~well].max() if (~well).any() else 0.0:.3e} on the other {int((~well).sum())}")
    assert np.all(err[well] <= atol), f"{name}: max err {err[well].max():.3e}"
    assert np.all(err[~well] <= 2.1 * lr), f"{name}: max err {err[~well].max():.3e} on near-zero gradient entries"


def _strategy():
    from ihmr_amd.strategies import make_mlp_strategy
    s = make_mlp_strategy()
    s[4]["loss_weights"]["shape_residual_loss"] = 1.0      # as the fixture was generated
    return s


def _prepare(batch, strategy):
    from ihmr_amd.mlp_model import MLPModel
    B = batch["init_cam"].shape[0]
    model = MLPModel(_opt(B))
    model.set_update_info(strategy, 10)
    with torch.no_grad():                                   # train_mlp.py:60-66
        model.set_input(batch)
        model.forward(forward_backbone=True)
        model.compute_loss()
        model.save_pred_to_prev()
    return model


def test_training_step_matches_reference_golden():
    from helpers import seeded_state_dict
    from ihmr_amd.mlp_train import TRAIN_LOSS_NAMES
    g = dict(np.load(os.path.join(GOLD, "mlp_train.npz")))
    batch = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("in_")}
    strategy = _strategy()
    model = _prepare(batch, strategy)
    names = [str(n) for n in g["loss_names"]]
    assert tuple(names[:-1]) == TRAIN_LOSS_NAMES
    for sid in range(len(strategy)):
        model.add_new_network(sid)
        net = model.sub_network_list[sid]
        net.load_state_dict(seeded_state_dict(net, 950 + sid, last_scale=0.05))
        model.trainers[sid].load_from_module()
        model.trainers[sid]._refresh_transposed_weights()
        model.set_input(batch)
        model.retrive_prev_prediction()
        model.forward()
        model.compute_loss(strategy[sid]["loss_weights"])
        err = model.get_current_errors()
        got = [err[n] for n in names[:-1]] + [err["total_loss"]]
        _close(f"stage {sid} loss terms", got, g[f"s{sid}_losses"], 2e-6, 2e-5)
        model.optimize_parameters()
        torch.cuda.synchronize()
        grads = model.trainers[sid].named_gradients()
        model.trainers[sid].sync_to_module()
        new = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        for k, gr in grads.items():
            gr = gr.cpu()
            norm = float(g[f"s{sid}_gradnorm_{k}"])
            scale = norm / np.sqrt(gr.numel())                                  # rms gradient entry
            assert abs(float(gr.double().norm()) - norm) <= 2e-4 * norm + 1e-12, (sid, k, float(gr.double().norm()), norm)
            w = new[k]
            if gr.numel() > 20000:
                gr, w = gr[::8, ::8], w[::8, ::8]
            _close(f"stage {sid} grad {k}", gr, g[f"s{sid}_grad_{k}"], 1e-3 * scale + 1e-10, 1e-3)
            # one Adam step moves every weight by about lr: a wrong gradient SIGN shows as a 2 lr = 2e-4 error
            _close_after_adam(f"stage {sid} weights after the step {k}", w, g[f"s{sid}_new_{k}"], g[f"s{sid}_grad_{k}"], strategy[sid]["lr"], 2e-5)


def test_param_gradient_matches_oracle_autograd(mano_arrays):
    """d loss / d final_params (B,122) of ihmr_mlp_train_grad vs torch autograd through the CPU oracle, every column at
    once (the golden only sees the columns a stage updates), with all train-only weights switched on."""
    import ctypes as C
    from ihmr_amd import hip
    from ihmr_amd.mlp_model import COLS
    from oracle.mlp_ref import MLPRef, PARAM_DIMS
    g = dict(np.load(os.path.join(GOLD, "mlp_train.npz")))
    batch = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("in_")}
    B = batch["init_cam"].shape[0]
    strategy = _strategy()
    w = dict(joints_2d_loss=10.0, joints_3d_loss=100.0, mano_pose_loss=10.0, mano_shape_loss=5.0, hand_trans_loss=50.0,
             shape_reg_loss=0.3, shape_residual_loss=2.0, collision_loss=1.0)
    right, left = mano_arrays
    orc = MLPRef(right, left, B, strategy, num_data=10)
    orc.set_input(batch)
    leaves = {}
    rng = np.random.RandomState(5)
    init = dict(pred_cam_params=batch["init_cam"], pred_hand_trans=batch["init_hand_trans"].reshape(B, 3),
                pred_right_orient=batch["init_pose_params"][:, :3], pred_right_pose_params=batch["init_pose_params"][:, 3:48],
                pred_left_orient=batch["init_pose_params"][:, 48:51], pred_left_pose_params=batch["init_pose_params"][:, 51:],
                pred_right_shape_params=batch["init_shape_params"][:, :10], pred_left_shape_params=batch["init_shape_params"][:, 10:])
    for n in PARAM_DIMS:
        t = (init[n].clone().float() + torch.tensor(rng.normal(0, 0.02, size=init[n].shape), dtype=torch.float32)).requires_grad_(True)
        leaves[n] = t
        setattr(orc, n, t)
    orc._gather()
    orc._forward_mano()
    terms = orc.compute_train_loss(w)
    terms["loss"].backward()
    ref = torch.zeros(B, 122)
    for n, sl in COLS.items():
        ref[:, sl] = leaves[n].grad
    final = torch.zeros(B, 122)
    for n, sl in COLS.items():
        final[:, sl] = leaves[n].detach()

    model = _prepare(batch, strategy)
    model.add_new_network(0)
    model.set_input(batch)
    model.final_params, model._stage_id = final.cuda().contiguous(), 0
    model.compute_loss(w)
    torch.cuda.synchronize()
    got = model._grad122.cpu()
    err = model.get_current_errors()
    for n in ("joints_2d_loss", "joints_3d_loss", "mano_pose_loss", "mano_shape_loss", "hand_trans_loss", "shape_reg_loss",
              "shape_residual_loss", "collision_loss"):
        _close(f"term {n}", err[n], float(terms[n].detach()), 2e-6, 2e-5)
    # the stage's columns, written straight into the head's dY operand
    _close("dY of stage 0 = trans columns", model.trainers[0].dy[3][:B, :3].cpu(), ref[:, 119:122], 2e-4 * 40)
    for n, sl in COLS.items():
        scale = float(ref[:, sl].abs().max())
        _close(f"d loss / d {n}", got[:, sl], ref[:, sl], 2e-4 * scale + 1e-7)


def test_head_backward_matches_torch():
    """The head's backward GEMMs / ReLU masks / bias sums alone, at the training batch size, vs torch autograd on the
    same weights (fp32 reference of the same op)."""
    from ihmr_amd.mlp_train import HeadTrainer
    from ihmr_amd.networks import InterHandSubNetwork
    torch.manual_seed(3)
    for B, k in ((128, 90), (64, 3), (20, 20)):
        net = InterHandSubNetwork(None, 1146, k)
        for m in net.regressor:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.weight, std=0.05)
                torch.nn.init.normal_(m.bias, std=0.05)
        x = torch.randn(B, 1146)
        dy = torch.randn(B, k)
        ref_net = torch.nn.Sequential(*[torch.nn.Linear(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else torch.nn.ReLU()
                                        for m in net.regressor])
        ref_net.load_state_dict(net.regressor.state_dict())
        y_ref = ref_net(x)
        y_ref.backward(dy)
        tr = HeadTrainer(net.cuda(), B, 1e-3, torch.device("cuda"))
        y = tr.forward(x.cuda()[:, :1024].contiguous(), x.cuda()[:, 1024:].contiguous())
        tr.backward(dy.cuda())
        torch.cuda.synchronize()
        _close(f"head forward B={B} k={k}", y.cpu(), y_ref.detach(), 1e-5, 1e-5)
        for (name, gr), p in zip(tr.named_gradients().items(), ref_net.parameters()):
            _close(f"head grad {name} B={B} k={k}", gr.cpu(), p.grad, 1e-5 * float(p.grad.abs().max()) + 1e-7, 1e-5)
        opt = torch.optim.Adam(ref_net.parameters(), lr=1e-3)
        ref_grads = [p.grad.clone() for p in ref_net.parameters()]
        opt.step()
        tr.optimizer_step()
        tr.sync_to_module()
        for (n, p), q, gr in zip(net.regressor.state_dict().items(), ref_net.state_dict().values(), ref_grads):
            _close_after_adam(f"head weights after Adam {n}", p.cpu(), q, gr, 1e-3, 5e-5)


def test_train_loop_runs_and_reduces_the_loss():
    """ihmr_amd.run_train_mlp (the train_mlp.py loop on synthetic data): two stages, the loss of every stage falls, the
    selection pass keeps a sensible share of the updates, and MLPModel.test() afterwards uses the trained weights."""
    from ihmr_amd import run_train_mlp
    log = run_train_mlp.main(["--num_samples", "64", "--batchSize", "32", "--epochs", "20", "--stages", "2"])
    assert len(log) == 2
    for row in log:
        assert row["loss_last"] < row["loss_first"], row
        assert 0 <= row["kept"] <= row["of"]


def test_mlp_checkpoint_resume_is_bit_identical(tmp_path):
    """MLPModel.save() after two steps of a stage, a fresh model + load_checkpoint(), a third step == three uninterrupted steps."""
    from helpers import seeded_state_dict
    g = dict(np.load(os.path.join(GOLD, "mlp_train.npz")))
    batch = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("in_")}
    strategy = _strategy()
    def make():
        from ihmr_amd.mlp_model import MLPModel
        m = MLPModel(_opt(batch["init_cam"].shape[0], checkpoints_dir=str(tmp_path)))
        m.set_update_info(strategy, 10)
        with torch.no_grad():
            m.set_input(batch); m.forward(forward_backbone=True); m.compute_loss(); m.save_pred_to_prev()
        m.add_new_network(0)
        net = m.sub_network_list[0]
        net.load_state_dict(seeded_state_dict(net, 950, last_scale=0.05))
        m.trainers[0].load_from_module(); m.trainers[0]._refresh_transposed_weights()
        return m
    def step(m):
        m.set_input(batch); m.retrive_prev_prediction(); m.forward(); m.compute_loss(strategy[0]["loss_weights"]); m.optimize_parameters()
    a = make()
    step(a); step(a)
    a.save("latest", 0)
    step(a)
    b = make()
    assert b.load_checkpoint("latest", 0) == "latest"
    step(b)
    torch.cuda.synchronize()
    for name in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a.trainers[0], name), getattr(b.trainers[0], name)), name
    assert a.trainers[0].step == b.trainers[0].step == 3
