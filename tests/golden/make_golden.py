#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF
(``/root/reference/src``, imported with CPU stubs -- see ``_ref_import.py``) in the build container.

    python tests/golden/make_golden.py        # writes tests/golden/*.npz  (small, committed)

Only inputs and expected outputs are stored -- no reference source text.  The reference cannot run
its two third-party seams (``smplx`` MANO layer, ``sdf`` collision module: absent, see SURVEY.md
8(c)); there this build's CPU restatements (``oracle/mano_ref.py``, ``oracle/sdf_ref.py``) are
injected, so the fixtures pin everything AROUND those seams: loss terms and their gradients,
Rodrigues / projection, snapshot filter + select, the full ``OptimizeModel.optimize()`` trajectory
logic, the ResNet-50 encoder + IEF head, the MLP refinement head, the evaluator metrics, and the image
preprocessing around ``cv2.resize`` (third seam of that kind: ``oracle/preprocess_ref.py``).
"""
import os
import os.path as osp
import sys

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, osp.dirname(HERE))

import numpy as np
import torch

from _ref_import import import_reference, make_opt


def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_losses(ns):
    """LossUtil._* on random B=4 inputs, values + gradients (loss_utils.py)."""
    B = 4
    g = torch.Generator().manual_seed(11)
    opt = make_opt(B)
    lu = ns.loss_utils.LossUtil(opt, _mano_models(ns, B))
    out = {}
    R = lambda *s: torch.randn(*s, generator=g)
    # joints 2d
    gt2, pr2 = R(B, 42, 2), R(B, 42, 2).requires_grad_(True)
    w2 = (torch.rand(B, 42, 1, generator=g) > 0.2).float()
    l, lb = lu._joints_2d_loss(gt2, pr2, w2)
    l.backward()
    out.update(j2d_gt=gt2, j2d_pred=pr2, j2d_w=w2, j2d_loss=l, j2d_loss_batch=lb, j2d_grad=pr2.grad)
    # joints 3d with the three alignment cases: right present / absent / in-between weight
    gt3 = R(B, 42, 3) * 0.1
    pr3_in = (R(B, 42, 3) * 0.1)
    w3 = (torch.rand(B, 42, 1, generator=g) > 0.2).float()
    w3[0, 0, 0], w3[1, 0, 0], w3[2, 0, 0], w3[3, 0, 0] = 1.0, 0.0, 0.3, 1.0
    leaf = pr3_in.clone().requires_grad_(True)
    pr3 = leaf * 1.0
    gt3c = gt3.clone()
    l, lb = lu._joints_3d_loss(gt3c, pr3, w3)
    l.backward()
    out.update(j3d_gt=gt3, j3d_pred=pr3_in, j3d_w=w3, j3d_loss=l, j3d_loss_batch=lb, j3d_grad=leaf.grad,
               j3d_gt_aligned=gt3c, j3d_pred_aligned=pr3)
    # hand trans
    gtt, prt = R(B, 1, 3) * 0.05, (R(B, 1, 3) * 0.05).requires_grad_(True)
    wt = torch.tensor([1.0, 0.0, 1.0, 1.0]).view(B, 1, 1)
    l = lu._hand_trans_loss(gtt, prt, wt)
    l.backward()
    out.update(tr_gt=gtt, tr_pred=prt, tr_w=wt, tr_loss=l, tr_grad=prt.grad)
    # shape reg
    sh = R(B, 20).requires_grad_(True)
    l = lu._shape_reg_loss(sh)
    l.backward()
    out.update(sh=sh, sh_loss=l, sh_grad=sh.grad)
    # finger reg
    j = (R(B, 42, 3) * 0.05).requires_grad_(True)
    l, lb = lu._finger_reg_loss(j)
    l.backward()
    out.update(fin_j=j, fin_loss=l, fin_loss_batch=lb, fin_grad=j.grad)
    # train-only losses on the path's "next" list: values only
    mp, pmp = R(B, 48) * 0.4, R(B, 48) * 0.4
    wmp = torch.tensor([1.0, 1.0, 0.0, 1.0]).view(B, 1)
    out.update(mp=mp, pmp=pmp, wmp=wmp, mano_pose_loss=lu._mano_pose_loss(mp, pmp, wmp),
               mano_shape_loss=lu._mano_shape_loss(sh.detach()[:, :10], sh.detach()[:, 10:], wmp),
               shape_residual_loss=lu._shape_residual_loss(sh.detach(), sh.detach() * 0.5))
    ht_gt = (torch.rand(B, 2, generator=g) > 0.5).float()
    ht_pr = torch.rand(B, 2, generator=g) * 0.98 + 0.01
    out.update(ht_gt=ht_gt, ht_pred=ht_pr, ht_valid=torch.tensor([1.0, 1.0, 0.0, 1.0]).view(B, 1),
               hand_type_loss=lu._hand_type_loss(ht_gt, ht_pr, torch.tensor([1.0, 1.0, 0.0, 1.0]).view(B, 1)))
    # transforms
    rv = R(10, 3)
    rv[0] = 0.0
    cam = torch.cat([torch.rand(B, 1, generator=g) * 5 + 1, R(B, 2) * 0.1], 1)
    X = R(B, 42, 3) * 0.1
    out.update(rod_in=rv, rod_out=ns.transform_utils.batch_rodrigues(_with_device(rv)), proj_X=X, proj_cam=cam,
               proj_out=ns.transform_utils.batch_orthogonal_project(X, cam))
    np.savez_compressed(osp.join(HERE, "losses.npz"), **t2n(out))
    print("losses.npz", len(out), "arrays")


def _with_device(t):
    """reference batch_rodrigues calls ``pose_params.get_device()`` inside ``torch.cuda.device(...)``:
    on CPU get_device() is -1; make the context manager a no-op for the call."""
    import contextlib
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    return t


def _mano_models(ns, B):
    import smplx
    models = {h: smplx.create("", "mano", use_pca=False, is_rhand=(h == "right"), batch_size=2 * B) for h in ("left", "right")}
    return models


def gen_select(ns):
    """filter_by_losses + select_params on random (S=7, B=6) tables incl. ties and the no-candidate case."""
    S, B = 7, 6
    g = torch.Generator().manual_seed(5)
    j3d = torch.rand(S, B, generator=g) + 0.5
    col = torch.rand(S, B, generator=g) + 0.5
    j3d[:, 0] = j3d[0, 0] * 2            # sample 0: nothing passes the filter -> index 0
    j3d[3, 1] = j3d[5, 1] = 0.01         # sample 1: exact tie between rows 3 and 5
    col[1:, 1] = col[0, 1] * 0.5
    col[:, 2] = 0.0                      # sample 2: zero collision everywhere (0 <= 0 * 0.901 passes)
    j3d[4, 3] = j3d[0, 3] * 1.0005       # sample 3: inside the +0.1 % band
    col[4, 3] = col[0, 3] * 0.85
    params = {"pred_left_pose_params": torch.randn(S, B, 45, generator=g), "pred_right_pose_params": torch.randn(S, B, 45, generator=g)}
    out = dict(j3d=j3d.clone(), col=col.clone(), **{k: v.clone() for k, v in params.items()})
    for tag, filt, sel in (("a", [("joints_3d_loss_p", "+0"), ("collision_loss", "-10")], "joints_3d_loss_p"),
                           ("b", [("joints_3d_loss_p", "+0"), ("collision_loss", "+0")], "collision_loss")):
        losses = {"joints_3d_loss_p": j3d.clone(), "collision_loss": col.clone()}
        upd = ns.opt_utils.filter_by_losses(losses, filt)
        selp = ns.opt_utils.select_params({k: v.clone() for k, v in params.items()}, upd, sel)
        out[f"{tag}_filtered_j3d"] = upd["joints_3d_loss_p"]
        out[f"{tag}_filtered_col"] = upd["collision_loss"]
        out[f"{tag}_idx"] = torch.argmin(upd[sel], dim=0)
        for k, v in selp.items():
            out[f"{tag}_{k}"] = v
    np.savez_compressed(osp.join(HERE, "select.npz"), **t2n(out))
    print("select.npz")


def gen_opt_traj(ns):
    """Full reference OptimizeModel.optimize() on the synthetic asset: B=3, 4 stages x 4 iterations,
    snapshot every 2.  Stores the input batch, per-stage selected parameters and the final export."""
    from ihmr_amd.strategies import make_opt_strategy
    from ihmr_amd.synthetic import synthetic_opt_batch
    from oracle.opt_ref import OptimizeRef
    from ihmr_amd.assets import synthetic_mano
    B, epoch, freq = 3, 3, 2
    ns.strategies.strategies["golden_small"] = make_opt_strategy(epoch)
    ns.optimize_model.strategies["golden_small"] = ns.strategies.strategies["golden_small"]
    opt = make_opt(B, strategy="golden_small", save_mid_freq=freq)
    ref = ns.optimize_model.OptimizeModel(opt)
    helper = OptimizeRef(synthetic_mano(True), synthetic_mano(False), B, [], save_mid_freq=1)

    def fwd(pose, shape, trans):
        helper.pred_right_orient, helper.pred_left_orient = pose[:, :3], pose[:, 48:51]
        helper.pred_right_pose_params, helper.pred_left_pose_params = pose[:, 3:48], pose[:, 51:]
        helper.pred_right_shape_params, helper.pred_left_shape_params = shape[:, :10], shape[:, 10:]
        helper.pred_hand_trans = trans.view(-1, 1, 3)
        return helper.get_mano_output()[2]

    batch = synthetic_opt_batch(B, fwd, seed=2024)
    # exercise the "no right wrist" alignment branch on the last sample (weights of joint 0 = 0)
    batch["init_joints_3d"][2, 0, 3] = 0.0
    batch["joints_3d"][2, 0, 3] = 0.0
    ref.set_input(batch)
    ref.init_optimize()
    ref.optimize(0, 1)
    res = ref.get_pred_result()
    out = {f"in_{k}": v for k, v in batch.items()}
    out.update({f"out_{k}": v for k, v in res.items()})
    out["out_pred_joints_2d"] = ref.pred_joints_2d.detach()
    out["out_joints_3d_loss_p_batch"] = ref.joints_3d_loss_p_batch.detach()
    out["out_joints_2d_loss_p_batch"] = ref.joints_2d_loss_p_batch.detach()
    out["out_loss"] = ref.loss.detach()
    out["meta_epoch_freq"] = np.array([epoch, freq])
    np.savez_compressed(osp.join(HERE, "opt_traj.npz"), **t2n(out))
    print("opt_traj.npz", {k: np.asarray(v).shape for k, v in res.items() if k.startswith("pred_")})


def _ref_opt_run(ns, B, strategy, freq, batch, optimizer="adam"):
    """One pass of the reference's own OptimizeModel over `batch`; returns the dict stored in the fixture."""
    ns.strategies.strategies["golden_tmp"] = strategy
    ns.optimize_model.strategies["golden_tmp"] = strategy
    opt = make_opt(B, strategy="golden_tmp", save_mid_freq=freq)
    opt.optimizer = optimizer
    ref = ns.optimize_model.OptimizeModel(opt)
    ref.set_input(batch)
    ref.init_optimize()
    ref.optimize(0, 1)
    res = ref.get_pred_result()
    out = {f"in_{k}": v for k, v in batch.items()}
    out.update({f"out_{k}": v for k, v in res.items()})
    out["out_pred_joints_2d"] = ref.pred_joints_2d.detach()
    out["out_joints_3d_loss_p_batch"] = ref.joints_3d_loss_p_batch.detach()
    out["out_joints_2d_loss_p_batch"] = ref.joints_2d_loss_p_batch.detach()
    out["out_loss"] = ref.loss.detach()
    return out


def _synthetic_batch(B, seed):
    from ihmr_amd.synthetic import synthetic_opt_batch
    from oracle.opt_ref import OptimizeRef
    from ihmr_amd.assets import synthetic_mano
    helper = OptimizeRef(synthetic_mano(True), synthetic_mano(False), B, [], save_mid_freq=1)

    def fwd(pose, shape, trans):
        helper.pred_right_orient, helper.pred_left_orient = pose[:, :3], pose[:, 48:51]
        helper.pred_right_pose_params, helper.pred_left_pose_params = pose[:, 3:48], pose[:, 51:]
        helper.pred_right_shape_params, helper.pred_left_shape_params = shape[:, :10], shape[:, 10:]
        helper.pred_hand_trans = trans.view(-1, 1, 3)
        return helper.get_mano_output()[2]

    return synthetic_opt_batch(B, fwd, seed=seed)


def gen_opt_traj_ragged(ns):
    """The reference's OptimizeModel.optimize() on a RAGGED batch (tests/helpers.py:ragged_opt_batch: single-hand samples,
    missing / half-weight wrists, zero-weight joints, no 3-D target, separated hands): B=8, 4 stages x 4 iterations,
    snapshot every 2."""
    from helpers import ragged_opt_batch
    from ihmr_amd.strategies import make_opt_strategy
    B, epoch, freq = 8, 3, 2
    batch = ragged_opt_batch(_synthetic_batch(B, 777))
    out = _ref_opt_run(ns, B, make_opt_strategy(epoch), freq, batch)
    out["meta_epoch_freq"] = np.array([epoch, freq])
    np.savez_compressed(osp.join(HERE, "opt_traj_ragged.npz"), **t2n(out))
    print("opt_traj_ragged.npz", float(out["out_loss"]))


def gen_opt_traj_variants(ns):
    """Two more passes of the reference's OptimizeModel (B=3, 4 x 6 iterations, snapshot every 2): (a) Adam with the
    non-default filter / select criteria of `variant_strategy`, (b) `--optimizer sgd` (torch.optim.SGD, momentum 0.9,
    optimize_model.py:345-347) with the default criteria.  Sample 2 has separated hands (origin collision loss 0)."""
    from helpers import variant_strategy
    from ihmr_amd.strategies import make_opt_strategy
    B, epoch, freq = 3, 5, 2
    batch = _synthetic_batch(B, 555)
    batch["init_hand_trans"][2, 0, 0] += 0.4
    out = {}
    for tag, strategy, optimizer in (("crit", variant_strategy(epoch), "adam"), ("sgd", make_opt_strategy(epoch), "sgd")):
        r = _ref_opt_run(ns, B, strategy, freq, {k: v.clone() for k, v in batch.items()}, optimizer)
        for k, v in r.items():
            if k.startswith("in_"):
                out[k] = v
            elif "verts" not in k:
                out[f"{tag}_{k}"] = v
    out["meta_epoch_freq"] = np.array([epoch, freq])
    np.savez_compressed(osp.join(HERE, "opt_traj_variants.npz"), **t2n(out))
    print("opt_traj_variants.npz", float(out["crit_out_loss"]), float(out["sgd_out_loss"]))


from helpers import seeded_state_dict  # tests/helpers.py (shared with the tests)


def gen_encoder(ns):
    """InterHandEncoder (ResNet-50 trunk + fc + IEF regressor + hand classifier, networks.py:45-80,
    resnet.py:97-156) in eval mode on a seeded 2 x 3 x 224 x 224 image with a seeded state_dict."""
    ns.networks.get_model = lambda arch: getattr(ns.resnet, arch)(pretrained=False, num_classes=512)
    opt = make_opt(2)
    rng = np.random.RandomState(3)
    mean_params = torch.tensor(rng.normal(0, 0.2, (1, 122)), dtype=torch.float32)
    mean_params[0, 0] = 5.0
    enc = ns.networks.InterHandEncoder(opt, mean_params.repeat(2, 1))
    enc.load_state_dict(seeded_state_dict(enc, 100))
    enc.eval()
    img = torch.tensor(np.random.RandomState(7).uniform(-1, 1, (2, 3, 224, 224)), dtype=torch.float32)
    with torch.no_grad():
        feat = enc.main_encoder(img)
        params, hand_class = enc(img)
        x = enc.main_encoder.maxpool(enc.main_encoder.relu(enc.main_encoder.bn1(enc.main_encoder.conv1(img))))
        l1 = enc.main_encoder.layer1(x)
    out = dict(mean_params=mean_params, main_feat=feat, params=params, hand_class=hand_class,
               stem_sample=x[:, :, ::8, ::8], layer1_sample=l1[:, ::16, ::8, ::8],
               state_keys=np.array(list(enc.state_dict().keys())))
    np.savez_compressed(osp.join(HERE, "encoder.npz"), **t2n(out))
    print("encoder.npz", params.shape, hand_class)


def gen_mlp_head(ns):
    """InterHandSubNetwork (networks.py:83-105) with seeded weights on a seeded input."""
    opt = make_opt(3)
    out = {}
    for k in (3, 90):
        net = ns.networks.InterHandSubNetwork(opt, 1024 + 122, k)
        net.load_state_dict(seeded_state_dict(net, 500 + k))
        x = torch.tensor(np.random.RandomState(9 + k).normal(0, 0.5, (3, 1146)), dtype=torch.float32)
        with torch.no_grad():
            out[f"x_{k}"] = x
            out[f"y_{k}"] = net(x.clone())
    np.savez_compressed(osp.join(HERE, "mlp_head.npz"), **t2n(out))
    print("mlp_head.npz")


def gen_metrics(ns):
    """metric_utils.get_single_joints_error / get_single_pa_inter_joints_error (no rotation) and the
    evaluator's collision statistics on seeded predictions."""
    rng = np.random.RandomState(21)
    mu = ns.metric_utils
    out = {}
    for i in range(4):
        pred = rng.normal(0, 0.05, (42, 3)).astype(np.float32)
        gt = (pred + rng.normal(0, 0.01, (42, 3))).astype(np.float32)
        valid = (rng.uniform(size=(42, 1)) > 0.15).astype(np.float32)
        if i == 1:
            valid[0] = 0.0      # right wrist missing
        if i == 2:
            valid[21] = 0.0     # left wrist missing
        scale = [1.0, 1.0, 0.8, 1.3][i]
        out[f"pred_{i}"], out[f"gt_{i}"], out[f"valid_{i}"], out[f"scale_{i}"] = pred, gt, valid, np.float32(scale)
        out[f"j3d_err_{i}"] = np.array(mu.get_single_joints_error(pred, gt, valid, scale), dtype=np.float64)
        out[f"pa_err_{i}"] = np.array(mu.get_single_pa_inter_joints_error(pred, gt, valid, scale, use_rot=False), dtype=np.float64)
    np.savez_compressed(osp.join(HERE, "metrics.npz"), **out)
    print("metrics.npz")


def gen_evaluator(ns):
    """The reference's own ``Evaluator.update`` (utils/evaluator.py:38-135: per-sample records, fp16 vertex storage with
    ``save_verts``, the flip-back of ``do_flip`` samples, ``remove_redunc`` + the four metrics) on seeded predictions, and
    the arguments its ``save_pred_obj`` (utils/opt_utils.py:45-54) hands to ``ry_utils.save_mesh_to_obj`` for the
    reference's own OptimizeModel result stored in opt_traj.npz (the absent ``ry_utils`` is a stub that records them)."""
    import importlib
    import types
    ev = importlib.import_module("utils.evaluator")
    rng = np.random.RandomState(33)
    B = 6
    models = _mano_models(ns, 1)
    data_list = [dict(img_path=f"cap0/seq{j // 2}/cam4/image{j}.jpg") for j in range(B)]
    data_list[1].update(hand_type="right", scale=0.8)
    data_list[2].update(hand_type="interacting", scale=1.25, annot_type="human")
    data_list[5]["img_path"] = data_list[0]["img_path"]       # a padding duplicate (opt_dataset.py:49-51)
    dataset = types.SimpleNamespace(name="synthetic", data_list=data_list, image_root="/data/root")
    model = types.SimpleNamespace(inputSize=224, mano_models=models)
    f32 = lambda *s, sc=1.0: (rng.normal(0, sc, s)).astype(np.float32)
    gtj = np.concatenate([f32(B, 42, 3, sc=0.05), (rng.uniform(size=(B, 42, 1)) > 0.15).astype(np.float32)], axis=2)
    gtj[3, 0, 3] = 0.0
    pred = dict(pred_cam_params=f32(B, 3), pred_shape_params=f32(B, 20), pred_pose_params=f32(B, 96, sc=0.3),
                pred_hand_trans=f32(B, 1, 3, sc=0.03), pred_joints_3d=(gtj[:, :, :3] + f32(B, 42, 3, sc=0.01)).astype(np.float32),
                gt_joints_3d=gtj, collision_loss_origin_scale=np.abs(f32(B, 1556, sc=0.004)),
                do_flip=np.array([0, 1, 0, 1, 0, 0], dtype=np.int32), pred_hand_type=np.ones((B, 2), dtype=np.int32),
                pred_right_hand_verts=f32(B, 778, 3, sc=0.07), pred_left_hand_verts=f32(B, 778, 3, sc=0.07),
                gt_right_hand_verts=f32(B, 778, 3, sc=0.07), gt_left_hand_verts=f32(B, 778, 3, sc=0.07))
    out = {f"in_{k}": v.copy() for k, v in pred.items()}
    out["in_img_path"] = np.array([d["img_path"] for d in data_list])
    out["in_hand_type"] = np.array([d.get("hand_type", "") for d in data_list])
    out["in_scale"] = np.array([d.get("scale", -1.0) for d in data_list])
    e = ev.Evaluator(None, dataset, model)
    e.update(list(range(B)), {k: v.copy() for k, v in pred.items()}, save_verts=True)
    num_keys = ["pred_cam_params", "pred_shape_params", "pred_pose_params", "pred_hand_trans", "pred_joints_3d", "gt_joints_3d",
                "collision_loss_origin_scale", "pred_right_hand_verts", "pred_left_hand_verts", "gt_right_hand_verts", "gt_left_hand_verts"]
    for i, rec in enumerate(e.pred_results):
        for k in num_keys:
            out[f"rec{i}_{k}"] = rec[k]
        out[f"rec{i}_j3d_error"] = np.array(rec["j3d_error"], dtype=np.float64)
        out[f"rec{i}_pa_error"] = np.array(rec["pa_no_rot_inter_j3d_error"], dtype=np.float64)
        out[f"rec{i}_meta"] = np.array([rec["img_path"], rec["img_path_relative"], rec["hand_type"], rec["annot_type"]])
        out[f"rec{i}_scale"] = np.float64(rec["scale"])
    e.remove_redunc()
    out["n_after_remove_redunc"] = np.array(len(e.pred_results))
    out["metrics"] = np.array([e.mpjpe_3d, e.inter_mpjpe_3d, e.collision_ave, e.collision_max], dtype=np.float64)
    # without vertex storage: records carry no mesh, the flip-back skips them
    e2 = ev.Evaluator(None, dataset, model)
    e2.update(list(range(B)), {k: v.copy() for k, v in pred.items()}, save_verts=False)
    out["nov_has_verts"] = np.array(["pred_right_hand_verts" in r for r in e2.pred_results])
    out["nov_rec1_pred_joints_3d"] = e2.pred_results[1]["pred_joints_3d"]
    # mesh export of the reference's own refinement result
    traj = np.load(osp.join(HERE, "opt_traj.npz"))
    calls = []
    ns.opt_utils.ry_utils.save_mesh_to_obj = lambda path, verts, faces: calls.append((path, np.array(verts), np.array(faces)))
    res = {k[4:]: traj[k] for k in traj.files if k.startswith("out_")}
    ns.opt_utils.save_pred_obj("/res/dir", res, _mano_models(ns, 3), 7, 2, 30)
    (path, verts, faces), = calls
    out["obj_path"], out["obj_verts"], out["obj_faces"] = np.array(path), verts, faces
    np.savez_compressed(osp.join(HERE, "evaluator.npz"), **out)
    print("evaluator.npz", out["metrics"], path, verts.shape, faces.shape, faces.dtype)


def make_mlp_batch(B, seed):
    """Synthetic IHMR-MLP batch (schema of data/mlp_dataset.py:185-208) from the OPT synthetic batch."""
    from ihmr_amd.assets import synthetic_mano
    from ihmr_amd.synthetic import synthetic_opt_batch
    from oracle.opt_ref import OptimizeRef
    helper = OptimizeRef(synthetic_mano(True), synthetic_mano(False), B, [], save_mid_freq=1)

    def fwd(pose, shape, trans):
        helper.pred_right_orient, helper.pred_left_orient = pose[:, :3], pose[:, 48:51]
        helper.pred_right_pose_params, helper.pred_left_pose_params = pose[:, 3:48], pose[:, 51:]
        helper.pred_right_shape_params, helper.pred_left_shape_params = shape[:, :10], shape[:, 10:]
        helper.pred_hand_trans = trans.view(-1, 1, 3)
        return helper.get_mano_output()[2]

    b = synthetic_opt_batch(B, fwd, seed=seed, with_feat=True)
    b["init_hand_trans"] = b["init_hand_trans"][:, 0, :3].contiguous()
    b["img"] = torch.zeros(B, 3, 8, 8)
    b.pop("init_hand_trans_j")
    return b


def gen_mlp_test(ns):
    """The reference's MLPModel.test() (mlp_model.py:683-699) with six seeded sub-networks on a synthetic batch."""
    import importlib
    import sys as _sys
    from helpers import seeded_state_dict
    _sys.modules["ry_utils"].load_pkl = lambda p: {"mean_pose": np.zeros(48), "mean_betas": np.zeros(10)}
    mlp_model = importlib.import_module("models.mlp_model")
    B = 3
    opt = make_opt(B, strategy="mlp_default")
    opt.total_epoch = 1
    opt.pretrain_weights_dir = None
    model = mlp_model.MLPModel(opt)
    batch = make_mlp_batch(B, 777)
    batch["joints_3d"][1, 0, 3] = 0.0          # one sample without a right wrist in the GT weights
    model.set_input({k: v.clone() for k, v in batch.items()})
    strategy = ns.strategies.strategies["mlp_default"]
    model.set_update_info(strategy, 10)
    for sid in range(len(strategy)):
        model.add_new_network(sid)
        net = model.sub_network_list[sid]
        net.load_state_dict(seeded_state_dict(net, 900 + sid, last_scale=0.02))
        net.eval()
    model.set_input({k: v.clone() for k, v in batch.items()})
    model.test()
    res = model.get_pred_result()
    out = {f"in_{k}": v for k, v in batch.items()}
    out.update({f"out_{k}": v for k, v in res.items()})
    out["out_joints_3d_loss_p_batch"] = model.joints_3d_loss_p_batch
    out["out_joints_2d_loss_p_batch"] = model.joints_2d_loss_p_batch
    np.savez_compressed(osp.join(HERE, "mlp_test.npz"), **t2n(out))
    print("mlp_test.npz", {k: np.asarray(v).shape for k, v in res.items() if k.startswith("pred_")})

def gen_preprocess(ns):
    """The reference's own ``DataProcessor.padding_and_resize`` / ``random_flip(do_flip=True)`` /
    ``normalize_joints_2d`` (data/data_preprocess.py:45-72,162-169) run on seeded images, with ``cv2.resize``
    (absent) replaced by ``oracle/preprocess_ref.resize_linear_u8``; ToTensor + Normalize are torch's own float32
    ops (``div(255)``, ``sub_(0.5).div_(0.5)``, what torchvision 0.7 does).  Pins the size arithmetic, the padding,
    the joint scaling / flip / normalisation and the float conversion -- not the resize itself."""
    import importlib
    import types
    from oracle import preprocess_ref as P
    dp = importlib.import_module("data.data_preprocess")
    sys.modules["cv2"].resize = lambda img, dsize: P.resize_linear_u8(img, dsize[0], dsize[1])
    dp.cv2 = sys.modules["cv2"]
    rng = np.random.RandomState(7)
    out = {}
    cases = [(37, 53, 32, 0), (53, 37, 32, 1), (64, 64, 32, 0), (32, 32, 32, 1), (32, 20, 32, 0), (17, 100, 32, 1),
             (90, 64, 48, 0), (48, 96, 48, 1), (5, 3, 48, 0)]
    for i, (h, w, S, flip) in enumerate(cases):
        proc = types.SimpleNamespace(opt=types.SimpleNamespace(inputSize=S))
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        joints = np.concatenate([rng.uniform(0, [w, h], size=(42, 2)), rng.randint(0, 2, size=(42, 1))], 1).astype(np.float32)
        new_img, j = dp.DataProcessor.padding_and_resize(proc, img.copy(), joints.copy())
        if flip:
            dummy3 = np.zeros((42, 4), np.float32)
            mano = (np.zeros(96, np.float32), np.zeros(20, np.float32), np.ones(2, np.float32))
            gu = importlib.import_module("utils.geometry_utils")
            res = dp.DataProcessor.random_flip(proc, new_img, np.array([0, 1], np.float32), j, dummy3, mano, do_flip=True)
            new_img, j = res[0], res[2]
        j = dp.DataProcessor.normalize_joints_2d(proc, j)
        t = torch.from_numpy(np.ascontiguousarray(new_img.transpose(2, 0, 1))).float().div(255)
        t = t.sub_(0.5).div_(0.5)
        out[f"img{i}"] = img; out[f"joints{i}"] = joints; out[f"size{i}"] = np.array([S, flip], np.int32)
        out[f"u8_{i}"] = np.ascontiguousarray(new_img); out[f"f32_{i}"] = t.numpy(); out[f"jout{i}"] = j.astype(np.float32)
    out["n"] = np.array(len(cases))
    np.savez_compressed(osp.join(HERE, "preprocess.npz"), **out)
    print("preprocess.npz", len(cases), "cases")

def gen_mlp_train(ns):
    """One training step of the reference's MLPModel per stage of ``mlp_default`` (train_mlp.py:93-99:
    ``set_input -> retrive_prev_prediction -> forward -> compute_loss(stage weights) -> optimize_parameters``) on a
    synthetic batch, with a seeded sub-network: loss terms, the gradient of every sub-network parameter and the
    weights after the Adam step."""
    import importlib
    import sys as _sys
    from helpers import seeded_state_dict
    _sys.modules["ry_utils"].load_pkl = lambda p: {"mean_pose": np.zeros(48), "mean_betas": np.zeros(10)}
    mlp_model = importlib.import_module("models.mlp_model")
    B = 4
    opt = make_opt(B, strategy="mlp_default", is_train=True)
    opt.total_epoch = 1
    opt.pretrain_weights_dir = None
    model = mlp_model.MLPModel(opt)
    batch = make_mlp_batch(B, 778)
    batch["joints_3d"][1, 0, 3] = 0.0                    # a sample whose GT lacks the right wrist (root = joint 21)
    batch["mano_params_weight"][2, 1] = 0.0              # a sample without left-hand MANO annotation
    batch["hand_trans"][3, 0, 3] = 0.0                   # a sample without translation annotation
    batch["hand_type_array"][0] = torch.tensor([1.0, 0.0])   # a single-hand sample (collision masked)
    import copy
    strategy = copy.deepcopy(ns.strategies.strategies["mlp_default"])
    strategy[4]["loss_weights"]["shape_residual_loss"] = 1.0     # zero everywhere in mlp_default: exercise the term once
    model.set_update_info(strategy, 10)
    with torch.no_grad():                                 # train_mlp.py:60-66: backbone prediction -> "prev" tables
        model.set_input({k: v.clone() for k, v in batch.items()})
        model.forward(forward_backbone=True)
        model.compute_loss()
        model.save_pred_to_prev()
    out = {f"in_{k}": v for k, v in batch.items()}
    names = ["joints_2d_loss", "joints_3d_loss", "mano_pose_loss", "mano_shape_loss", "hand_trans_loss", "shape_reg_loss",
             "shape_residual_loss", "collision_loss", "loss"]
    for sid in range(len(strategy)):
        model.add_new_network(sid)
        net = model.sub_network_list[sid]
        net.load_state_dict(seeded_state_dict(net, 950 + sid, last_scale=0.05))
        net.train()
        model.set_input({k: v.clone() for k, v in batch.items()})
        model.retrive_prev_prediction()
        model.forward()
        model.compute_loss(strategy[sid]["loss_weights"])
        out[f"s{sid}_losses"] = np.array([float(getattr(model, n)) for n in names], np.float64)
        model.optimize_parameters()
        for k, prm in net.named_parameters():
            g, w = prm.grad.detach().clone(), prm.detach().clone()
            out[f"s{sid}_gradnorm_{k}"] = np.array(float(g.double().norm()))
            if g.numel() > 20000:          # the three big matrices: a regular sample of rows / columns (fixture size)
                g, w = g[::8, ::8], w[::8, ::8]
            out[f"s{sid}_grad_{k}"] = g
            out[f"s{sid}_new_{k}"] = w
        out[f"s{sid}_residual_cols"] = np.array(sum(model._MLPModel__get_param_dim(n) for n in strategy[sid]["update_params"]))
    out["loss_names"] = np.array(names)
    np.savez_compressed(osp.join(HERE, "mlp_train.npz"), **t2n(out))
    print("mlp_train.npz", {k: np.round(v, 6).tolist() for k, v in out.items() if k.endswith("_losses")})

def gen_encoder_train(ns):
    """The reference's InterHandEncoder in TRAIN mode (BatchNorm on batch statistics) on seeded weights / images: outputs, the
    gradient of every parameter for a seeded linear functional of the outputs (norms + samples), the running statistics of
    the first and the last BatchNorm after the step."""
    ns.networks.get_model = lambda arch: getattr(ns.resnet, arch)(pretrained=False, num_classes=512)
    B = 4
    opt = make_opt(B)
    rng = np.random.RandomState(3)
    mean_params = torch.tensor(rng.normal(0, 0.2, (1, 122)), dtype=torch.float32)
    mean_params[0, 0] = 5.0
    enc = ns.networks.InterHandEncoder(opt, mean_params.repeat(B, 1))
    enc.load_state_dict(seeded_state_dict(enc, 100))
    enc.train()
    img = torch.tensor(rng.uniform(-1, 1, (B, 3, 224, 224)), dtype=torch.float32)
    A = torch.tensor(rng.normal(0, 1, (B, 122)), dtype=torch.float32)
    Bm = torch.tensor(rng.normal(0, 1, (B, 2)), dtype=torch.float32)
    params, hand = enc(img)
    ((params * A).sum() + (hand * Bm).sum()).backward()
    out = dict(params=params.detach(), hand_class=hand.detach(),
               bn1_running_mean=enc.main_encoder.bn1.running_mean, last_bn_running_var=enc.main_encoder.layer4[2].bn3.running_var)
    names = []
    for k, p in enc.named_parameters():
        names.append(k)
        out[f"gradnorm/{k}"] = np.array(float(p.grad.double().norm()))
    for k in ("main_encoder.conv1.weight", "main_encoder.layer1.0.bn1.weight", "main_encoder.layer2.0.downsample.0.weight",
              "main_encoder.layer4.2.conv3.weight", "main_encoder.fc1.bias", "regressor_ih.0.weight", "hand_classifier.0.weight"):
        g = dict(enc.named_parameters())[k].grad
        out[f"grad/{k}"] = g.reshape(g.shape[0], -1)[::4, ::16] if g.dim() > 1 else g
    out["param_names"] = np.array(names)
    np.savez_compressed(osp.join(HERE, "encoder_train.npz"), **t2n(out))
    print("encoder_train.npz", len(names), "parameters")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns = import_reference()
    which = sys.argv[1:] or ["losses", "select", "opt_traj", "opt_traj_ragged", "opt_traj_variants", "encoder", "mlp_head", "metrics", "evaluator", "mlp_test", "preprocess", "mlp_train", "encoder_train"]
    for w in which:
        globals()[f"gen_{w}"](ns)
