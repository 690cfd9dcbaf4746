"""CPU tests (``-m "not gpu"``): the oracle against the golden vectors captured from the reference
(``tests/golden/*.npz``, made by ``tests/golden/make_golden.py``), known-answer tests for the two
unpinned third-party seams, host logic, and the C-ABI library's exports."""
import ctypes
import os
import os.path as osp

import numpy as np
import pytest
import torch

GOLD = osp.join(osp.dirname(osp.abspath(__file__)), "golden")
ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


def gold(name):
    return dict(np.load(osp.join(GOLD, name), allow_pickle=False))


def T(a):
    return torch.tensor(np.asarray(a))


def close(a, b, atol, rtol=0.0, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(np.abs(a - b) <= atol + rtol * np.abs(b)), f"{what}: max err {np.abs(a - b).max():.3e}"


# ------------------------------------------------------------------------------------------ losses
def test_losses_match_reference():
    from oracle import losses_ref as L
    g = gold("losses.npz")
    pr = T(g["j2d_pred"]).requires_grad_(True)
    l, lb = L.joints_2d_loss(T(g["j2d_gt"]), pr, T(g["j2d_w"]))
    l.backward()
    close(l.detach(), g["j2d_loss"], 1e-7, what="j2d loss")
    close(lb.detach(), g["j2d_loss_batch"], 1e-7, what="j2d batch")
    close(pr.grad, g["j2d_grad"], 1e-9, what="j2d grad")

    leaf = T(g["j3d_pred"]).requires_grad_(True)
    pr3 = leaf * 1.0
    gt3 = T(g["j3d_gt"]).clone()
    l, lb = L.joints_3d_loss_(gt3, pr3, T(g["j3d_w"]))
    l.backward()
    close(l.detach(), g["j3d_loss"], 1e-9, what="j3d loss")
    close(lb.detach(), g["j3d_loss_batch"], 1e-9, what="j3d batch")
    close(leaf.grad, g["j3d_grad"], 1e-10, what="j3d grad")
    close(gt3, g["j3d_gt_aligned"], 0, what="gt aligned in place")
    close(pr3.detach(), g["j3d_pred_aligned"], 0, what="pred aligned in place")

    prt = T(g["tr_pred"]).requires_grad_(True)
    l = L.hand_trans_loss(T(g["tr_gt"]), prt, T(g["tr_w"]))
    l.backward()
    close(l.detach(), g["tr_loss"], 1e-10, what="trans loss")
    close(prt.grad, g["tr_grad"], 1e-10, what="trans grad")

    sh = T(g["sh"]).requires_grad_(True)
    l = L.shape_reg_loss(sh)
    l.backward()
    close(l.detach(), g["sh_loss"], 1e-7, what="shape reg")
    close(sh.grad, g["sh_grad"], 1e-8, what="shape reg grad")

    j = T(g["fin_j"]).requires_grad_(True)
    l, lb = L.finger_reg_loss(j)
    l.backward()
    close(l.detach(), g["fin_loss"], 1e-12, rtol=1e-6, what="finger loss")
    close(lb.detach(), g["fin_loss_batch"], 1e-12, rtol=1e-6, what="finger batch")
    close(j.grad, g["fin_grad"], 1e-12, rtol=1e-5, what="finger grad")


def test_transforms_match_reference():
    from oracle import losses_ref as L
    g = gold("losses.npz")
    close(L.batch_rodrigues_ref(T(g["rod_in"])), g["rod_out"], 1e-7, what="batch_rodrigues")
    close(L.batch_orthogonal_project(T(g["proj_X"]), T(g["proj_cam"])), g["proj_out"], 0, what="projection")


def test_filter_select_match_reference():
    from oracle.opt_ref import filter_by_losses, select_params
    g = gold("select.npz")
    params = {k: T(g[k]) for k in ("pred_left_pose_params", "pred_right_pose_params")}
    for tag, filt, sel in (("a", [("joints_3d_loss_p", "+0"), ("collision_loss", "-10")], "joints_3d_loss_p"),
                           ("b", [("joints_3d_loss_p", "+0"), ("collision_loss", "+0")], "collision_loss")):
        losses = {"joints_3d_loss_p": T(g["j3d"]).clone(), "collision_loss": T(g["col"]).clone()}
        upd = filter_by_losses(losses, filt)
        close(upd["joints_3d_loss_p"], g[f"{tag}_filtered_j3d"], 0, what="filtered j3d")
        close(upd["collision_loss"], g[f"{tag}_filtered_col"], 0, what="filtered col")
        selp, idx = select_params(params, upd, sel)
        assert np.array_equal(idx.numpy(), g[f"{tag}_idx"])
        for k in params:
            close(selp[k], g[f"{tag}_{k}"], 0, what=k)
    assert g["a_idx"][0] == 0 and g["a_idx"][1] == 3  # no-candidate -> origin; tie -> first index


def test_host_select_args_match_reference_filter():
    """The float32 filter factors the product hands to ihmr_opt_run_stage reproduce the reference's
    `bar = origin * (1 + (c + 0.1)/100)` comparison on the golden table."""
    from ihmr_amd.optimize_model import stage_to_args
    from ihmr_amd.strategies import make_opt_strategy
    g = gold("select.npz")
    a = stage_to_args(make_opt_strategy(3)[2])
    assert list(a.use_filter) == [0, 1, 1] and a.select_loss == 1 and a.param_mask == 16 + 32 and a.optimizer == 0 and a.n_iters == 4
    j3d, col = g["j3d"], g["col"]
    S, B = j3d.shape
    idx = np.zeros(B, np.int64)
    for b in range(B):
        best, bv = 0, j3d[0, b]
        for s in range(1, S):
            ok = (j3d[s, b] <= np.float32(j3d[0, b] * np.float32(a.filter_factor[1]))) and (col[s, b] <= np.float32(col[0, b] * np.float32(a.filter_factor[2])))
            key = j3d[s, b] if ok else np.float32(1e11)
            if key < bv:
                best, bv = s, key
        idx[b] = best
    assert np.array_equal(idx, g["a_idx"])


def test_stage_to_args_covers_what_the_reference_accepts():
    """Any subset of the leaf parameters, the three non-GT per-sample losses as criteria, several criteria on one loss,
    both optimizers; what the reference itself rejects raises a ValueError that names the strategy field."""
    from helpers import variant_strategy
    from ihmr_amd.optimize_model import stage_to_args
    st = variant_strategy(5)
    a = stage_to_args(st[0], "sgd", 2)
    assert list(a.use_filter) == [1, 0, 0] and a.select_loss == 2 and a.optimizer == 1 and a.save_freq == 2 and a.param_mask == 2
    assert abs(a.filter_factor[0] - 1.051) < 1e-6
    a = stage_to_args(st[3])
    assert list(a.use_filter) == [1, 1, 0] and abs(a.filter_factor[1] - 0.991) < 1e-6 and abs(a.filter_factor[0] - 1.201) < 1e-6
    mixed = dict(st[1], update_params=["pred_cam_params", "pred_left_orient", "pred_hand_trans"])
    assert stage_to_args(mixed).param_mask == 1 + 8 + 2
    for bad, field in ((dict(st[1], select_loss="joints_3d_loss"), "select_loss"), (dict(st[1], filter_loss=[("finger_reg_loss", "+0")]), "filter_loss"),
                       (dict(st[1], update_params=["pred_pose_params"]), "update_params"), (dict(st[1], filter_loss=[]), "filter_loss")):
        with pytest.raises(ValueError, match=field):
            stage_to_args(bad)
    with pytest.raises(ValueError, match="optimizer"):
        stage_to_args(st[1], "lbfgs")


# ------------------------------------------------------------------------------------------ OPT loop
def test_opt_trajectory_matches_reference(mano_arrays):
    """oracle OptimizeRef == the reference's OptimizeModel.optimize() (captured in opt_traj.npz)."""
    from ihmr_amd.strategies import make_opt_strategy
    from oracle.opt_ref import OptimizeRef
    g = gold("opt_traj.npz")
    epoch, freq = (int(x) for x in g["meta_epoch_freq"])
    batch = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    B = batch["init_cam"].shape[0]
    right, left = mano_arrays
    torch.set_num_threads(8)
    orc = OptimizeRef(right, left, B, make_opt_strategy(epoch), save_mid_freq=freq)
    orc.set_input(batch)
    orc.init_optimize()
    orc.optimize()
    res = orc.get_pred_result()
    for k, v in res.items():
        close(v, g[f"out_{k}"], 1e-6, what=k)   # same ops on the same machine class: round-off only
    close(orc.pred_joints_2d.detach(), g["out_pred_joints_2d"], 1e-6, what="joints_2d")
    close(orc.joints_3d_loss_p_batch.detach(), g["out_joints_3d_loss_p_batch"], 1e-6, what="j3d batch")
    close(orc.loss.detach(), g["out_loss"], 1e-5, what="total loss")


# ------------------------------------------------------------------------------------------ encoder / heads
def test_encoder_matches_reference():
    from helpers import seeded_state_dict
    from oracle.encoder_ref import InterHandEncoderRef
    g = gold("encoder.npz")
    enc = InterHandEncoderRef(T(g["mean_params"]).repeat(2, 1))
    assert list(enc.state_dict().keys()) == list(g["state_keys"]), "state_dict keys must equal the reference's"
    enc.load_state_dict(seeded_state_dict(enc, 100))
    enc.eval()
    img = torch.tensor(np.random.RandomState(7).uniform(-1, 1, (2, 3, 224, 224)), dtype=torch.float32)
    torch.set_num_threads(8)
    with torch.no_grad():
        params, hc = enc(img)
        feat = enc.main_encoder(img)
    close(feat, g["main_feat"], 1e-4, rtol=1e-4, what="main_feat")
    close(params, g["params"], 1e-4, rtol=1e-4, what="params")
    close(hc, g["hand_class"], 1e-5, what="hand_class")


def test_mlp_head_matches_reference():
    from helpers import seeded_state_dict
    from oracle.encoder_ref import InterHandSubNetworkRef
    g = gold("mlp_head.npz")
    for k in (3, 90):
        net = InterHandSubNetworkRef(1146, k)
        net.load_state_dict(seeded_state_dict(net, 500 + k))
        with torch.no_grad():
            y = net(T(g[f"x_{k}"]))
        close(y, g[f"y_{k}"], 1e-6, what=f"mlp head {k}")


def test_metrics_match_reference():
    from oracle import metrics_ref as M
    g = gold("metrics.npz")
    for i in range(4):
        a = M.single_joints_error(g[f"pred_{i}"], g[f"gt_{i}"], g[f"valid_{i}"], float(g[f"scale_{i}"]))
        close(np.array(a), g[f"j3d_err_{i}"], 1e-9, what="mpjpe")
        p = M.pa_no_rot_inter_joints_error(g[f"pred_{i}"], g[f"gt_{i}"], g[f"valid_{i}"], float(g[f"scale_{i}"]))
        close(np.array(p), g[f"pa_err_{i}"], 1e-7, what="pa mpjpe")


def _evaluator_inputs(g):
    pred = {k[3:]: g[k].copy() for k in g if k.startswith("in_") and k not in ("in_img_path", "in_hand_type", "in_scale")}
    data_list = []
    for path, ht, sc in zip(g["in_img_path"], g["in_hand_type"], g["in_scale"]):
        d = dict(img_path=str(path))
        if str(ht):
            d["hand_type"] = str(ht)
        if float(sc) > 0:
            d["scale"] = float(sc)
        data_list.append(d)
    data_list[2]["annot_type"] = "human"
    return pred, data_list


def test_evaluator_records_match_reference():
    """The reference's own Evaluator.update (utils/evaluator.py:38-135) on seeded predictions: every field of every record --
    incl. the fp16 meshes of ``save_verts`` and the flip-back of the two ``do_flip`` samples -- bit for bit, the duplicate
    removal, and the four metrics."""
    from ihmr_amd.evaluator import Evaluator
    g = gold("evaluator.npz")
    pred, data_list = _evaluator_inputs(g)
    B = len(data_list)
    ev = Evaluator(None, data_list, image_root="/data/root")
    ev.update(list(range(B)), pred, save_verts=True)
    keys = ["pred_cam_params", "pred_shape_params", "pred_pose_params", "pred_hand_trans", "pred_joints_3d", "gt_joints_3d",
            "collision_loss_origin_scale", "pred_right_hand_verts", "pred_left_hand_verts", "gt_right_hand_verts", "gt_left_hand_verts"]
    for i, rec in enumerate(ev.pred_results):
        for k in keys:
            assert rec[k].dtype == g[f"rec{i}_{k}"].dtype, (i, k, rec[k].dtype)
            assert np.array_equal(rec[k], g[f"rec{i}_{k}"]), (i, k)
        assert rec["pred_right_hand_verts"].dtype == np.float16
        close(np.array(rec["j3d_error"]), g[f"rec{i}_j3d_error"], 1e-12, what="mpjpe")
        close(np.array(rec["pa_no_rot_inter_j3d_error"]), g[f"rec{i}_pa_error"], 1e-9, what="pa mpjpe")
        assert [rec["img_path"], rec["img_path_relative"], rec["hand_type"], rec["annot_type"]] == [str(x) for x in g[f"rec{i}_meta"]]
        assert float(rec["scale"]) == float(g[f"rec{i}_scale"])
    # a flipped sample really was flipped (guards against a golden without the branch)
    assert not np.array_equal(g["rec1_pred_joints_3d"], g["in_pred_joints_3d"][1])
    ev.remove_redunc()
    assert len(ev.pred_results) == int(g["n_after_remove_redunc"]) == B - 1
    close(np.array([ev.mpjpe_3d, ev.inter_mpjpe_3d, ev.collision_ave, ev.collision_max]), g["metrics"], 1e-12, rtol=1e-6, what="metrics")   # the reference averages the depths in float32, this build in float64
    # save_verts=False: no mesh in the records, the flip-back of the rest unchanged
    pred, _ = _evaluator_inputs(g)
    ev2 = Evaluator(None, data_list, image_root="/data/root")
    ev2.update(list(range(B)), pred, save_verts=False)
    assert np.array_equal(np.array(["pred_right_hand_verts" in r for r in ev2.pred_results]), g["nov_has_verts"]) and not g["nov_has_verts"].any()
    assert np.array_equal(ev2.pred_results[1]["pred_joints_3d"], g["nov_rec1_pred_joints_3d"])


def read_obj(path):
    v, f = [], []
    for line in open(path):
        t = line.split()
        if t and t[0] == "v":
            v.append([float(x) for x in t[1:]])
        elif t and t[0] == "f":
            f.append([int(x) for x in t[1:]])
    return np.array(v), np.array(f, dtype=np.int64)


def test_mesh_export_matches_reference_save_pred_obj(tmp_path):
    """utils/opt_utils.py:45-54 run by the reference itself on its own OptimizeModel result (opt_traj.npz): the vertex block
    and the combined face index it hands to the OBJ writer, and the file name.  This build's export of the same result, read
    back from the file: faces identical integers, vertices to the writer's 1e-6 print precision."""
    import types
    from ihmr_amd import ry_utils
    from ihmr_amd.assets import synthetic_mano
    g, traj = gold("evaluator.npz"), gold("opt_traj.npz")
    res = {k[4:]: traj[k] for k in traj if k.startswith("out_")}
    models = dict(right=types.SimpleNamespace(faces=synthetic_mano(True)["faces"]), left=types.SimpleNamespace(faces=synthetic_mano(False)["faces"]))
    path = ry_utils.save_pred_obj(str(tmp_path), res, models, 7, 2, 30)
    assert osp.basename(path) == osp.basename(str(g["obj_path"]))
    v, f = read_obj(path)
    assert np.array_equal(f - 1, g["obj_faces"]) and g["obj_faces"].dtype == np.int64
    assert v.shape == g["obj_verts"].shape == (1556, 3)
    close(v, g["obj_verts"], 5.01e-7, what="OBJ vertices")


# ------------------------------------------------------------------------------------------ seam KATs (unpinned)
def test_mano_oracle_known_answers(mano_arrays):
    from oracle.mano_ref import ManoRef, rodrigues_smplx
    right, _ = mano_arrays
    arr = dict(right)
    arr["hands_mean"] = np.zeros(45, np.float32)
    m = ManoRef(arr)
    z = lambda d: torch.zeros(1, d)
    out = m(global_orient=z(3), hand_pose=z(45), betas=z(10))
    close(out.vertices[0], arr["v_template"], 1e-7, what="zero pose -> template")
    close(out.joints[0], arr["J_regressor"] @ arr["v_template"], 1e-7, what="zero pose -> regressed joints")
    # pure global rotation: rigid rotation about the root joint
    r = torch.tensor([[0.3, -0.7, 0.5]])
    out2 = m(global_orient=r, hand_pose=z(45), betas=z(10))
    R = rodrigues_smplx(r)[0]
    root = out.joints[0, 0]
    close(out2.vertices[0], (out.vertices[0] - root) @ R.T + root, 2e-7, what="rigid rotation about the root")
    # fp64 finite-difference check of the autograd gradient
    m64 = ManoRef(right, dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    o = (torch.randn(1, 3, generator=g, dtype=torch.float64) * 0.5).requires_grad_(True)
    p = (torch.randn(1, 45, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    b = torch.randn(1, 10, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(1, 778, 3, generator=g, dtype=torch.float64)
    f = lambda o_, p_, b_: (m64(global_orient=o_, hand_pose=p_, betas=b_).vertices * w).sum()
    f(o, p, b).backward()
    eps = 1e-6
    for t in (o, p, b):
        for idx in (0, t.shape[1] - 1):
            d = torch.zeros_like(t)
            d[0, idx] = eps
            args_p = [x.detach() + (d if x is t else 0) for x in (o, p, b)]
            args_m = [x.detach() - (d if x is t else 0) for x in (o, p, b)]
            fd = (f(*args_p) - f(*args_m)) / (2 * eps)
            assert abs(float(fd) - float(t.grad[0, idx])) < 1e-6 * max(1.0, abs(float(fd)))


def test_mirrored_left_equals_explicit_mirror(mano_arrays):
    """Left hand through the right model (optimize_model.py:180-211) == the explicit left model."""
    from oracle.mano_ref import ManoRef
    right, left = mano_arrays
    l_arr = dict(left)
    l_arr["shapedirs"] = left["shapedirs"].copy()
    l_arr["shapedirs"][:, 0, :] *= -1  # the reference's sign fix (:109-113)
    g = torch.Generator().manual_seed(3)
    o, p, b = torch.randn(2, 3, generator=g) * 0.5, torch.randn(2, 45, generator=g) * 0.3, torch.randn(2, 10, generator=g)
    sgn = torch.tensor([1.0, -1.0, -1.0])
    via_right = ManoRef(right)(global_orient=o * sgn, hand_pose=(p.view(-1, 3) * sgn).view(2, 45), betas=b)
    explicit = ManoRef(l_arr)(global_orient=o, hand_pose=p, betas=b)
    flip = torch.tensor([-1.0, 1.0, 1.0])
    close(via_right.vertices * flip, explicit.vertices, 2e-7, what="mirrored verts")
    close(via_right.joints * flip, explicit.joints, 2e-7, what="mirrored joints")


def _icosphere(radius=0.6, sub=3):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, float) / np.linalg.norm(x) for x in v]
    for _ in range(sub):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, np.int32)


def test_sdf_oracle_known_answers():
    """Closed convex mesh: phi == radius - |p| inside (within the faceting error), 0 outside; the
    point-triangle distance and the ray test against hand-computed values."""
    from oracle import sdf_ref
    v, f = _icosphere(0.6, 3)
    phi = sdf_ref.sdf_grid(torch.tensor(v)[None], torch.tensor(f))[0].numpy()
    c = (2 * np.arange(32) + 1) / 32 - 1
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    inside = r < 0.59
    outside = r > 0.61
    assert np.all(phi[outside] == 0)
    assert np.all(phi[inside] > 0)
    assert np.abs(phi[inside] - (0.6 - r[inside])).max() < 0.004  # icosphere faceting (sub 3) sags < 0.4 %
    L = sdf_ref._lib()
    a, b, cc = (np.array(x, np.float32) for x in ((0, 0, 0), (1, 0, 0), (0, 1, 0)))
    P = lambda *x: np.array(x, np.float32)
    d2 = lambda p: L.ihmr_oracle_point_tri_dist2(a.ctypes.data, b.ctypes.data, cc.ctypes.data, p.ctypes.data)
    assert abs(d2(P(0.25, 0.25, 2.0)) - 4.0) < 1e-6          # above the interior
    assert abs(d2(P(-1.0, -1.0, 0.0)) - 2.0) < 1e-6          # vertex region a
    assert abs(d2(P(0.5, -2.0, 0.0)) - 4.0) < 1e-6           # edge ab
    assert abs(d2(P(1.0, 1.0, 0.0)) - 0.5) < 1e-6            # edge bc
    a2, b2, c2 = (np.array(x, np.float32) for x in ((1, -1, -1), (1, 1, -1), (1, 0, 1)))
    hit = lambda p: L.ihmr_oracle_ray_hit_px(a2.ctypes.data, b2.ctypes.data, c2.ctypes.data, p.ctypes.data)
    assert hit(P(0, 0, 0)) == 1 and hit(P(2, 0, 0)) == 0 and hit(P(0, 0.9, 0.9)) == 0


def test_sdf_oracle_invariances(mano_arrays):
    """disjoint hands -> 0; translation invariance; swapping the hands swaps the two 778-halves."""
    from oracle.sdf_ref import SDFLossRef
    right, left = mano_arrays
    v = torch.tensor(right["v_template"])
    l = v.clone()
    l[:, 0] = -l[:, 0]
    pair = torch.stack([v, l + torch.tensor([0.15, 0.0, 0.02])])[None]
    mod = SDFLossRef(right["faces"], right["faces"])
    loss, pv, os_ = mod(pair, return_per_vert_loss=True, return_origin_scale_loss=True)
    assert float(loss) > 0
    far = torch.stack([v, l - torch.tensor([0.5, 0.0, 0.0])])[None]
    assert float(mod(far)) == 0.0
    loss_t, pv_t, _ = mod(pair + torch.tensor([0.25, -0.5, 1.0]), return_per_vert_loss=True, return_origin_scale_loss=True)
    close(pv_t, pv, 2e-4, what="translation invariance")  # fp32 cancellation after the shift
    loss_s, pv_s, _ = mod(pair.flip(1), return_per_vert_loss=True, return_origin_scale_loss=True)
    close(pv_s[:, :778], pv[:, 778:], 0, what="swap halves")
    close(pv_s[:, 778:], pv[:, :778], 0, what="swap halves")


# ------------------------------------------------------------------------------------------ product host logic / ABI
def test_library_exports_every_declared_symbol():
    """libihmr_hip.so loads on a CPU-only host and exports every function include/ihmr_hip.h declares."""
    import re
    from ihmr_amd import hip
    path = hip.build()
    L = ctypes.CDLL(path)
    header = open(osp.join(ROOT, "include", "ihmr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ihmr_[a-z0-9_]+)\s*\(", header)))
    assert sorted(hip.EXPORTED_SYMBOLS) == declared, (declared, hip.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert b"gfx950" in hip.lib().ihmr_version()


def test_product_fails_loudly_without_gpu():
    from ihmr_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip.require_gpu()
    import types
    from ihmr_amd.optimize_model import OptimizeModel
    with pytest.raises(RuntimeError):
        OptimizeModel(types.SimpleNamespace(batchSize=2))


def test_product_never_imports_oracle():
    import re
    for dirpath, _, files in os.walk(osp.join(ROOT, "ihmr_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(osp.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"


def test_strategies_shape():
    from ihmr_amd.strategies import make_opt_strategy, strategies
    assert [s["epoch"] for s in strategies["opt_default"]] == [300] * 4
    assert sum(s["epoch"] + 1 for s in make_opt_strategy(49)) == 200
    assert len(strategies["mlp_default"]) == 6


def test_synthetic_asset_is_mano_shaped(mano_arrays):
    right, left = mano_arrays
    assert right["v_template"].shape == (778, 3) and right["faces"].shape == (1538, 3)
    assert right["posedirs"].shape == (135, 2334) and right["shapedirs"].shape == (778, 3, 10)
    close(right["J_regressor"].sum(1), np.ones(16), 1e-5)
    close(right["lbs_weights"].sum(1), np.ones(778), 1e-5)
    # combined two-hand face index (utils/opt_utils.py:48-54) is bit-exact integer work
    comb = np.concatenate([right["faces"], left["faces"] + 778], 0)
    assert comb.dtype == np.int64 and comb.max() == 2 * 778 - 1 and comb.shape == (3076, 3)
    assert np.abs(left["shapedirs"][:, 0, :] - right["shapedirs"][:, 0, :]).mean() < 1e-7  # triggers the sign fix


def test_mlp_test_path_matches_reference(mano_arrays):
    """oracle MLPRef.test() == the reference's MLPModel.test() (captured in mlp_test.npz)."""
    from helpers import seeded_state_dict
    from ihmr_amd.strategies import make_mlp_strategy
    from oracle.mlp_ref import MLPRef
    g = gold("mlp_test.npz")
    batch = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    B = batch["init_cam"].shape[0]
    right, left = mano_arrays
    torch.set_num_threads(8)
    m = MLPRef(right, left, B, make_mlp_strategy(), num_data=10)
    for sid, net in enumerate(m.nets):
        net.load_state_dict(seeded_state_dict(net, 900 + sid, last_scale=0.02))
    m.set_input(batch)
    m.test()
    res = m.get_pred_result()
    for k, v in res.items():
        close(v, g[f"out_{k}"], 2e-6, what=k)
    close(m.joints_3d_loss_p_batch, g["out_joints_3d_loss_p_batch"], 1e-6, what="j3d_p batch")
    close(m.joints_2d_loss_p_batch, g["out_joints_2d_loss_p_batch"], 1e-6, what="j2d_p batch")
    print("kept per stage:", [k.tolist() for k in m.kept_history])


def test_mlp_train_step_matches_reference(mano_arrays):
    """oracle MLPRef.train_forward_backward() == one training step of the reference's MLPModel per stage (mlp_train.npz):
    every loss term, the gradient of every sub-network parameter, and the weights after torch.optim.Adam's step."""
    from helpers import seeded_state_dict
    from ihmr_amd.strategies import make_mlp_strategy
    from oracle.mlp_ref import MLPRef
    g = gold("mlp_train.npz")
    batch = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    B = batch["init_cam"].shape[0]
    right, left = mano_arrays
    torch.set_num_threads(8)
    strategy = make_mlp_strategy()
    strategy[4]["loss_weights"]["shape_residual_loss"] = 1.0      # as the fixture was generated
    m = MLPRef(right, left, B, strategy, num_data=10)
    m.set_input(batch)
    m.init_prev_from_backbone()
    names = [str(n) for n in g["loss_names"]]
    for sid, net in enumerate(m.nets):
        net.load_state_dict(seeded_state_dict(net, 950 + sid, last_scale=0.05))
        m.set_input(batch)
        terms = m.train_forward_backward(sid)
        close([terms[n] for n in names], g[f"s{sid}_losses"], 1e-6, 1e-5, what=f"stage {sid} losses")
        opt = torch.optim.Adam(net.parameters(), lr=strategy[sid]["lr"])
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        opt.step()
        for k, p in net.named_parameters():
            gr, new = grads[k], p.detach()
            scale = float(g[f"s{sid}_gradnorm_{k}"]) / np.sqrt(gr.numel())       # rms gradient entry
            assert abs(float(gr.double().norm()) - float(g[f"s{sid}_gradnorm_{k}"])) <= 1e-4 * float(g[f"s{sid}_gradnorm_{k}"]) + 1e-12
            if gr.numel() > 20000:
                gr, new = gr[::8, ::8], new[::8, ::8]
            close(gr, g[f"s{sid}_grad_{k}"], 2e-4 * scale + 1e-10, 1e-4, what=f"stage {sid} grad {k}")
            close(new, g[f"s{sid}_new_{k}"], 2e-6, what=f"stage {sid} weights after the step {k}")


def test_encoder_train_mode_matches_reference():
    """oracle InterHandEncoderRef in train() mode (BatchNorm on batch statistics) + autograd == the reference's own
    InterHandEncoder (encoder_train.npz): outputs, the gradient norm of every parameter, gradient samples, running statistics.
    This is the reference the GPU training path (tests/test_gpu_encoder_train.py) is compared with."""
    from helpers import seeded_state_dict
    from oracle.encoder_ref import InterHandEncoderRef
    g = gold("encoder_train.npz")
    B = 4
    rng = np.random.RandomState(3)
    mean_params = torch.tensor(rng.normal(0, 0.2, (1, 122)), dtype=torch.float32)
    mean_params[0, 0] = 5.0
    ref = InterHandEncoderRef(mean_params.repeat(B, 1))
    ref.load_state_dict(seeded_state_dict(ref, 100))
    ref.train()
    torch.set_num_threads(8)
    img = torch.tensor(rng.uniform(-1, 1, (B, 3, 224, 224)), dtype=torch.float32)
    A = torch.tensor(rng.normal(0, 1, (B, 122)), dtype=torch.float32)
    Bm = torch.tensor(rng.normal(0, 1, (B, 2)), dtype=torch.float32)
    p, h = ref(img)
    ((p * A).sum() + (h * Bm).sum()).backward()
    close(p.detach(), g["params"], 1e-5, what="train-mode params")
    close(h.detach(), g["hand_class"], 1e-6, what="train-mode hand type")
    params = dict(ref.named_parameters())
    assert [str(n) for n in g["param_names"]] == list(params)
    for k, prm in params.items():
        n_ref = float(g[f"gradnorm/{k}"])
        assert abs(float(prm.grad.double().norm()) - n_ref) <= 1e-3 * n_ref + 1e-12, k
    for key in [k for k in g if k.startswith("grad/")]:
        gr = params[key[5:]].grad
        gr = gr.reshape(gr.shape[0], -1)[::4, ::16] if gr.dim() > 1 else gr
        close(gr, g[key], 1e-3 * float(np.abs(g[key]).max()) + 1e-9, what=key)
    close(ref.main_encoder.bn1.running_mean, g["bn1_running_mean"], 1e-6, what="bn1 running mean")
    close(ref.main_encoder.layer4[2].bn3.running_var, g["last_bn_running_var"], 1e-5, what="last bn running var")


def _replay_opt_golden(mano_arrays, g, prefix, strategy, optimizer="adam"):
    from oracle.opt_ref import OptimizeRef
    epoch, freq = (int(x) for x in g["meta_epoch_freq"])
    batch = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    B = batch["init_cam"].shape[0]
    right, left = mano_arrays
    torch.set_num_threads(8)
    orc = OptimizeRef(right, left, B, strategy(epoch), save_mid_freq=freq, optimizer=optimizer)
    orc.set_input(batch)
    orc.init_optimize()
    orc.optimize()
    res = orc.get_pred_result()
    for k, v in res.items():
        if f"{prefix}out_{k}" in g:
            close(v, g[f"{prefix}out_{k}"], 1e-6, what=k)
    close(orc.pred_joints_2d.detach(), g[f"{prefix}out_pred_joints_2d"], 1e-6, what="joints_2d")
    close(orc.joints_3d_loss_p_batch.detach(), g[f"{prefix}out_joints_3d_loss_p_batch"], 1e-6, what="j3d batch")
    close(orc.joints_2d_loss_p_batch.detach(), g[f"{prefix}out_joints_2d_loss_p_batch"], 1e-6, what="j2d batch")
    close(orc.loss.detach(), g[f"{prefix}out_loss"], 1e-5, what="total loss")
    return orc


def test_opt_ragged_trajectory_matches_reference(mano_arrays):
    """oracle == the reference's OptimizeModel on the RAGGED batch (single-hand samples, missing / half-weight wrists,
    zero-weight joints, no 3-D target, separated hands; tests/helpers.py:ragged_opt_batch)."""
    from ihmr_amd.strategies import make_opt_strategy
    g = gold("opt_traj_ragged.npz")
    orc = _replay_opt_golden(mano_arrays, g, "", make_opt_strategy)
    ht = g["in_hand_type_array"]
    assert (ht.sum(1) < 1.5).sum() == 2 and np.all(g["out_collision_loss"][ht.sum(1) < 1.5] == 0.0)
    assert g["out_collision_loss"][7] == 0.0 and np.abs(g["out_collision_loss_origin_scale"][7]).max() == 0.0
    assert np.abs(g["out_collision_loss_origin_scale"][1]).max() > 0  # unmasked per-vertex depths (loss_utils.py:189)
    assert len({tuple(s) for s in np.stack(orc.selected).T.tolist()}) > 1, "samples must not all select the same snapshots"


def test_opt_variant_trajectories_match_reference(mano_arrays):
    """oracle == the reference's OptimizeModel with (a) non-default filter / select criteria, (b) optimizer = 'sgd'."""
    from helpers import variant_strategy
    from ihmr_amd.strategies import make_opt_strategy
    g = gold("opt_traj_variants.npz")
    a = _replay_opt_golden(mano_arrays, g, "crit_", variant_strategy)
    b = _replay_opt_golden(mano_arrays, g, "sgd_", make_opt_strategy, optimizer="sgd")
    print("selected (criteria):", np.stack(a.selected).tolist(), " (sgd):", np.stack(b.selected).tolist())
    assert np.stack(a.selected).max() > 0 and np.stack(b.selected).max() > 0


def test_sdf_oracle_convention_switches(mano_arrays):
    """SDFLossRef(align_corners=, loss_divisor=): the divisor only rescales the per-sample loss; align_corners=True
    samples the same grid on the corner-aligned lattice (a query on a voxel centre of THAT lattice returns the voxel)."""
    from oracle import sdf_ref
    right, left = mano_arrays
    v = torch.tensor(right["v_template"])
    other = v.clone(); other[:, 0] = -other[:, 0] + 0.03
    hv = torch.stack([v, other])[None]
    a = sdf_ref.SDFLossRef(right["faces"], left["faces"])(hv, return_per_vert_loss=True, return_origin_scale_loss=True)
    b = sdf_ref.SDFLossRef(right["faces"], left["faces"], loss_divisor=1.0)(hv, return_per_vert_loss=True, return_origin_scale_loss=True)
    c = sdf_ref.SDFLossRef(right["faces"], left["faces"], align_corners=True)(hv, return_per_vert_loss=True, return_origin_scale_loss=True)
    assert float(a[0]) > 0
    close(b[0], 4.0 * a[0], 1e-6, what="divisor")
    close(b[1], a[1], 0, what="per-vertex values do not depend on the divisor")
    assert float((c[1] - a[1]).abs().max()) > 1e-4, "align_corners must change the sampled values"
    # swap_xz: the query's x and z exchange roles -- the same as sampling the default module at queries mirrored in their own normalised
    # frame; checked against a direct statement: phi transposed in (x, z) and sampled with the default order gives the same values
    import torch.nn.functional as F
    d = sdf_ref.SDFLossRef(right["faces"], left["faces"], swap_xz=True)(hv, return_per_vert_loss=True, return_origin_scale_loss=True)
    centre, scale = sdf_ref.hand_boxes(hv, sdf_ref.SCALE_FACTOR)
    vn = (hv - centre) / scale
    phi_r = sdf_ref.sdf_grid(vn[:, 0].contiguous(), torch.tensor(right["faces"].astype(np.int32)), 32)        # [z][y][x]
    q = (hv[:, 1] - centre[:, 0]) / scale[:, 0]
    direct = F.grid_sample(phi_r.permute(0, 3, 2, 1)[:, None].contiguous(), q.view(1, -1, 1, 1, 3), mode="bilinear", padding_mode="zeros",
                           align_corners=False).view(1, -1)
    close(d[1][:, :778], direct, 1e-7, what="swap_xz = the [x][y][z] tensor sampled as it is")
    assert float((d[1] - a[1]).abs().max()) > 1e-4
