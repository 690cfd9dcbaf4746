"""Host-side logic that needs no GPU: prediction-file schema, refinement-batch construction, strategy plumbing."""
import os
import sys
import types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_preds(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(pred_cam_params=r(B, 3), pred_shape_params=r(B, 20), pred_pose_params=r(B, 96), pred_hand_trans=r(B, 3),
                joints_2d=r(B, 42, 2), joints_3d=r(B, 42, 3), img_feat=r(B, 1024))


def test_prediction_file_roundtrip_uses_reference_schema(tmp_path):
    """data_utils.py:42-70 reads {img_path: {pred_cam_params, pred_shape_params, pred_pose_params, pred_hand_trans,
    joints_2d, joints_3d, img_feat}}: the writer produces exactly that, and the reader returns the same bits in the
    order of the requested image paths."""
    from ihmr_amd import pipeline, ry_utils
    B = 5
    preds = _fake_preds(B)
    names = [f"cam4/img_{i:04d}.jpg" for i in range(B)]
    f = str(tmp_path / "pred.pkl")
    pipeline.save_pred_file(f, names, preds)
    raw = ry_utils.load_pkl(f)
    assert sorted(raw) == sorted(names)
    assert sorted(raw[names[0]]) == sorted(pipeline.PRED_KEYS)
    assert raw[names[2]]["pred_pose_params"].shape == (96,) and raw[names[2]]["img_feat"].dtype == np.float32
    back = pipeline.load_pred_file(f, names[::-1])
    for k in pipeline.PRED_KEYS:
        assert torch.equal(back[k], preds[k].flip(0)), k


def test_refinement_batch_follows_opt_dataset():
    """opt_dataset.py:134-151: unit scores appended to the predicted joints, init_hand_trans_j = joint 21 - joint 0 with
    weight 1, init_hand_trans (B,1,4) with weight 1 for OPT and (B,3) for the MLP dataset (mlp_dataset.py:179)."""
    from ihmr_amd import pipeline
    B = 3
    preds = _fake_preds(B, 1)
    anno = dict(joints_2d=torch.zeros(B, 42, 3), joints_3d=torch.zeros(B, 42, 4), mano_pose=torch.zeros(B, 96), mano_betas=torch.zeros(B, 20),
                mano_params_weight=torch.ones(B, 2), hand_trans=torch.zeros(B, 1, 4), hand_type_array=torch.ones(B, 2), index=torch.arange(B))
    b = pipeline.refinement_batch(preds, anno)
    assert b["init_joints_2d"].shape == (B, 42, 3) and torch.all(b["init_joints_2d"][:, :, 2] == 1)
    assert b["init_joints_3d"].shape == (B, 42, 4) and torch.all(b["init_joints_3d"][:, :, 3] == 1)
    assert torch.equal(b["init_joints_3d"][:, :, :3], preds["joints_3d"])
    assert b["init_hand_trans"].shape == (B, 1, 4) and torch.equal(b["init_hand_trans"][:, 0, :3], preds["pred_hand_trans"])
    assert torch.all(b["init_hand_trans"][:, 0, 3] == 1)
    assert torch.equal(b["init_hand_trans_j"][:, 0, :3], preds["joints_3d"][:, 21] - preds["joints_3d"][:, 0])
    assert torch.equal(b["init_pose_params"], preds["pred_pose_params"]) and torch.equal(b["index"], anno["index"])
    m = pipeline.refinement_batch(preds, anno, for_mlp=True)
    assert m["init_hand_trans"].shape == (B, 3) and torch.equal(m["img_feat"], preds["img_feat"])


def test_packed_parameter_columns_match_reference_order():
    """mlp_model.py:426-439 / baseline_model.py:262-270: final_params = [cam 3 | R orient 3, R pose 45, L orient 3,
    L pose 45 | R shape 10, L shape 10 | trans 3]; the column table of the MLP model and the per-stage update sizes
    (strategies/mlp_default.py) must tile those 122 columns."""
    from ihmr_amd.mlp_model import COLS, PARAM_DIMS
    from ihmr_amd.strategies import make_mlp_strategy
    cover = np.zeros(122, int)
    for n, sl in COLS.items():
        assert sl.stop - sl.start == PARAM_DIMS[n], n
        cover[sl] += 1
    assert np.all(cover == 1)
    assert COLS["pred_cam_params"] == slice(0, 3) and COLS["pred_hand_trans"] == slice(119, 122)
    assert COLS["pred_right_orient"].start < COLS["pred_right_pose_params"].start < COLS["pred_left_orient"].start
    assert COLS["pred_right_shape_params"].start < COLS["pred_left_shape_params"].start
    dims = [sum(PARAM_DIMS[p] for p in st["update_params"]) for st in make_mlp_strategy()]
    assert dims == [3, 3, 3, 90, 20, 3]


def test_load_mano_pkl_reads_the_original_file_layout(tmp_path):
    """A file in the layout of the licence-gated ``MANO_RIGHT.pkl`` (chumpy objects for v_template / shapedirs /
    posedirs / J_regressor-as-scipy-sparse / uint32 faces / kintree_table with 2^32-1 as the root's parent), written
    from the synthetic asset with a throw-away ``chumpy`` module that is gone again at load time: the loader must
    return exactly the arrays it was written from, without chumpy installed (smplx 0.1.28 ``MANO.__init__`` fields)."""
    import pickle
    import sys
    import types

    import scipy.sparse as sp

    from ihmr_amd.assets import load_mano_pkl, synthetic_mano
    ref = synthetic_mano(True)
    mod_ch, mod_pkg = types.ModuleType("chumpy.ch"), types.ModuleType("chumpy")

    class Ch:                                     # chumpy.ch.Ch pickles its value under `x`
        def __init__(self, x): self.x = np.asarray(x, dtype=np.float64)
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    mod_ch.Ch = Ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod_pkg, mod_ch
    try:
        kin = np.stack([ref["parents"].astype(np.int64), np.arange(16)]).astype(np.uint32)   # parents[0] = -1 -> 4294967295
        data = dict(v_template=Ch(ref["v_template"]), shapedirs=Ch(ref["shapedirs"]),
                    posedirs=Ch(ref["posedirs"].T.reshape(778, 3, 135)), J_regressor=sp.csc_matrix(ref["J_regressor"].astype(np.float64)),
                    weights=Ch(ref["lbs_weights"]), kintree_table=kin, f=ref["faces"].astype(np.uint32),
                    hands_mean=ref["hands_mean"].astype(np.float64), hands_components=np.eye(45), bs_style="lbs", bs_type="lrotmin")
        f = tmp_path / "MANO_RIGHT.pkl"
        with open(f, "wb") as fh:
            pickle.dump(data, fh, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    got = load_mano_pkl(str(f))
    assert int(got["parents"][0]) == -1 and np.array_equal(got["parents"][1:], ref["parents"][1:])
    assert got["faces"].dtype == np.int64 and np.array_equal(got["faces"], ref["faces"])
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "hands_mean"):
        assert got[k].dtype == np.float32 and got[k].shape == ref[k].shape, k
        np.testing.assert_array_equal(got[k], ref[k].astype(np.float32), err_msg=k)


def test_two_hand_obj_export_face_indexing(tmp_path, mano_arrays):
    """utils/opt_utils.py:45-54 + ry_utils.save_mesh_to_obj: right vertices first, left faces shifted by 778, 1-based OBJ
    indices -- integer work, reproduced exactly when the file is read back."""
    import types

    from ihmr_amd import ry_utils
    right, left = mano_arrays
    rng = np.random.RandomState(0)
    pred = dict(pred_right_hand_verts=rng.randn(2, 778, 3).astype(np.float32), pred_left_hand_verts=rng.randn(2, 778, 3).astype(np.float32))
    models = dict(right=types.SimpleNamespace(faces=right["faces"]), left=types.SimpleNamespace(faces=left["faces"]))
    path = ry_utils.save_pred_obj(str(tmp_path), pred, models, iter_id=3, data_id=1, opt_iter=40, sample=1)
    assert path.endswith("iter_0003_stage_01_opt_iter_0040.obj")
    v, f = [], []
    for line in open(path):
        t = line.split()
        if t[0] == "v":
            v.append([float(x) for x in t[1:]])
        elif t[0] == "f":
            f.append([int(x) for x in t[1:]])
    v, f = np.array(v), np.array(f)
    assert v.shape == (1556, 3) and f.shape == (3076, 3)
    assert np.array_equal(f - 1, np.concatenate([right["faces"], left["faces"] + 778], 0))     # bit-exact face index
    assert f.min() == 1 and f.max() == 1556
    np.testing.assert_allclose(v[:778], pred["pred_right_hand_verts"][1], atol=5e-7)
    np.testing.assert_allclose(v[778:], pred["pred_left_hand_verts"][1], atol=5e-7)


def test_hand_type_bce_gradient_is_finite_when_the_sigmoid_saturates():
    """The handedness term of the Baseline training step (loss_utils.py:40-43) against torch's own
    F.binary_cross_entropy, including probabilities that are exactly 0 and 1 in fp32 (logit beyond +-17)."""
    import torch.nn.functional as F
    from ihmr_amd.baseline_train import hand_type_bce, hand_type_bce_grad
    s = torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.3, 0.999999], [1e-30, 0.5]], requires_grad=True)
    t = torch.tensor([[1.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    valid = torch.tensor([1.0, 1.0, 1.0, 0.0])
    loss = (F.binary_cross_entropy(s, t, reduction="none") * valid.reshape(-1, 1)).mean()
    loss.backward()
    assert torch.equal(hand_type_bce(s.detach(), t, valid).mean(), loss.detach())
    got = hand_type_bce_grad(s.detach(), t, valid)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, s.grad, rtol=1e-6, atol=0)
    # what reaches the logit: grad * s (1 - s) -- zero at saturation instead of NaN
    assert torch.isfinite(got * s.detach() * (1 - s.detach())).all()


def test_mpvpe_known_answers():
    """MPVPE (ihmr_amd.evaluator.get_single_verts_error / Evaluator.mpvpe_3d) on hand-computed cases."""
    from ihmr_amd.evaluator import Evaluator, get_single_verts_error
    rng = np.random.RandomState(0)
    gt = rng.randn(778, 3).astype(np.float32) * 0.05
    one_hot = np.zeros(778, np.float32); one_hot[0] = 1.0
    # a rigid offset disappears with the root alignment
    assert np.allclose(get_single_verts_error(gt + np.float32([0.1, -0.2, 0.3]), gt, one_hot, 1.0), 0.0, atol=1e-7)
    # one vertex (not the root) moved by a 3-4-5 vector: that vertex alone is off, by 5 mm
    p = gt.copy(); p[5] += np.float32([0.003, 0.004, 0.0])
    e = np.array(get_single_verts_error(p, gt, one_hot, 1.0))
    assert abs(e[5] - 0.005) < 1e-7 and np.allclose(np.delete(e, 5), 0.0, atol=1e-7) and abs(e.mean() - 0.005 / 778) < 1e-9
    # a root regressed from two vertices: moving one of them by 2 mm moves the root by 1 mm -> every vertex is 1 mm off
    half = np.zeros(778, np.float32); half[0] = half[1] = 0.5
    p = gt.copy(); p[0, 0] += 0.002
    e = np.array(get_single_verts_error(p, gt, half, 1.0))
    assert np.allclose(e, 0.001, atol=1e-7)
    assert np.allclose(get_single_verts_error(p, gt, half, 2.0), 0.0005, atol=1e-7)      # scale_factor divides, as in the MPJPE
    # through the Evaluator: only hands with a MANO annotation count; the mean runs over vertices of all counted hands
    mano = types.SimpleNamespace(faces=np.zeros((1538, 3), np.int64), J_regressor=np.stack([half] + [one_hot] * 15))
    ev = Evaluator(dict(right=mano, left=mano))
    B = 2
    res = dict(pred_cam_params=np.zeros((B, 3)), pred_shape_params=np.zeros((B, 20)), pred_pose_params=np.zeros((B, 96)), pred_hand_trans=np.zeros((B, 3)),
               pred_joints_3d=np.zeros((B, 42, 3), np.float32), gt_joints_3d=np.ones((B, 42, 4), np.float32), collision_loss_origin_scale=np.zeros((B, 1556), np.float32),
               gt_right_hand_verts=np.stack([gt, gt]), gt_left_hand_verts=np.stack([gt, gt]), pred_right_hand_verts=np.stack([p, gt]),
               pred_left_hand_verts=np.stack([gt, p]), mano_params_weight=np.float32([[1, 1], [1, 0]]))
    ev.update(np.arange(B), res)
    # counted hands: sample 0 right (1 mm everywhere), sample 0 left (0), sample 1 right (0); sample 1 left has no annotation
    assert abs(ev.mpvpe_3d - 0.001 / 3) < 1e-9
    assert ev.metric_sums()[8] == 3 * 778


def test_bench_quotes_only_profiles_of_the_loaded_library(monkeypatch):
    """bench.py takes `roofline.traffic` from a committed rocprofv3 PMC summary only when the summary's `_meta.json` records the source
    hash of the library that is loaded (round 2's line quoted a profile of older code); a profile of other code is never quoted."""
    import json as _json
    import bench
    from ihmr_amd import hip
    metas = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_meta.json")]
    assert metas, "profiles/ holds no source-hash records"
    with open(os.path.join(ROOT, "profiles", sorted(metas)[0])) as fh:
        good = _json.load(fh)["srchash"]
    monkeypatch.setattr(hip, "loaded_source_hash", lambda: good)
    prefix, meta = bench.committed_profile(None, None)
    assert prefix is not None and meta["srchash"] == good
    t, src, us = bench.pmc_traffic("sdf_dist_kernel", meta["batches_per_launch"], meta["config"]) if meta["config"] == "opt" else (1.0, "x", 1.0)
    assert t and src
    monkeypatch.setattr(hip, "loaded_source_hash", lambda: "0" * 64)
    assert bench.committed_profile(None, None) == (None, None)
    assert bench.pmc_traffic("sdf_dist_kernel", 7) == (None, None, None)
    monkeypatch.setattr(hip, "loaded_source_hash", lambda: None)          # IHMR_HIP_LIBRARY override / no build record
    assert bench.committed_profile(None, None) == (None, None)


def test_committed_profiles_belong_to_the_committed_sources():
    """The final-code profile series of the round was taken from exactly the sources in the tree (hash of csrc/ + the header + the
    hipcc flags), so the driver's bench line can quote it."""
    import json as _json
    from ihmr_amd import hip
    cur = hip._source_hash()
    hits = []
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        if f.endswith("_meta.json"):
            with open(os.path.join(ROOT, "profiles", f)) as fh:
                if _json.load(fh)["srchash"] == cur:
                    hits.append(f)
    assert any("_f7_" in f for f in hits) and any("baseline" in f for f in hits) and any("mlp" in f for f in hits), hits


def test_bench_finds_the_encoder_traffic_in_the_committed_profile(monkeypatch):
    """`secondary_configs.baseline.roofline.traffic` comes from the committed PMC profile of `bench.py --config baseline`: the stem is
    found by its template MODE (round 4's line showed null: the filter still looked for a template argument the kernel no longer has)."""
    import json as _json
    import bench
    from ihmr_amd import hip
    prof = os.path.join(ROOT, "profiles")
    metas = sorted(f for f in os.listdir(prof) if f.endswith("_baseline_meta.json") and f[:2] >= "r4")
    assert metas
    with open(os.path.join(prof, metas[-1])) as fh:
        good = _json.load(fh)["srchash"]
    monkeypatch.setattr(hip, "loaded_source_hash", lambda: good)
    traffic, src = bench.encoder_traffic()
    assert src and src.startswith("profiles/") and 2e9 < traffic < 3e10, (traffic, src)


def test_profiles_readme_table_is_generated_from_the_committed_csv_files():
    """profiles/README.md quotes per-kernel durations, instruction counts and HBM bytes of the round's final profile series; round 4's
    table had drifted from its csv files.  The rows of the current series are generated (scripts/profiles_table.py) and must stand in
    the README verbatim."""
    import json as _json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import profiles_table
    from ihmr_amd import hip
    cur = hip._source_hash()
    prof = os.path.join(ROOT, "profiles")
    series = sorted({f[:-len("_f7_meta.json")] for f in os.listdir(prof) if f.endswith("_f7_meta.json")
                     and _json.load(open(os.path.join(prof, f)))["srchash"] == cur})
    assert series, "no profile series of the committed sources"
    readme = open(os.path.join(prof, "README.md")).read()
    rows = profiles_table.table(series[-1])
    assert len(rows) >= 2
    for r in rows:
        assert r in readme, r[:160]


def test_committed_profile_carries_what_the_bench_line_quotes():
    """The rows `bench.py` reads from the committed profile of the driver's launch size exist and are plausible: HBM bytes per launch
    of the dominant kernel (`roofline.traffic`), its mean duration, and its vector-instruction count (`roofline.valu.issue_slots`)."""
    import csv
    import json as _json
    from ihmr_amd import hip
    cur = hip._source_hash()
    prof = os.path.join(ROOT, "profiles")
    metas = [f for f in os.listdir(prof) if f.endswith("_f7_meta.json")]
    metas = [f for f in metas if _json.load(open(os.path.join(prof, f)))["srchash"] == cur]
    assert metas, "no profile of the driver's launch size for these sources"
    prefix = os.path.join(prof, metas[0][:-len("_meta.json")])

    def row(suffix):
        with open(f"{prefix}_{suffix}.csv", newline="") as fh:
            rows = [r for r in csv.DictReader(fh) if r["kernel"].startswith("sdf_dist_kernel")]
        assert rows, suffix
        return max(rows, key=lambda r: int(r.get("launches") or r.get("calls")))

    traffic = float(row("pmc_traffic")["hbm_bytes_per_launch"])
    us = float(row("kernel_stats")["avg_us"])
    inst = float(row("pmc_sq")["SQ_INSTS_VALU_per_launch"])
    assert 2e7 < traffic < 5e8 and 10.0 < us < 200.0 and 1e6 < inst < 1e8
    # issue slots at the peak clock (bench.py: roofline.valu.issue_slots): a fraction of one
    assert 0.05 < inst * 3.0 / (1024.0 * us * 1e-6 * 2.4e9) < 1.0


def test_design_section_6_is_generated_from_the_committed_bench_line():
    """DESIGN.md section 6 quotes the round's measurements; round 5's section still carried round 4's numbers.  Its figures are now a
    block generated from the committed bench line of the committed sources (scripts/design_numbers.py, profiles/<series>_bench_line.json)
    and must stand in DESIGN.md verbatim; the line itself must be a line of the driver's command with the contract's keys."""
    import json as _json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import design_numbers
    from ihmr_amd import hip
    cur = hip._source_hash()
    prof = os.path.join(ROOT, "profiles")
    series = sorted(f[:-len("_f7_meta.json")] for f in os.listdir(prof) if f.endswith("_f7_meta.json")
                    and _json.load(open(os.path.join(prof, f)))["srchash"] == cur)
    assert series, "no profile series of the committed sources"
    line = os.path.join(prof, f"{series[-1]}_bench_line.json")
    assert os.path.isfile(line), line
    d = _json.loads([l for l in open(line).read().splitlines() if l.startswith("{")][-1])
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["config"]["latency_ms_per_refine_iter"] == d["latency"]["ms_per_refine_iter"]
    assert d["roofline"]["traffic"] and d["cpu_baseline"]["value"] > 0 and d["parity"]["vs_oracle"]["within_tolerance"]
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for row in design_numbers.block(line):
        assert row in design, row[:160]
