"""Host-side evaluation, mirroring the metric half of the reference's ``src/utils/evaluator.py`` and
``src/utils/metric_utils.py`` (CPU / numpy in the reference too; the visualisation half is out of scope).

``Evaluator.update(data_idxs, pred_results)`` consumes the dict of ``get_pred_result()`` exactly like
``evaluator.py:38-97``; the four reported metrics are ``mpjpe_3d``, ``inter_mpjpe_3d``, ``collision_ave``,
``collision_max`` (``evaluator.py:149-181``, printed by ``optimize.py:98-102``), plus ``mpvpe_3d`` for the models that
export GT meshes (IHMR-Baseline / IHMR-MLP; see :func:`get_single_verts_error`).  For multi-GPU runs
``metric_sums()`` returns the additive form that :func:`ihmr_amd.dist.reduce_metrics` all-reduces.

``Evaluator.update_device(...)`` is the same arithmetic on the GPU (``ihmr_eval_metrics``): it takes
the DEVICE tensors of a model (no export to numpy), leaves (B,6) float64 partial results on the device and only
adds them up when the metrics are asked for -- the per-sample Python loop and the 1 MB/batch device-to-host copies
of ``get_pred_result()`` drop out of a throughput run.
"""
from __future__ import annotations

import numpy as np


def get_single_joints_error(pred, gt, joint_weights, scale_factor):
    """metric_utils.py:23-38 -- per-hand MPJPE; the root subtraction is cumulative on the same copies."""
    a, b = pred.copy(), gt.copy()
    errors = []
    for i in (0, 21):
        if joint_weights[i, 0] > 0:
            a -= a[i:i + 1, :]
            b -= b[i:i + 1, :]
            for j in range(21):
                if joint_weights[i + j, 0] > 0:
                    errors.append(np.linalg.norm(a[i + j] - b[i + j]) / scale_factor)
    return errors


def calc_transform_no_rot(S1, S2):
    """metric_utils.py:107-117 -- per-axis mean / std alignment."""
    m1, m2 = np.mean(S1, axis=0).reshape(1, 3), np.mean(S2, axis=0).reshape(1, 3)
    s1, s2 = np.std(S1, axis=0).reshape(1, 3), np.std(S2, axis=0).reshape(1, 3)
    return (S1 - m1) / s1 * s2 + m2


def get_single_pa_inter_joints_error(pred, gt, joints_valid, scale_factor):
    """metric_utils.py:120-143 with use_rot=False."""
    v = joints_valid[:, 0] if joints_valid.ndim == 2 else joints_valid
    if np.sum(v) < 2.0:
        return []
    p, g = pred[v > 0, :3], gt[v > 0, :3]
    return (np.linalg.norm(calc_transform_no_rot(p.copy(), g.copy()) - g, axis=1) / scale_factor).tolist()


def get_single_verts_error(pred_verts, gt_verts, root_weights, scale_factor):
    """MPVPE of ONE hand (BASELINE.json names the metric; the reference exports what it needs -- predicted and GT
    meshes, ``baseline_model.py:365-368``, ``mlp_model.py:708-711`` -- but never computes it).
    Defined like the reference's MPJPE (``metric_utils.py:23-38``: root-relative per hand, L2 per point, / scale):
    both meshes are made relative to their own wrist, the wrist being regressed from the mesh itself with row 0 of
    the MANO joint regressor (``root_weights`` (778,)), then the 778 per-vertex distances."""
    pr = pred_verts.astype(np.float64) - root_weights.astype(np.float64) @ pred_verts.astype(np.float64)
    gr = gt_verts.astype(np.float64) - root_weights.astype(np.float64) @ gt_verts.astype(np.float64)
    return (np.linalg.norm(pr - gr, axis=1) / scale_factor).tolist()


class Evaluator:
    def __init__(self, mano_models=None, data_list=None, image_root=""):
        self.left_hand_faces = None if mano_models is None else mano_models["left"].faces
        self.right_hand_faces = None if mano_models is None else mano_models["right"].faces
        # wrist row of the joint regressor per hand (MPVPE root), (2,778): 0 = right, 1 = left
        self.root_weights = None
        if mano_models is not None and hasattr(mano_models["right"], "J_regressor"):
            jr = lambda m: np.asarray(m.J_regressor.detach().cpu().numpy() if hasattr(m.J_regressor, "detach") else m.J_regressor)[0]
            self.root_weights = np.stack([jr(mano_models["right"]), jr(mano_models["left"])]).astype(np.float32)
        self.data_list = data_list or {}        # list (the reference's dataset.data_list) or dict: data_idx -> metadata
        self.image_root = image_root
        self.save_verts = True
        self.pred_results = []
        self._device_parts = []        # (B,6) float64 device tensors + keep masks, summed lazily
        self._device_vert_parts = []   # (B,2) float64 [sum of per-vertex errors, count]

    def clear(self):
        self.pred_results = []
        self._device_parts = []
        self._device_vert_parts = []

    def update_device(self, pred_joints_3d, gt_joints_3d, collision_loss_origin_scale, keep=None, interacting=None, scale=None):
        """Metrics of one batch straight from device tensors: pred_joints_3d (B,42,3), gt_joints_3d (B,42,4),
        collision_loss_origin_scale (B,1556); ``keep`` (B) bool masks out padding duplicates (evaluator.py:137-146),
        ``interacting`` (B) bool = hand_type == 'interacting' (default all), ``scale`` (B) (default 1)."""
        import ctypes as C

        import torch

        from . import hip
        hip.require_gpu()
        B, dev = pred_joints_3d.shape[0], pred_joints_3d.device
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        p, g, c = f(pred_joints_3d), f(gt_joints_3d), f(collision_loss_origin_scale)
        sc = None if scale is None else f(scale)
        it = None if interacting is None else interacting.to(dev, torch.uint8).contiguous()
        out = torch.empty(B, 6, device=dev, dtype=torch.float64)
        hip.check(hip.lib().ihmr_eval_metrics(p.data_ptr(), g.data_ptr(), c.data_ptr(), None if sc is None else sc.data_ptr(),
                                              None if it is None else it.data_ptr(), B, out.data_ptr(), hip.stream_ptr()), "ihmr_eval_metrics")
        n_inter = torch.ones(B, device=dev, dtype=torch.float64) if it is None else it.to(torch.float64)
        part = torch.cat([out, n_inter[:, None]], dim=1)      # [.., n_interacting]
        if keep is not None:
            part = part * keep.to(dev, torch.float64)[:, None]
        self._device_parts.append(part)

    def update_device_verts(self, pred_right, pred_left, gt_right, gt_left, mano_params_weight, keep=None, scale=None):
        """MPVPE partial sums of one batch from device tensors: the four meshes (B,778,3) and ``mano_params_weight`` (B,2)
        (a hand counts when its MANO annotation exists, i.e. weight > 0).  ``ihmr_eval_mpvpe``."""
        import torch

        from . import hip
        hip.require_gpu()
        assert self.root_weights is not None, "Evaluator needs the MANO models (J_regressor) for MPVPE"
        B, dev = pred_right.shape[0], pred_right.device
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        pr, pl, gr, gl, w = f(pred_right), f(pred_left), f(gt_right), f(gt_left), f(mano_params_weight)
        if not hasattr(self, "_root_w_dev") or self._root_w_dev.device != dev:
            self._root_w_dev = torch.from_numpy(self.root_weights).to(dev).contiguous()
        sc = None if scale is None else f(scale)
        out = torch.empty(B, 4, device=dev, dtype=torch.float64)
        hip.check(hip.lib().ihmr_eval_mpvpe(pr.data_ptr(), pl.data_ptr(), gr.data_ptr(), gl.data_ptr(), self._root_w_dev.data_ptr(),
                                            w.data_ptr(), None if sc is None else sc.data_ptr(), B, out.data_ptr(), hip.stream_ptr()),
                  "ihmr_eval_mpvpe")
        out = out[:, 0:2] + out[:, 2:4]
        if keep is not None:
            out = out * keep.to(dev, torch.float64)[:, None]
        self._device_vert_parts.append(out)

    def gather_pred(self, pred_results):
        self.pred_results += pred_results

    VERT_KEYS = tuple(f"{mode}_{side}_hand_verts" for mode in ("pred", "gt") for side in ("left", "right"))

    def _meta(self, data_idx):
        dl = self.data_list
        if isinstance(dl, dict):
            return dl.get(data_idx, {})
        return dl[data_idx] if 0 <= data_idx < len(dl) else {}

    def update(self, data_idxs, pred_results, save_verts=True, hand_type="interacting", scale=1.0):
        """evaluator.py:38-97.  One record per sample: the exported arrays, the per-sample metadata with the reference's
        defaults (``annot_type`` 'machine', ``hand_type`` 'interacting', ``hand_type_valid`` 1, ``scale`` 1 -- the two
        keyword arguments override the latter defaults for data without a ``data_list``), with ``save_verts`` the meshes
        present in ``pred_results`` stored as **float16** (``:66-72``), the two joint metrics, and for ``do_flip`` samples
        the flip back to the original image (``:100-134``)."""
        self.save_verts = save_verts
        for i, data_idx in enumerate(np.asarray(data_idxs).tolist()):
            meta = self._meta(data_idx)
            rel = meta.get("img_path", f"synthetic/{data_idx:08d}.jpg")
            single = dict(
                data_idx=data_idx, pred_cam_params=pred_results["pred_cam_params"][i], pred_shape_params=pred_results["pred_shape_params"][i],
                pred_pose_params=pred_results["pred_pose_params"][i], pred_hand_trans=pred_results["pred_hand_trans"][i],
                pred_joints_3d=pred_results["pred_joints_3d"][i], collision_loss_origin_scale=pred_results["collision_loss_origin_scale"][i],
                gt_joints_3d=pred_results["gt_joints_3d"][i],
                img_path=rel if not self.image_root else self.image_root.rstrip("/") + "/" + rel, img_path_relative=rel)
            for key, default in (("annot_type", "machine"), ("hand_type", hand_type), ("hand_type_valid", 1.0), ("scale", scale)):
                single[key] = meta.get(key, default)
            if save_verts:
                for key in self.VERT_KEYS:
                    if key in pred_results:
                        single[key] = np.asarray(pred_results[key][i]).astype(np.float16)
            gt = single["gt_joints_3d"]
            single["j3d_error"] = get_single_joints_error(single["pred_joints_3d"], gt[:, :3], gt[:, 3:], single["scale"])
            single["pa_no_rot_inter_j3d_error"] = get_single_pa_inter_joints_error(
                single["pred_joints_3d"], gt[:, :3], gt[:, 3:], single["scale"])
            single["v3d_error"] = []
            if "gt_right_hand_verts" in pred_results and self.root_weights is not None:   # Baseline / MLP exports
                for h, side in enumerate(("right", "left")):
                    if pred_results["mano_params_weight"][i][h] > 0:
                        single["v3d_error"] += get_single_verts_error(pred_results[f"pred_{side}_hand_verts"][i],
                                                                      pred_results[f"gt_{side}_hand_verts"][i],
                                                                      self.root_weights[h], single["scale"])
            if "do_flip" in pred_results and pred_results["do_flip"][i]:
                self._flip_back_data(single)
            self.pred_results.append(single)

    def _flip_back_data(self, single):
        """evaluator.py:100-134: a sample that the loader mirrored (left-only image turned into a right hand) goes back to
        the original image -- camera x and translation x negated, the two hands' pose blocks swapped with the y / z
        axis-angle components negated, joint halves swapped with x negated, the two 778-halves of the per-vertex
        penetration depths swapped, and (with ``save_verts``) the stored meshes swapped and mirrored.  Acts in place on the
        record's arrays like the reference (which are rows of the exported batch arrays); the metrics were taken before."""
        single["pred_cam_params"][1] *= -1
        single["pred_hand_trans"][0] *= -1
        pose = single["pred_pose_params"].copy()
        single["pred_pose_params"][:48], single["pred_pose_params"][48:] = pose[48:], pose[:48]
        single["pred_pose_params"][1::3] *= -1
        single["pred_pose_params"][2::3] *= -1
        for key in ("pred_joints_3d", "gt_joints_3d"):
            j = single[key].copy()
            single[key][:21], single[key][21:] = j[21:], j[:21]
            single[key][:, 0] *= -1
        c = single["collision_loss_origin_scale"].copy()
        single["collision_loss_origin_scale"][:778], single["collision_loss_origin_scale"][778:] = c[778:], c[:778]
        if self.save_verts:
            # the reference indexes all four mesh keys here and raises KeyError when one is missing; this build swaps what
            # is stored (IHMR-OPT exports no GT meshes and never flips: do_flip is all zeros, optimize_model.py:433)
            saved = {k: single[k].copy() for k in self.VERT_KEYS if k in single}
            for key in saved:
                other = key.replace("left", "right") if "left" in key else key.replace("right", "left")
                if other in saved:
                    single[key] = saved[other]
                    single[key][:, 0] *= -1

    def remove_redunc(self):
        """evaluator.py:137-146: drop the padding duplicates (same image id)."""
        seen, out = set(), []
        for d in self.pred_results:
            if d["img_path_relative"] not in seen:
                out.append(d)
                seen.add(d["img_path_relative"])
        self.pred_results = out

    def metric_sums(self):
        """[sum mpjpe, n, sum inter, n, sum coll_ave, sum coll_max, n_interacting, sum mpvpe, n] (float64, additive over ranks)."""
        e = [x for p in self.pred_results for x in p["j3d_error"]]
        pa = [x for p in self.pred_results for x in p["pa_no_rot_inter_j3d_error"]]
        inter = [p for p in self.pred_results if p["hand_type"] == "interacting"]
        ca = [np.mean(p["collision_loss_origin_scale"].astype(np.float64)) * 1000 for p in inter]
        cm = [np.max(p["collision_loss_origin_scale"].astype(np.float64)) * 1000 for p in inter]
        f64 = lambda x: float(np.sum(np.asarray(x, dtype=np.float64)))
        ve = [x for p in self.pred_results for x in p.get("v3d_error", [])]
        sums = np.array([f64(e), len(e), f64(pa), len(pa), f64(ca), f64(cm), len(inter), f64(ve), len(ve)], dtype=np.float64)
        if self._device_parts:
            import torch
            # one reduction over all samples in the order they were added: the float64 sum does not depend on how
            # the samples were grouped into batches / launch sequences
            dsum = torch.cat(self._device_parts, dim=0).sum(dim=0).cpu().numpy()
            sums[:7] = sums[:7] + dsum
        if self._device_vert_parts:
            import torch
            sums[7:9] = sums[7:9] + torch.cat(self._device_vert_parts, dim=0).sum(dim=0).cpu().numpy()
        return sums

    @staticmethod
    def metrics_from_sums(s):
        d = lambda a, b: float(a / b) if b > 0 else float("nan")
        out = dict(mpjpe_3d=d(s[0], s[1]), inter_mpjpe_3d=d(s[2], s[3]), collision_ave=d(s[4], s[6]), collision_max=d(s[5], s[6]))
        if len(s) > 8 and s[8] > 0:        # only the models that export GT meshes (IHMR-Baseline / IHMR-MLP) have it
            out["mpvpe_3d"] = d(s[7], s[8])
        return out

    @property
    def mpjpe_3d(self): return self.metrics_from_sums(self.metric_sums())["mpjpe_3d"]
    @property
    def inter_mpjpe_3d(self): return self.metrics_from_sums(self.metric_sums())["inter_mpjpe_3d"]
    @property
    def collision_ave(self): return self.metrics_from_sums(self.metric_sums())["collision_ave"]
    @property
    def collision_max(self): return self.metrics_from_sums(self.metric_sums())["collision_max"]
    @property
    def mpvpe_3d(self): return self.metrics_from_sums(self.metric_sums()).get("mpvpe_3d", float("nan"))
