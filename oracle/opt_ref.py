"""ORACLE (test infrastructure) -- PyTorch-CPU restatement of the IHMR-OPT refinement loop
(reference ``src/models/optimize_model.py``) and of the snapshot filter/select helpers
(``src/utils/opt_utils.py:70-153``).  It is the ``cpu_baseline`` of ``bench.py`` (kind "port") and
the checker of the HIP path; it never runs on the product path.

Pinned by ``tests/golden/opt_traj.npz`` and ``tests/golden/select.npz``: the reference's own
``OptimizeModel`` / ``opt_utils`` run in the build container (with this package's MANO and SDF
restatements injected at the two third-party seams) must give the same trajectory.

Same op graph as the reference: autograd through torch LBS and torch losses,
``torch.optim.Adam(lr, betas=(0.9, 0.999))`` recreated per stage, ``epoch + 1`` iterations per stage,
snapshot before the step every ``save_mid_freq`` iterations, filter + argmin select, final forward
with the default weights.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import losses_ref as L
from .mano_ref import ManoRef
from .sdf_ref import SDFLossRef

TIP_IDS = [744, 320, 443, 554, 671]  # optimize_model.py:99


# ----------------------------------------------------------------------------- opt_utils.py
def gather_params_losses(mid_results, stage, dtype=torch.float32):
    """opt_utils.py:70-98: stack the snapshots -> (S,B,...)  (`.float()` in the reference; `dtype` = float64 for the arbiter run)."""
    names_l = [n for n, _ in stage["filter_loss"]] + [stage["select_loss"]]
    all_params = {n: torch.stack([m[n] for m in mid_results]).to(dtype) for n in stage["update_params"]}
    all_losses = {n: torch.stack([m[n] for m in mid_results]).to(dtype) for n in names_l}
    return all_params, all_losses


def filter_by_losses(all_losses, filter_losses):
    """opt_utils.py:104-141: keep snapshots with loss <= origin * (1 + (c + 0.1)/100) for every
    filter; others -> 1e11; row 0 (the stage's starting point) restored."""
    origin = {n: all_losses[n][0].clone().reshape(1, -1) for n in all_losses}
    first = next(iter(all_losses.values()))
    keep = torch.ones(first.size(), dtype=torch.bool, device=first.device)
    for name, crit in filter_losses:
        assert crit[0] in "+-"
        percent = (float(crit) + 0.1) / 100
        keep = keep & (all_losses[name] <= origin[name] * (1 + percent))
    for n in all_losses:
        all_losses[n][~keep] = 100000000000.0
        all_losses[n][0] = origin[n]
    return all_losses


def select_params(all_params, all_losses, select_loss_name):
    """opt_utils.py:144-153: per-sample argmin over the snapshot axis."""
    idxs = torch.argmin(all_losses[select_loss_name], dim=0)
    ar = torch.arange(idxs.numel(), device=idxs.device).long()
    return {n: p[idxs, ar, ...] for n, p in all_params.items()}, idxs


# ----------------------------------------------------------------------------- optimize_model.py
class OptimizeRef:
    def __init__(self, mano_right_arrays, mano_left_arrays, batch_size, strategy, save_mid_freq=1,
                 default_loss_weights=None, record=False, optimizer="adam", smplx_create=None, sdf_loss_cls=None, device="cpu",
                 dtype=torch.float32):
        """``smplx_create`` / ``sdf_loss_cls`` / ``device``: the two third-party seams of the reference
        (``smplx.create``, ``sdf.SDFLoss``) can be filled with another implementation -- the tests use
        this to run THIS reference-shaped loop (autograd + torch.optim) over the product's seam-A / seam-B modules on the
        GPU, i.e. the import-swap integration route of INTEGRATION.md.  Default: the CPU restatements.
        ``dtype=torch.float64``: the ARBITER -- the same loop, MANO layer and voxel grid in double precision (float32 inputs converted
        exactly): two float32 implementations of 200 Adam steps end a few 1e-5 m apart, and their distances from this run say whether
        that is the arithmetic or an implementation (tests/test_gpu_parity.py, bench.py: parity.vs_f64)."""
        self.batch_size = batch_size
        self.dtype = dtype
        self.device = torch.device(device)
        if smplx_create is None:
            self.mano_right = ManoRef(mano_right_arrays, batch_size=2 * batch_size, dtype=dtype)
            self.mano_left = ManoRef(mano_left_arrays, batch_size=2 * batch_size, dtype=dtype)
        else:   # optimize_model.py:102-108
            self.mano_right = smplx_create("", "mano", use_pca=False, is_rhand=True, batch_size=2 * batch_size)
            self.mano_left = smplx_create("", "mano", use_pca=False, is_rhand=False, batch_size=2 * batch_size)
        # optimize_model.py:109-113 -- flip the left shapedirs x-sign if identical to the right (IN PLACE, before .cuda())
        d = torch.mean(torch.abs(self.mano_left.shapedirs[:, 0, :] - self.mano_right.shapedirs[:, 0, :]))
        if d < 1e-7:
            self.mano_left.shapedirs[:, 0, :] *= -1
        if self.device.type == "cuda":   # optimize_model.py:114-117, loss_utils.py:38
            self.mano_right, self.mano_left = self.mano_right.cuda(), self.mano_left.cuda()
        self.sdf = (sdf_loss_cls or SDFLossRef)(self.mano_right.faces, self.mano_left.faces, robustifier=None)
        if self.device.type == "cuda":
            self.sdf = self.sdf.cuda()
        self.strategy = strategy
        self.save_mid_freq = save_mid_freq
        self.default_loss_weights = default_loss_weights or dict(
            joints_2d_loss=10.0, joints_3d_loss=1000.0, trans_loss_weight=100.0,
            shape_reg_loss_weight=0.1, collision_loss_weight=1.0, finger_reg_loss_weight=100000.0)
        self.record = record
        self.optimizer_name = optimizer
        self.trace = []
        self.selected = []
        self.select_table = []

    # optimize_model.py:120-168
    def set_input(self, data):
        f = lambda k: data[k].detach().clone().float().to(self.device).to(self.dtype)
        self.hand_type_array = f("hand_type_array")
        self.joints_2d = f("joints_2d")
        self.joints_3d = f("joints_3d")
        self.hand_trans = f("hand_trans")
        self.mano_params_weight = f("mano_params_weight")
        self.init_cam = f("init_cam")
        self.init_pose_params = f("init_pose_params")
        self.init_shape_params = f("init_shape_params")
        self.init_hand_trans = f("init_hand_trans")
        self.init_joints_2d = f("init_joints_2d")
        self.init_joints_3d = f("init_joints_3d")
        self.init_hand_trans_j = f("init_hand_trans_j")

    # optimize_model.py:235-251
    def init_optimize(self):
        self.pred_cam_params = self.init_cam.clone()
        self.pred_hand_trans = self.init_hand_trans.clone()[..., :3]
        pose = self.init_pose_params.clone()
        shape = self.init_shape_params.clone()
        self.pred_right_orient = pose[:, :3]
        self.pred_left_orient = pose[:, 48:51]
        self.pred_right_pose_params = pose[:, 3:48]
        self.pred_left_pose_params = pose[:, 51:]
        self.pred_right_shape_params = shape[:, :10]
        self.pred_left_shape_params = shape[:, 10:]

    # optimize_model.py:171-232
    def get_mano_output(self):
        bs = self.batch_size
        lo = self.pred_left_orient.clone()
        lo[:, 1] *= -1
        lo[:, 2] *= -1
        lp = self.pred_left_pose_params.clone().reshape(bs * 15, 3)
        lp[:, 1] *= -1
        lp[:, 2] *= -1
        lp = lp.reshape(bs, 45)
        out = self.mano_right(global_orient=torch.cat([self.pred_right_orient, lo], 0),
                              hand_pose=torch.cat([self.pred_right_pose_params, lp], 0),
                              betas=torch.cat([self.pred_right_shape_params, self.pred_left_shape_params], 0))
        verts = out.vertices
        joints = torch.cat([out.joints, verts[:, TIP_IDS, :]], dim=1)
        rv, rj = verts[:bs], joints[:bs]
        lv, lj = verts[bs:], joints[bs:]
        lv[:, :, 0] *= -1
        lj[:, :, 0] *= -1
        shift = self.pred_hand_trans.view(bs, 1, 3) + (rj[:, 0:1, :] - lj[:, 0:1, :])
        lv = lv + shift
        lj = lj + shift
        return rv, lv, torch.cat([rj, lj], dim=1)

    # optimize_model.py:254-273
    def forward(self):
        self.pred_right_hand_verts, self.pred_left_hand_verts, self.pred_joints_3d = self.get_mano_output()
        self.pred_joints_2d = L.batch_orthogonal_project(self.pred_joints_3d, self.pred_cam_params)
        self.pred_shape_params = torch.cat([self.pred_right_shape_params, self.pred_left_shape_params], 1)
        self.pred_pose_params = torch.cat([self.pred_right_orient, self.pred_right_pose_params,
                                           self.pred_left_orient, self.pred_left_pose_params], 1)

    # optimize_model.py:276-330
    def compute_loss(self, w):
        self.joints_2d_loss, _ = L.joints_2d_loss(self.joints_2d[:, :, :2], self.pred_joints_2d, self.joints_2d[:, :, 2:3])
        l2, l2b = L.joints_2d_loss(self.init_joints_2d[:, :, :2], self.pred_joints_2d, self.init_joints_2d[:, :, 2:3])
        self.joints_2d_loss_p = l2 * w["joints_2d_loss"]
        self.joints_2d_loss_p_batch = l2b * w["joints_2d_loss"]
        loss = self.joints_2d_loss_p

        l3g, _ = L.joints_3d_loss_(self.joints_3d[:, :, :3].clone(), self.pred_joints_3d, self.joints_3d[:, :, 3:4])
        self.joints_3d_loss = l3g * 1000
        l3, l3b = L.joints_3d_loss_(self.init_joints_3d[:, :, :3].clone(), self.pred_joints_3d, self.init_joints_3d[:, :, 3:4])
        self.joints_3d_loss_p = l3 * w["joints_3d_loss"]
        self.joints_3d_loss_p_batch = l3b * w["joints_3d_loss"]
        loss = loss + self.joints_3d_loss_p

        self.hand_trans_loss = L.hand_trans_loss(self.hand_trans[:, :, :3], self.pred_hand_trans, self.hand_trans[:, :, 3:4]) * 10
        self.hand_trans_loss_p = L.hand_trans_loss(self.init_hand_trans_j[:, :, :3], self.pred_hand_trans,
                                                   self.init_hand_trans_j[:, :, 3:4]) * w["trans_loss_weight"]
        loss = loss + self.hand_trans_loss_p

        cl, self.collision_loss_batch, self.collision_loss_origin_scale = L.collision_loss(
            self.sdf, self.pred_right_hand_verts, self.pred_left_hand_verts, self.hand_type_array)
        self.collision_loss = cl * w["collision_loss_weight"]
        loss = loss + self.collision_loss

        self.shape_reg_loss = L.shape_reg_loss(torch.cat((self.pred_right_shape_params, self.pred_left_shape_params), 1)) \
            * w["shape_reg_loss_weight"]
        loss = loss + self.shape_reg_loss

        fl, self.finger_reg_loss_batch = L.finger_reg_loss(self.pred_joints_3d)
        self.finger_reg_loss = fl * w["finger_reg_loss_weight"]
        self.loss = loss + self.finger_reg_loss

    # optimize_model.py:390-415
    def optimize(self):
        for stage in self.strategy:
            params = []
            for name in stage["update_params"]:
                p = getattr(self, name)
                p.requires_grad = True
                params.append(p)
            if self.optimizer_name == "adam":   # optimize_model.py:343-347
                optimizer = torch.optim.Adam(params, lr=stage["lr"], betas=(0.9, 0.999))
            else:
                assert self.optimizer_name == "sgd"
                optimizer = torch.optim.SGD(params, lr=stage["lr"], momentum=0.9)
            mid = []
            for j in range(stage["epoch"] + 1):
                self.forward()
                self.compute_loss(stage["loss_weights"])
                if j % self.save_mid_freq == 0:
                    snap = {n: getattr(self, n).detach().clone() for n in stage["update_params"]}
                    for ln in [n for n, _ in stage["filter_loss"]] + [stage["select_loss"]]:
                        snap[ln] = getattr(self, f"{ln}_batch").detach().clone()
                    mid.append(snap)
                optimizer.zero_grad()
                self.loss.backward()
                if self.record:
                    self.trace.append(dict(
                        loss=float(self.loss.detach()),
                        grads={n: getattr(self, n).grad.detach().clone().cpu().numpy() for n in stage["update_params"]},
                        j3d_batch=self.joints_3d_loss_p_batch.detach().clone().cpu().numpy(),
                        coll_batch=self.collision_loss_batch.detach().clone().cpu().numpy()))
                optimizer.step()
            all_params, all_losses = gather_params_losses(mid, stage, self.dtype)
            all_losses = filter_by_losses(all_losses, stage["filter_loss"])
            sel, idxs = select_params(all_params, all_losses, stage["select_loss"])
            self.selected.append(idxs.cpu().numpy().copy())
            # (S,B) table the argmin ran over: lets a test tell a different choice between numerically tied snapshots from a wrong one
            self.select_table.append(all_losses[stage["select_loss"]].reshape(len(mid), -1).cpu().numpy().copy())
            for n, v in sel.items():
                setattr(self, n, v)
        self.forward()
        self.compute_loss(self.default_loss_weights)

    # optimize_model.py:418-435
    def get_pred_result(self):
        d = lambda t: t.detach().cpu().numpy()
        return OrderedDict(
            pred_cam_params=d(self.pred_cam_params), pred_hand_trans=d(self.pred_hand_trans),
            pred_shape_params=d(self.pred_shape_params), pred_pose_params=d(self.pred_pose_params),
            pred_right_hand_verts=d(self.pred_right_hand_verts), pred_left_hand_verts=d(self.pred_left_hand_verts),
            mano_params_weight=d(self.mano_params_weight), pred_joints_3d=d(self.pred_joints_3d),
            gt_joints_3d=d(self.joints_3d), collision_loss=d(self.collision_loss_batch),
            collision_loss_origin_scale=d(self.collision_loss_origin_scale),
            do_flip=np.zeros(self.batch_size).astype(np.int32), pred_hand_type=np.ones(self.batch_size).astype(np.int32))
