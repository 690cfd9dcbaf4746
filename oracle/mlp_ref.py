"""ORACLE (test infrastructure) -- PyTorch-CPU restatement of the IHMR-MLP inference path, i.e. the
reference's ``MLPModel.test()`` (``src/models/mlp_model.py:683-699``) with ``forward`` (:504-511),
``__update_params_single`` (:459-472), ``__forward_mano`` (:480-501), the selection-relevant part of
``compute_loss`` (:514-583), ``select_better_params`` (:592-637) and the prev tables (:297-356, 408-423).
Pinned by ``tests/golden/mlp_test.npz`` (the reference's own MLPModel run with seeded sub-networks).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import losses_ref as L
from .encoder_ref import InterHandSubNetworkRef
from .mano_ref import ManoRef
from .sdf_ref import SDFLossRef

TIP_IDS = [744, 320, 443, 554, 671]
PARAM_DIMS = dict(pred_hand_trans=3, pred_left_orient=3, pred_right_orient=3, pred_left_pose_params=45, pred_right_pose_params=45,
                  pred_left_shape_params=10, pred_right_shape_params=10, pred_cam_params=3)
DEFAULT_W = dict(joints_2d_loss=10.0, joints_3d_loss=10.0, collision_loss=1.0)   # mlp_model.py:219-231 (selection-relevant)


class MLPRef:
    def __init__(self, mano_right_arrays, mano_left_arrays, batch_size, strategy, num_data):
        self.bs = batch_size
        self.mano_right = ManoRef(mano_right_arrays, batch_size=2 * batch_size)
        left = ManoRef(mano_left_arrays, batch_size=2 * batch_size)
        if torch.mean(torch.abs(left.shapedirs[:, 0, :] - self.mano_right.shapedirs[:, 0, :])) < 1e-7:
            left.shapedirs[:, 0, :] *= -1
        self.sdf = SDFLossRef(self.mano_right.faces, left.faces, robustifier=None)
        self.strategy = strategy
        self.nets = [InterHandSubNetworkRef(1024 + 122, sum(PARAM_DIMS[p] for p in st["update_params"])) for st in strategy]
        self.loss_names = sorted({n for st in strategy for n, _ in st["filter_loss"]} | {st["select_loss"] for st in strategy})
        self.prev_params = {n: torch.zeros(num_data, d) for n, d in PARAM_DIMS.items()}
        self.prev_losses = {n: torch.zeros(num_data) for n in self.loss_names}

    def set_input(self, data):
        f = lambda k: data[k].detach().clone().float()
        self.hand_type_array, self.joints_2d, self.joints_3d = f("hand_type_array"), f("joints_2d"), f("joints_3d")
        self.hand_trans, self.gt_pose_params, self.gt_shape_params = f("hand_trans"), f("mano_pose"), f("mano_betas")
        self.mano_params_weight = f("mano_params_weight")
        self.data_idxs = data["index"].long()
        self.img_feat, self.init_joints_2d, self.init_joints_3d = f("img_feat"), f("init_joints_2d"), f("init_joints_3d")
        self.init_cam, self.init_pose_params = f("init_cam"), f("init_pose_params")
        self.init_shape_params, self.init_hand_trans = f("init_shape_params"), f("init_hand_trans")

    def _mano(self, ro, lo, rp, lp, rs, ls, trans):
        bs = self.bs
        sgn = torch.tensor([1.0, -1.0, -1.0])
        out = self.mano_right(global_orient=torch.cat([ro, lo * sgn], 0),
                              hand_pose=torch.cat([rp, (lp.reshape(bs * 15, 3) * sgn).reshape(bs, 45)], 0),
                              betas=torch.cat([rs, ls], 0))
        verts = out.vertices
        joints = torch.cat([out.joints, verts[:, TIP_IDS, :]], dim=1)
        flip = torch.tensor([-1.0, 1.0, 1.0])
        rv, rj, lv, lj = verts[:bs], joints[:bs], verts[bs:] * flip, joints[bs:] * flip
        shift = trans.view(bs, 1, 3) + (rj[:, 0:1] - lj[:, 0:1])
        return rv, lv + shift, torch.cat([rj, lj + shift], dim=1)

    def _gather(self):
        self.pred_shape_params = torch.cat([self.pred_right_shape_params, self.pred_left_shape_params], 1)
        self.pred_pose_params = torch.cat([self.pred_right_orient, self.pred_right_pose_params, self.pred_left_orient, self.pred_left_pose_params], 1)
        self.final_params = torch.cat([self.pred_cam_params, self.pred_pose_params, self.pred_shape_params, self.pred_hand_trans], 1)

    def _forward_mano(self):
        self.pred_right_hand_verts, self.pred_left_hand_verts, self.pred_joints_3d = self._mano(
            self.pred_right_orient, self.pred_left_orient, self.pred_right_pose_params, self.pred_left_pose_params,
            self.pred_right_shape_params, self.pred_left_shape_params, self.pred_hand_trans)
        self.pred_joints_2d = L.batch_orthogonal_project(self.pred_joints_3d, self.pred_cam_params)
        g = self.gt_pose_params
        self.gt_right_hand_verts, self.gt_left_hand_verts, _ = self._mano(
            g[:, :3], g[:, 48:51], g[:, 3:48], g[:, 51:], self.gt_shape_params[:, :10], self.gt_shape_params[:, 10:], self.hand_trans[:, :, :3])

    def _compute_loss(self, w=DEFAULT_W):
        _, b = L.joints_2d_loss(self.init_joints_2d[:, :, :2], self.pred_joints_2d, self.init_joints_2d[:, :, 2:3])
        self.joints_2d_loss_p_batch = b * w["joints_2d_loss"]
        # NOTE the GT joints are passed WITHOUT clone (mlp_model.py:530-531): root-aligned in place
        L.joints_3d_loss_(self.joints_3d[:, :, :3], self.pred_joints_3d, self.joints_3d[:, :, 3:4])
        _, b = L.joints_3d_loss_(self.init_joints_3d[:, :, :3].clone(), self.pred_joints_3d, self.init_joints_3d[:, :, 3:4])
        self.joints_3d_loss_p_batch = b * w["joints_3d_loss"]
        _, cb, self.collision_loss_origin_scale = L.collision_loss(self.sdf, self.pred_right_hand_verts, self.pred_left_hand_verts, self.hand_type_array)
        self.collision_loss_batch = cb * w["collision_loss"]

    # mlp_model.py:514-583 with a stage's loss weights: the TRAINING objective (every term vs the annotation)
    def compute_train_loss(self, w):
        t = {}
        l, _ = L.joints_2d_loss(self.joints_2d[:, :, :2], self.pred_joints_2d, self.joints_2d[:, :, 2:3])
        t["joints_2d_loss"] = l * w["joints_2d_loss"]
        l, _ = L.joints_3d_loss_(self.joints_3d[:, :, :3], self.pred_joints_3d, self.joints_3d[:, :, 3:4])
        t["joints_3d_loss"] = l * w["joints_3d_loss"]
        g, mw = self.gt_pose_params, self.mano_params_weight
        t["mano_pose_loss"] = (L.mano_pose_loss(g[:, 3:48], self.pred_right_pose_params, mw[:, 0:1])
                               + L.mano_pose_loss(g[:, 51:], self.pred_left_pose_params, mw[:, 1:2])) * w["mano_pose_loss"]
        gs = self.gt_shape_params
        t["mano_shape_loss"] = (L.mano_shape_loss(gs[:, :10], self.pred_right_shape_params, mw[:, 0:1])
                                + L.mano_shape_loss(gs[:, 10:], self.pred_left_shape_params, mw[:, 1:2])) * w["mano_shape_loss"]
        # (B,3) difference times a (B,1,1) weight: the reference broadcasts to (B,B,3) -- mean(w) * mean(d^2) (mlp_model.py:557-558)
        t["hand_trans_loss"] = L.hand_trans_loss(self.hand_trans[:, 0, :3], self.pred_hand_trans, self.hand_trans[:, :, 3:4]) * w["hand_trans_loss"]
        t["shape_reg_loss"] = L.shape_reg_loss(torch.cat([self.pred_right_shape_params, self.pred_left_shape_params], 1)) * w["shape_reg_loss"]
        t["shape_residual_loss"] = (L.shape_residual_loss(self.pred_right_shape_params, self.init_shape_params[:, :10])
                                    + L.shape_residual_loss(self.pred_left_shape_params, self.init_shape_params[:, 10:])) * w["shape_residual_loss"]
        c, _, _ = L.collision_loss(self.sdf, self.pred_right_hand_verts, self.pred_left_hand_verts, self.hand_type_array)
        t["collision_loss"] = c * w["collision_loss"]
        t["loss"] = sum(t.values())
        return t

    # train_mlp.py:93-99 for one batch: retrive_prev_prediction -> forward -> compute_loss(stage weights) -> backward
    def train_forward_backward(self, sid):
        stage = self.strategy[sid]
        for n in PARAM_DIMS:
            setattr(self, n, self.prev_params[n][self.data_idxs].clone())
        self._gather()
        net = self.nets[sid]
        for prm in net.parameters():
            prm.grad = None
        res = net(torch.cat([self.img_feat, self.final_params], dim=1))
        o = 0
        for n in stage["update_params"]:
            setattr(self, n, getattr(self, n) + res[:, o:o + PARAM_DIMS[n]])
            o += PARAM_DIMS[n]
        self._gather()
        self._forward_mano()
        terms = self.compute_train_loss(stage["loss_weights"])
        terms["loss"].backward()
        return {k: float(v.detach()) for k, v in terms.items()}

    def init_prev_from_backbone(self):
        """train_mlp.py:60-66: the backbone prediction fills the "prev" tables."""
        with torch.no_grad():
            p, s = self.init_pose_params.clone(), self.init_shape_params.clone()
            self.pred_cam_params, self.pred_hand_trans = self.init_cam.clone(), self.init_hand_trans.clone()
            self.pred_right_orient, self.pred_left_orient = p[:, :3], p[:, 48:51]
            self.pred_right_pose_params, self.pred_left_pose_params = p[:, 3:48], p[:, 51:]
            self.pred_right_shape_params, self.pred_left_shape_params = s[:, :10], s[:, 10:]
            self._gather()
            self._forward_mano()
            self._compute_loss()
            self._save_prev()

    def _save_prev(self):
        for n in PARAM_DIMS:
            self.prev_params[n][self.data_idxs] = getattr(self, n).clone()
        for n in self.loss_names:
            self.prev_losses[n][self.data_idxs] = getattr(self, n + "_batch").clone()

    def _select(self, stage):
        ok = torch.ones(self.bs, dtype=torch.bool)
        for name, pct in stage["filter_loss"]:
            ok &= getattr(self, name + "_batch") < self.prev_losses[name][self.data_idxs] * (1 + float(pct) / 100)
        sel = stage["select_loss"]
        ok &= getattr(self, sel + "_batch") <= self.prev_losses[sel][self.data_idxs]
        rep = ~ok
        for n in stage["update_params"]:
            p = getattr(self, n)
            p[rep] = self.prev_params[n][self.data_idxs][rep]
        for n in self.loss_names:
            l = getattr(self, n + "_batch")
            l[rep] = self.prev_losses[n][self.data_idxs][rep]
        self._gather()
        self.kept = ok.clone()

    @torch.no_grad()
    def test(self):
        p, s = self.init_pose_params.clone(), self.init_shape_params.clone()
        self.pred_cam_params, self.pred_hand_trans = self.init_cam.clone(), self.init_hand_trans.clone()
        self.pred_right_orient, self.pred_left_orient = p[:, :3], p[:, 48:51]
        self.pred_right_pose_params, self.pred_left_pose_params = p[:, 3:48], p[:, 51:]
        self.pred_right_shape_params, self.pred_left_shape_params = s[:, :10], s[:, 10:]
        self._gather()
        self._forward_mano()
        self._compute_loss()
        self._save_prev()
        self.kept_history = []
        for sid, stage in enumerate(self.strategy):
            for n in PARAM_DIMS:
                setattr(self, n, self.prev_params[n][self.data_idxs].clone())
            self._gather()
            res = self.nets[sid](torch.cat([self.img_feat, self.final_params], dim=1))
            o = 0
            for n in stage["update_params"]:
                setattr(self, n, getattr(self, n) + res[:, o:o + PARAM_DIMS[n]])
                o += PARAM_DIMS[n]
            self._gather()
            self._forward_mano()
            self._compute_loss()
            self._select(stage)
            self.kept_history.append(self.kept.numpy().copy())
            self._save_prev()
        self._forward_mano()
        self._compute_loss()

    def get_pred_result(self):
        n = lambda t: t.detach().cpu().numpy()
        return OrderedDict(
            pred_cam_params=n(self.pred_cam_params), pred_pose_params=n(self.pred_pose_params), pred_shape_params=n(self.pred_shape_params),
            pred_hand_trans=n(self.pred_hand_trans), gt_right_hand_verts=n(self.gt_right_hand_verts), gt_left_hand_verts=n(self.gt_left_hand_verts),
            pred_right_hand_verts=n(self.pred_right_hand_verts), pred_left_hand_verts=n(self.pred_left_hand_verts),
            mano_params_weight=n(self.mano_params_weight), pred_joints_3d=n(self.pred_joints_3d), gt_joints_3d=n(self.joints_3d),
            do_flip=np.zeros(self.bs).astype(np.int32), collision_loss=n(self.collision_loss_batch),
            collision_loss_origin_scale=n(self.collision_loss_origin_scale))
