"""``MLPModel`` -- IHMR-MLP inference on the HIP path, with the call surface ``src/test_mlp.py:45-68`` uses:
``MLPModel(opt)``, ``set_update_info(strategy, num_data)``, ``add_new_network(i)``, ``load(epoch, i)``,
``eval()``, ``set_input(data)``, ``test()``, ``get_pred_result()`` (``src/models/mlp_model.py``).

``test()`` (``mlp_model.py:683-699``) = backbone prediction, then for each of the 6 stages: residual MLP on
``[img_feat | final_params]`` (fp32 GEMMs on the matrix cores, :mod:`ihmr_amd.networks`), two-hand MANO +
joint / collision terms (ONE captured launch of the fused kernels that also serve IHMR-OPT,
``ihmr_opt_forward_losses``), keep the update per sample only if every filter loss got strictly better-or-equal
as ``select_better_params`` prescribes (``:592-637``), store to the per-dataset-index "prev" tables
(``:337-356``).  No backward pass.  Training, ``sync`` (pickle gather) and visualisation are out of scope.
"""
from __future__ import annotations

import copy
import os.path as osp
import types
from collections import OrderedDict

import numpy as np
import torch

from . import hip, two_hand
from .networks import InterHandSubNetwork
from .optimize_model import OptimizeModel

PARAM_DIMS = OrderedDict(pred_hand_trans=3, pred_left_orient=3, pred_right_orient=3, pred_left_pose_params=45,
                         pred_right_pose_params=45, pred_left_shape_params=10, pred_right_shape_params=10, pred_cam_params=3)
LOSS_SLOT = dict(joints_2d_loss_p=0, joints_3d_loss_p=1, collision_loss=2)      # rows of ihmr_opt_io.loss_batch


class MLPModel:
    name = "InterHandModel"  # sic, mlp_model.py:26-27

    def __init__(self, opt):
        hip.require_gpu()
        self.opt = opt
        self.inputSize, self.batch_size = opt.inputSize, opt.batchSize
        self.save_dir = getattr(opt, "checkpoints_dir", "./checkpoints")
        core_opt = types.SimpleNamespace(**{**vars(opt), "strategy": "opt_default", "save_mid_freq": 1, "optimizer": "adam", "opt_epoch": 0})
        self._core = OptimizeModel(core_opt)           # buffers, MANO constants, fused forward + losses
        self.mano_models = self._core.mano_models
        self.device = self._core.device
        self.sub_network_list = []
        # mlp_model.py:219-231 -- only the three terms that drive the selection are evaluated at inference
        self.default_loss_weights = dict(joints_2d_loss=10.0, joints_3d_loss=10.0, collision_loss=1.0)
        self._w = dict(joints_2d_loss=10.0, joints_3d_loss=10.0, trans_loss_weight=0.0, shape_reg_loss_weight=0.0,
                       collision_loss_weight=1.0, finger_reg_loss_weight=0.0)

    # mlp_model.py:297-334
    def set_update_info(self, strategy, num_data):
        self.strategy = copy.deepcopy(strategy)
        self.loss_names = sorted({n for st in strategy for n, _ in st["filter_loss"]} | {st["select_loss"] for st in strategy})
        for n in self.loss_names:
            if n not in LOSS_SLOT:
                raise ValueError(f"unsupported filter/select loss {n}")
        dev = self.device
        self.data_idxs_all = torch.zeros(num_data, dtype=torch.bool, device=dev)
        self.img_feat_all = torch.zeros(num_data, 1024, device=dev)
        self.prev_params = {n: torch.zeros(num_data, d, device=dev) for n, d in PARAM_DIMS.items()}
        self.prev_losses = {n: torch.zeros(num_data, device=dev) for n in self.loss_names}

    # mlp_model.py:370-405 (inference part)
    def add_new_network(self, stage_id):
        dim = sum(PARAM_DIMS[p] for p in self.strategy[stage_id]["update_params"])
        self.sub_network_list.append(InterHandSubNetwork(self.opt, self.opt.total_params_dim + 1024, dim).to(self.device))

    def load(self, epoch, stage_id):
        path = osp.join(self.save_dir, f"{epoch}_net_mlp_stage_{stage_id:02d}.pth")
        if not osp.exists(path):
            print(f"{path} does not exist !!!")
            return False
        self.sub_network_list[stage_id].load_state_dict(torch.load(path, map_location="cpu"))
        return True

    def eval(self):
        for n in self.sub_network_list:
            n.eval()
        return self

    # mlp_model.py:156-216
    def set_input(self, input):
        dev, B = self.device, self.batch_size
        g = lambda k: input[k].to(dev, dtype=torch.float32, non_blocking=True)
        c = self._core.buf
        self.hand_type_array = g("hand_type_array")
        c["hand_type_array"].copy_(self.hand_type_array)
        self.joints_2d, self.joints_3d, self.hand_trans = g("joints_2d"), g("joints_3d"), g("hand_trans")
        c["gt_joints_2d"].copy_(self.joints_2d)
        c["gt_joints_3d"].copy_(self.joints_3d)
        c["gt_hand_trans"].copy_(self.hand_trans.reshape(B, 4))
        self.gt_pose_params, self.gt_shape_params, self.mano_params_weight = g("mano_pose"), g("mano_betas"), g("mano_params_weight")
        self.data_idxs = input["index"].to(dev).long()
        self.img_feat = g("img_feat")
        c["init_joints_2d"].copy_(g("init_joints_2d"))
        c["init_joints_3d"].copy_(g("init_joints_3d"))
        c["init_hand_trans_j"].zero_()
        self.init_cam, self.init_pose_params = g("init_cam"), g("init_pose_params")
        self.init_shape_params, self.init_hand_trans = g("init_shape_params"), g("init_hand_trans").reshape(B, 3)

    def _gather(self):  # mlp_model.py:426-439
        self.pred_shape_params = torch.cat([self.pred_right_shape_params, self.pred_left_shape_params], 1)
        self.pred_pose_params = torch.cat([self.pred_right_orient, self.pred_right_pose_params, self.pred_left_orient, self.pred_left_pose_params], 1)
        self.final_params = torch.cat([self.pred_cam_params, self.pred_pose_params, self.pred_shape_params, self.pred_hand_trans], 1)

    def _forward_mano_and_losses(self):
        """__forward_mano + the selection-relevant part of compute_loss: one captured launch of the fused kernels."""
        c = self._core.buf
        c["cam"].copy_(self.pred_cam_params)
        c["trans"].copy_(self.pred_hand_trans)
        c["orient"][0].copy_(self.pred_right_orient); c["orient"][1].copy_(self.pred_left_orient)
        c["pose"][0].copy_(self.pred_right_pose_params); c["pose"][1].copy_(self.pred_left_pose_params)
        c["shape"][0].copy_(self.pred_right_shape_params); c["shape"][1].copy_(self.pred_left_shape_params)
        self._core.forward_losses(self._w)
        lb = c["loss_batch"]
        self.joints_2d_loss_p_batch, self.joints_3d_loss_p_batch, self.collision_loss_batch = lb[0].clone(), lb[1].clone(), lb[2].clone()

    def _save_prev(self):  # mlp_model.py:337-356
        self.data_idxs_all[self.data_idxs] = True
        self.img_feat_all[self.data_idxs] = self.img_feat
        for n in PARAM_DIMS:
            self.prev_params[n][self.data_idxs] = getattr(self, n)
        for n in self.loss_names:
            self.prev_losses[n][self.data_idxs] = getattr(self, n + "_batch")

    def _select_better_params(self, stage):  # mlp_model.py:592-637
        ok = torch.ones(self.batch_size, dtype=torch.bool, device=self.device)
        for name, pct in stage["filter_loss"]:
            ok &= getattr(self, name + "_batch") < self.prev_losses[name][self.data_idxs] * (1 + float(pct) / 100)
        sel = stage["select_loss"]
        ok &= getattr(self, sel + "_batch") <= self.prev_losses[sel][self.data_idxs]
        rep = ~ok
        for n in stage["update_params"]:
            setattr(self, n, torch.where(rep[:, None], self.prev_params[n][self.data_idxs], getattr(self, n)))
        for n in self.loss_names:
            setattr(self, n + "_batch", torch.where(rep, self.prev_losses[n][self.data_idxs], getattr(self, n + "_batch")))
        self.data_idxs_all[self.data_idxs] = False
        self._gather()
        self.kept = ok

    # mlp_model.py:683-699
    @torch.no_grad()
    def test(self):
        p, s = self.init_pose_params, self.init_shape_params
        self.pred_cam_params, self.pred_hand_trans = self.init_cam.clone(), self.init_hand_trans.clone()
        self.pred_right_orient, self.pred_left_orient = p[:, :3].clone(), p[:, 48:51].clone()
        self.pred_right_pose_params, self.pred_left_pose_params = p[:, 3:48].clone(), p[:, 51:].clone()
        self.pred_right_shape_params, self.pred_left_shape_params = s[:, :10].clone(), s[:, 10:].clone()
        self._gather()
        self._forward_mano_and_losses()
        self._save_prev()
        self.kept_history = []
        for sid, stage in enumerate(self.strategy):
            assert bool(torch.all(self.data_idxs_all[self.data_idxs]))
            self.img_feat = self.img_feat_all[self.data_idxs]
            for n in PARAM_DIMS:
                setattr(self, n, self.prev_params[n][self.data_idxs].clone())
            self._gather()
            res = self.sub_network_list[sid](torch.cat([self.img_feat, self.final_params], dim=1))
            o = 0
            for n in stage["update_params"]:
                setattr(self, n, getattr(self, n) + res[:, o:o + PARAM_DIMS[n]])
                o += PARAM_DIMS[n]
            self._gather()
            self._forward_mano_and_losses()
            self._select_better_params(stage)
            self.kept_history.append(self.kept.clone())
            self._save_prev()
        self._forward_mano_and_losses()
        c = self._core.buf
        self.pred_right_hand_verts, self.pred_left_hand_verts = c["verts"][0], c["verts"][1]
        self.pred_joints_3d, self.collision_loss_origin_scale = c["joints_3d"], c["coll_origin_scale"]
        # ground-truth meshes for the export (mlp_model.py:497-501) -- seam-A path
        g = self.gt_pose_params
        self.gt_right_hand_verts, self.gt_left_hand_verts, _ = two_hand.two_hand_forward(
            self.mano_models["right"], g[:, :3], g[:, 48:51], g[:, 3:48], g[:, 51:], self.gt_shape_params[:, :10],
            self.gt_shape_params[:, 10:], self.hand_trans[:, :, :3])

    # mlp_model.py:702-719
    def get_pred_result(self):
        n = lambda t: t.detach().cpu().numpy()
        # the reference root-aligns its GT joint buffer in place (no clone at mlp_model.py:530-531) and exports it
        gt = self.joints_3d.clone()
        w0 = gt[:, 0, 3]
        root = torch.where(w0 > 0.5, 0, torch.where(w0 < 1e-7, 21, -1))
        for b in range(gt.shape[0]):
            if int(root[b]) >= 0:
                gt[b, :, :3] = gt[b, :, :3] - gt[b, int(root[b]):int(root[b]) + 1, :3]
        return OrderedDict(
            pred_cam_params=n(self.pred_cam_params), pred_pose_params=n(self.pred_pose_params), pred_shape_params=n(self.pred_shape_params),
            pred_hand_trans=n(self.pred_hand_trans), gt_right_hand_verts=n(self.gt_right_hand_verts), gt_left_hand_verts=n(self.gt_left_hand_verts),
            pred_right_hand_verts=n(self.pred_right_hand_verts), pred_left_hand_verts=n(self.pred_left_hand_verts),
            mano_params_weight=n(self.mano_params_weight), pred_joints_3d=n(self.pred_joints_3d), gt_joints_3d=n(gt),
            do_flip=np.zeros(self.batch_size).astype(np.int32), collision_loss=n(self.collision_loss_batch),
            collision_loss_origin_scale=n(self.collision_loss_origin_scale))
