"""IHMR-Baseline training step on the HIP path: ``src/train_baseline.py:75-80`` per batch --

    model.set_input(data); model.forward(); model.optimize_parameters()

for ``InterHandModel`` (``models/baseline_model.py:257-347``): the encoder in train mode
(:class:`ihmr_amd.encoder_train.EncoderTrainer`: batch-statistics BatchNorm, backward to every parameter, Adam on one flat
buffer, bucketed all-reduces of the flat gradient overlapped with the backward pass for data parallelism), the loss of ``backward_E`` (``:285-341``), and
``torch.optim.Adam(encoder.parameters(), lr=opt.lr)`` (``:69-71``).

The loss gradient ``d loss / d final_params`` comes from ``ihmr_mlp_train_grad`` -- the fused two-hand forward, the joint /
collision terms, the LBS backward and the direct terms (``_mano_pose_loss`` on the 15 finger joints as with
``use_hand_rotation`` off, ``_mano_shape_loss``, ``_hand_trans_loss``, ``_shape_reg_loss``) that the IHMR-MLP training step
uses and that are pinned by ``tests/golden/mlp_train.npz``; the handedness term (``_hand_type_loss``, binary cross entropy
times ``hand_type_valid``) is added here.

Deviations from the reference, all forced by it: ``backward_E`` as checked in raises at its ``_hand_trans_loss`` line (it
unpacks two values from a function that returns one, ``baseline_model.py:322`` vs ``loss_utils.py:114-118``), so no
reference step exists to compare with; the translation term here is the one the MLP stage uses
(``mean_i(w_i) * mean((gt - pred)^2)``); the two hands go through the mirrored right-hand model as in IHMR-OPT / IHMR-MLP
(``optimize_model.py:171-232``), which equals the separate left model of ``baseline_model.py:208-254`` whenever the left
asset is the mirror image of the right one (true for MANO and for the synthetic asset).
"""
from __future__ import annotations

import ctypes as C
import os
import os.path as osp
import types
from collections import OrderedDict

import torch

from . import hip
from .encoder_train import EncoderTrainer
from .optimize_model import OptimizeModel


def hand_type_bce(s, t, valid):
    """loss_utils.py:40-43 per element: F.binary_cross_entropy(s, t, 'none') * valid (both logarithms clamped at -100, as torch)."""
    return -(t * torch.log(s).clamp_min(-100.0) + (1.0 - t) * torch.log(1.0 - s).clamp_min(-100.0)) * valid.reshape(-1, 1)


def hand_type_bce_grad(s, t, valid):
    """d mean(hand_type_bce) / d s."""
    return (s - t) / (s * (1.0 - s)).clamp_min(1e-12) * valid.reshape(-1, 1) / float(s.numel())


class BaselineTrainMixin:
    """Training half of ``InterHandModel``; mixed into :class:`ihmr_amd.baseline_model.InterHandModel`."""

    def _init_train(self):
        opt, B, dev = self.opt, self.batch_size, self.device
        self.world_size = 1
        if getattr(opt, "dist", False):
            import torch.distributed as dist
            self.world_size = dist.get_world_size()
        self.loss_weights = dict(joints_2d=getattr(opt, "joints_2d_loss_weight", 10.0), joints_3d=getattr(opt, "joints_3d_loss_weight", 10.0),
                                 mano_pose=getattr(opt, "pose_param_weight", 10.0), mano_shape=getattr(opt, "shape_param_weight", 10.0),
                                 hand_trans=getattr(opt, "trans_loss_weight", 10.0), shape_reg=getattr(opt, "shape_reg_loss_weight", 0.1),
                                 collision=getattr(opt, "collision_loss_weight", 1.0) if getattr(opt, "use_collision_loss", False) else 0.0)
        core_opt = types.SimpleNamespace(**{**vars(opt), "isTrain": False, "strategy": "opt_default", "save_mid_freq": 1, "optimizer": "adam",
                                            "opt_epoch": 0})
        self._core = OptimizeModel(core_opt)            # buffers, MANO constants and workspace of the fused forward / backward
        self.trainer = EncoderTrainer(self.encoder, B, getattr(opt, "lr", 1e-5), dev)
        if self.world_size > 1:                          # DistributedDataParallel starts every rank from rank 0's weights
            import torch.distributed as dist
            dist.broadcast(self.trainer.flat.params, src=0)
            self.trainer._refresh_derived()
        self._grad122 = torch.zeros(B, 122, device=dev)
        self._terms5 = torch.zeros(B, 5, device=dev)
        self._zeros20 = torch.zeros(B, 20, device=dev)

    # baseline_model.py:257-282 (the MANO forward itself runs inside the fused loss launch of optimize_parameters)
    def forward_train(self):
        self.final_params, self.pred_hand_type = self.trainer.forward(self.input_img)

    # baseline_model.py:285-347
    def optimize_parameters(self):
        core, B, w = self._core, self.batch_size, self.loss_weights
        io, c = core.io, core.buf
        c["gt_joints_2d"].copy_(self.joints_2d)
        c["gt_joints_3d"].copy_(self.joints_3d)
        c["gt_hand_trans"].copy_(self.hand_trans.reshape(B, 4))
        c["hand_type_array"].copy_(self.hand_type_array)
        keep = (io.init_joints_2d, io.init_joints_3d)
        io.init_joints_2d, io.init_joints_3d = io.gt_joints_2d, io.gt_joints_3d      # every term compares with the annotation
        try:
            hip.check(hip.lib().ihmr_opt_set_params(C.byref(io), self.final_params.data_ptr(), B, hip.stream_ptr()), "ihmr_opt_set_params")
            ow = hip.OptWeights(w["joints_2d"], w["joints_3d"], 0.0, 0.0, w["collision"], 0.0)
            tw = hip.TrainWeights(w["joints_2d"], w["mano_pose"], w["mano_shape"], w["hand_trans"], w["shape_reg"], 0.0)
            mr, ml = core._mano_handles()
            hip.check(hip.lib().ihmr_mlp_train_grad(mr, ml, C.byref(io), B, C.byref(ow), C.byref(tw), hip.ptr(self.gt_pose_params),
                                                    hip.ptr(self.gt_shape_params), hip.ptr(self.mano_params_weight), hip.ptr(self._zeros20), None,
                                                    hip.ptr(self._grad122), hip.ptr(self._terms5), None, 0, None, 0, hip.stream_ptr()),
                      "ihmr_mlp_train_grad")
        finally:
            io.init_joints_2d, io.init_joints_3d = keep
        # _hand_type_loss (loss_utils.py:40-43): mean(BCE(pred, gt) * valid) over (B, 2); gradient w.r.t. the sigmoid output as
        # torch's binary_cross_entropy backward forms it, (s - t) / max(s (1 - s), 1e-12): finite when the fp32 sigmoid saturates
        # (s == 0 or 1 exactly), where the plain quotient -(t / s - (1 - t) / (1 - s)) is 0/0
        self._d_hand = hand_type_bce_grad(self.pred_hand_type, self.hand_type_array, self.hand_type_valid)
        self.trainer.backward(self._grad122, self._d_hand)
        self.trainer.optimizer_step(self.world_size)

    # baseline_model.py:378-400 (evaluated on demand)
    def get_current_errors(self):
        lb, t5, w, B = self._core.buf["loss_batch"], self._terms5.sum(dim=0), self.loss_weights, self.batch_size
        bce = hand_type_bce(self.pred_hand_type, self.hand_type_array, self.hand_type_valid)
        d = OrderedDict(hand_type_loss=float(bce.mean()), joints_2d_loss=float(lb[0].mean()), joints_3d_loss=float(lb[1].mean()),
                        mano_pose_loss=float(t5[0]), mano_shape_loss=float(t5[1]), hand_trans_loss=float(t5[2]), shape_reg_loss=float(t5[3]))
        if w["collision"]:
            d["collision_loss"] = float(lb[2].mean()) * w["collision"]
        d["total_loss"] = sum(d.values())
        return d

    def update_learning_rate(self, epoch):
        import math
        lr, kind = getattr(self.opt, "lr", 1e-5), getattr(self.opt, "lr_decay_type", "none")
        if kind == "cosine":
            lr = 0.5 * (1.0 + math.cos(math.pi * epoch / self.opt.total_epoch)) * lr
        self.trainer.lr = float(lr)
        return lr

    # baseline_model.py:491-495 / base_model.py:23-43: weights + {'epoch', 'optimizer'} of torch.optim.Adam, torch formats
    def save(self, label, epoch=None):
        self.trainer.sync_to_module()
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save({k: v.cpu() for k, v in self.encoder.state_dict().items()}, osp.join(self.save_dir, f"{label}_net_baseline.pth"))
        torch.save(dict(epoch=epoch, optimizer=self.trainer.optimizer_state_dict()), osp.join(self.save_dir, f"{label}_info.pth"))

    # opt.continue_train (baseline_model.py:75-85): resume from <which_epoch>_net_baseline.pth + <which_epoch>_info.pth
    def load_checkpoint(self, label):
        if not self.load_network(self.encoder, "baseline", label):
            return None
        self.trainer.load_from_module()
        info = torch.load(osp.join(self.save_dir, f"{label}_info.pth"), map_location="cpu", weights_only=False)
        self.trainer.load_optimizer_state_dict(info["optimizer"])
        return info["epoch"]
