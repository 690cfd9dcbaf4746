"""IHMR-MLP training step on the HIP path: what ``src/train_mlp.py:93-99`` does per batch --

    model.set_input(data); model.retrive_prev_prediction(); model.forward()
    model.compute_loss(stage['loss_weights']); model.optimize_parameters()

for the newest sub-network (``models/mlp_model.py:370-405,459-511,514-589``).

* :class:`HeadTrainer` owns one ``InterHandSubNetwork`` in training form: its packed K-major weights and biases live in
  ONE flat device buffer (so does the gradient, so do Adam's moments), the forward keeps the three hidden
  activations, the backward is six fp32 GEMMs on the matrix cores through ``ihmr_conv_igemm``
  (``dX = dY . W`` with the transposed weights, ``dW = X^T . dY`` with the transposed activations) plus the
  ReLU-mask / column-sum / transpose helpers, the update is ``ihmr_adam_step`` on the flat buffer.
* The gradient that enters the head, ``d loss / d final_params`` (B,122), comes from ``ihmr_mlp_train_grad``: the fused
  two-hand forward + collision + joint terms + LBS backward of IHMR-OPT, plus the train-only terms.
* Data parallel training: the flat gradient is summed over the ranks with ONE ``all_reduce`` (RCCL; 3 MB for the
  largest head -- a single bucket, far below what one xGMI link moves in a step) and Adam divides by the world size:
  what ``DistributedDataParallel`` (``mlp_model.py:383-385``) does for this model.

No CPU fallback; PyTorch is used for buffers, indexing of the prev tables and ``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch

from . import hip
from .networks import InterHandSubNetwork, _ceil, _splitk_workspace


def _ldw(n):
    return _ceil(n, 128) if n > 64 else 64


class _Gemm:
    """Argument pack for ``ihmr_conv_igemm`` as a plain GEMM y[M][N] = x[M][K] . w[K][N] (+ bias)."""

    @staticmethod
    def run(x, ldx, M, K, w, ldw, N, bias, y, ldy, act=0):
        ws = _splitk_workspace(x.device)
        hip.check(hip.lib().ihmr_conv_igemm(hip.ptr(x), hip.ptr(w), hip.ptr(bias), None, hip.ptr(y), M, 1, 1, K, 1, 1, N, 1, 1, 1, 0,
                                            ldx, ldw, ldy, 0, act, ws.data_ptr(), ws.numel() * 4, hip.stream_ptr()), "ihmr_conv_igemm")


class HeadTrainer:
    """Training state of one ``InterHandSubNetwork`` (networks.py:83-105) for a fixed batch size."""

    def __init__(self, net: InterHandSubNetwork, batch_size: int, lr: float, device):
        hip.require_gpu()
        self.net, self.B, self.lr, self.dev = net, batch_size, float(lr), device
        lins = [net.regressor[i] for i in (0, 2, 4, 6)]
        self.dims = [(l.in_features, l.out_features) for l in lins]
        self.kpad = [_ceil(i, 16) for i, _ in self.dims]
        self.ldw = [_ldw(o) for _, o in self.dims]
        # ---- one flat buffer: [w0 | b0 | w1 | b1 | ...], w_l = [kpad_l][ldw_l] K-major (the forward layout), b_l padded to ldw_l
        sizes = []
        for kp, ld in zip(self.kpad, self.ldw):
            sizes += [kp * ld, ld]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(self.offsets[-1])
        self.params = torch.zeros(n, device=device)
        self.grads = torch.zeros(n, device=device)
        self.exp_avg = torch.zeros(n, device=device)
        self.exp_avg_sq = torch.zeros(n, device=device)
        self.step = 0
        view = lambda buf, l, which: buf[self.offsets[2 * l + which]:self.offsets[2 * l + which + 1]]
        self.w = [view(self.params, l, 0).view(self.kpad[l], self.ldw[l]) for l in range(4)]
        self.b = [view(self.params, l, 1) for l in range(4)]
        self.gw = [view(self.grads, l, 0).view(self.kpad[l], self.ldw[l]) for l in range(4)]
        self.gb = [view(self.grads, l, 1) for l in range(4)]
        self.load_from_module()
        # ---- activations, their transposes, gradient buffers (rows padded to 16: they serve as GEMM operands)
        B, Bp = batch_size, _ceil(batch_size, 16)
        self.x = [torch.zeros(B, self.kpad[0], device=device)] + [torch.zeros(B, o, device=device) for _, o in self.dims[:3]]
        self.xT = [torch.zeros(self.kpad[l], Bp, device=device) for l in range(4)]
        self.out = torch.zeros(B, self.dims[3][1], device=device)
        self.dy = [torch.zeros(Bp, self.ldw[l], device=device) for l in range(4)]      # gradient w.r.t. layer l's output
        self.wT = [None] + [torch.zeros(_ceil(self.dims[l][1], 16), _ldw(self.dims[l][0]), device=device) for l in (1, 2, 3)]
        self.zero_bias = torch.zeros(max(max(self.ldw), _ldw(self.kpad[0])), device=device)
        self._refresh_transposed_weights()

    # ---- parameters <-> the nn.Module that carries the reference's state_dict keys
    def load_from_module(self):
        for l, lin in enumerate(self.net.regressor[i] for i in (0, 2, 4, 6)):
            i, o = self.dims[l]
            self.w[l].zero_()
            self.w[l][:i, :o].copy_(lin.weight.detach().t().to(self.dev))
            self.b[l].zero_()
            self.b[l][:o].copy_(lin.bias.detach().to(self.dev))

    @torch.no_grad()
    def sync_to_module(self):
        for l, lin in enumerate(self.net.regressor[i] for i in (0, 2, 4, 6)):
            i, o = self.dims[l]
            lin.weight.copy_(self.w[l][:i, :o].t())
            lin.bias.copy_(self.b[l][:o])
        self.net._packed = None

    def _refresh_transposed_weights(self):
        for l in (1, 2, 3):                      # [out][in] = what dX = dY . W reads as its K-major operand
            i, o = self.dims[l]
            hip.check(hip.lib().ihmr_transpose(hip.ptr(self.w[l]), hip.ptr(self.wT[l]), i, o, self.ldw[l], self.wT[l].shape[1],
                                               hip.stream_ptr()), "ihmr_transpose")

    # ---- forward (keeps the activations), networks.py:103-105
    def forward(self, *inputs: torch.Tensor) -> torch.Tensor:
        """``inputs``: the pieces of the input row (img_feat (B,1024), final_params (B,122)), written side by side."""
        B = self.B
        o = 0
        for t in inputs:
            self.x[0][:, o:o + t.shape[1]].copy_(t)
            o += t.shape[1]
        assert o == self.dims[0][0]
        for l in range(4):
            i, o = self.dims[l]
            y = self.x[l + 1] if l < 3 else self.out
            _Gemm.run(self.x[l], self.x[l].shape[1], B, self.kpad[l] if l == 0 else i, self.w[l], self.ldw[l], o, self.b[l], y, o,
                      act=1 if l < 3 else 0)
        return self.out

    # ---- backward from d loss / d output (B, k): fills self.grads
    def backward(self, d_out: torch.Tensor | None = None):
        """d_out (B,k), or None when the caller has already written it into ``self.dy[3]`` (ihmr_mlp_train_grad does)."""
        B, L = self.B, hip.lib()
        st = hip.stream_ptr
        if d_out is not None:
            self.dy[3][:B, :self.dims[3][1]].copy_(d_out)
        for l in (3, 2, 1, 0):
            i, o = self.dims[l]
            kin = self.kpad[l] if l == 0 else i
            # dW[in][out] = X^T [in][B] . dY [B][out]
            hip.check(L.ihmr_transpose(hip.ptr(self.x[l]), hip.ptr(self.xT[l]), B, kin, self.x[l].shape[1], self.xT[l].shape[1], st()),
                      "ihmr_transpose")
            _Gemm.run(self.xT[l], self.xT[l].shape[1], kin, B, self.dy[l], self.ldw[l], o, self.zero_bias, self.gw[l], self.ldw[l])
            hip.check(L.ihmr_colsum(hip.ptr(self.dy[l]), hip.ptr(self.gb[l]), B, o, self.ldw[l], st()), "ihmr_colsum")
            if l > 0:
                # dX[B][in] = dY [B][out] . W^T [out][in], then the ReLU mask of the layer below
                _Gemm.run(self.dy[l], self.ldw[l], B, o, self.wT[l], self.wT[l].shape[1], i, self.zero_bias, self.dy[l - 1], self.ldw[l - 1])
                hip.check(L.ihmr_relu_backward(hip.ptr(self.dy[l - 1]), hip.ptr(self.x[l]), B, i, self.ldw[l - 1], self.x[l].shape[1], st()),
                          "ihmr_relu_backward")

    # ---- DistributedDataParallel's gradient averaging + torch.optim.Adam(lr) (mlp_model.py:383-385,403-405)
    def optimizer_step(self, world_size: int = 1):
        scale = 1.0
        if world_size > 1:
            from .dist import all_reduce_gradients
            scale = all_reduce_gradients(self.grads)
        self.step += 1
        hip.check(hip.lib().ihmr_adam_step(hip.ptr(self.params), hip.ptr(self.grads), hip.ptr(self.exp_avg), hip.ptr(self.exp_avg_sq),
                                           self.params.numel(), scale, self.lr, 0.9, 0.999, 1e-8, self.step, hip.stream_ptr()),
                  "ihmr_adam_step")
        self._refresh_transposed_weights()

    def _to_named(self, buf):
        """A flat buffer as tensors under the reference's parameter names (``regressor.{0,2,4,6}.{weight,bias}``), torch layout."""
        out = OrderedDict()
        for l, idx in enumerate((0, 2, 4, 6)):
            i, o = self.dims[l]
            w = buf[self.offsets[2 * l]:self.offsets[2 * l + 1]].view(self.kpad[l], self.ldw[l])
            out[f"regressor.{idx}.weight"] = w[:i, :o].t().contiguous()
            out[f"regressor.{idx}.bias"] = buf[self.offsets[2 * l + 1]:self.offsets[2 * l + 2]][:o].clone()
        return out

    def _from_named(self, buf, named):
        buf.zero_()
        for l, idx in enumerate((0, 2, 4, 6)):
            i, o = self.dims[l]
            buf[self.offsets[2 * l]:self.offsets[2 * l + 1]].view(self.kpad[l], self.ldw[l])[:i, :o].copy_(named[f"regressor.{idx}.weight"].to(self.dev).t())
            buf[self.offsets[2 * l + 1]:self.offsets[2 * l + 2]][:o].copy_(named[f"regressor.{idx}.bias"])

    def named_gradients(self):
        return self._to_named(self.grads)

    # torch.optim.Adam's own state format: what the reference stores under 'optimizer' (mlp_model.py:834-839)
    def optimizer_state_dict(self):
        m, v = self._to_named(self.exp_avg), self._to_named(self.exp_avg_sq)
        names = list(m)
        state = {i: dict(step=torch.tensor(float(self.step)), exp_avg=m[k].cpu(), exp_avg_sq=v[k].cpu()) for i, k in enumerate(names)} if self.step else {}
        group = dict(lr=self.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, params=list(range(len(names))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, sd):
        names = list(self._to_named(self.grads))
        self.lr = float(sd["param_groups"][0]["lr"])
        self.step = int(float(sd["state"][0]["step"])) if sd["state"] else 0
        if sd["state"]:
            self._from_named(self.exp_avg, {k: sd["state"][i]["exp_avg"] for i, k in enumerate(names)})
            self._from_named(self.exp_avg_sq, {k: sd["state"][i]["exp_avg_sq"] for i, k in enumerate(names)})
        else:
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()


TRAIN_LOSS_NAMES = ("joints_2d_loss", "joints_3d_loss", "mano_pose_loss", "mano_shape_loss", "hand_trans_loss", "shape_reg_loss",
                    "shape_residual_loss", "collision_loss")


class MLPTrainMixin:
    """Training half of ``MLPModel`` (``models/mlp_model.py``); mixed into :class:`ihmr_amd.mlp_model.MLPModel`."""

    def _init_train(self):
        self.isTrain = bool(getattr(self.opt, "isTrain", False))
        self.trainers = {}
        self.world_size = 1
        if getattr(self.opt, "dist", False):
            import torch.distributed as dist
            self.world_size = dist.get_world_size()
        B, dev = self.batch_size, self.device
        self._grad122 = torch.zeros(B, 122, device=dev)
        self._terms5 = torch.zeros(B, 5, device=dev)
        self._out_cols = {}

    def _make_trainer(self, stage_id):
        tr = HeadTrainer(self.sub_network_list[stage_id], self.batch_size, self.strategy[stage_id]["lr"], self.device)
        if self.world_size > 1:                  # DistributedDataParallel starts every rank from rank 0's weights
            import torch.distributed as dist
            dist.broadcast(tr.params, src=0)
            tr._refresh_transposed_weights()
        self.trainers[stage_id] = tr

    # mlp_model.py:408-423
    def retrive_prev_prediction(self):
        idx = self.data_idxs          # (no check that the rows exist: it would stall the host on the device every step)
        self._prev_final = self.prev_final[idx]
        self.img_feat = self.img_feat_all[idx]

    # mlp_model.py:504-511 (+ 442-472)
    def forward(self, forward_backbone=False, stage_id=-1):
        from .mlp_model import COLS, PARAM_DIMS
        if forward_backbone:
            self.final_params = torch.cat([self.init_cam, self.init_pose_params, self.init_shape_params, self.init_hand_trans], dim=1).contiguous()
            self._stage_id = -1
            return
        if stage_id < 0:
            stage_id = len(self.sub_network_list) - 1
        prev = self._prev_final
        tr = self.trainers.get(stage_id) if torch.is_grad_enabled() else None
        if tr is not None:
            res = tr.forward(self.img_feat, prev)
        else:
            if stage_id in self.trainers:
                self.trainers[stage_id].sync_to_module()
            with torch.no_grad():
                res = self.sub_network_list[stage_id](torch.cat([self.img_feat, prev], dim=1))
        new = prev.clone()
        o = 0
        for n in self.strategy[stage_id]["update_params"]:           # mlp_model.py:459-472
            new[:, COLS[n]] += res[:, o:o + PARAM_DIMS[n]]
            o += PARAM_DIMS[n]
        self.final_params, self._stage_id = new, stage_id

    # mlp_model.py:514-583
    def compute_loss(self, loss_weights=None):
        training = loss_weights is not None and torch.is_grad_enabled() and self._stage_id in self.trainers
        if not training:
            loss = self._forward_mano_and_losses(self.final_params)
            self.joints_2d_loss_p_batch, self.joints_3d_loss_p_batch, self.collision_loss_batch = loss[:, 0], loss[:, 1], loss[:, 2]
            self._loss3 = loss
            return
        w = loss_weights
        core, B = self._core, self.batch_size
        io = core.io
        # the training terms compare with the annotation: point the fused kernels' targets at it (restored below)
        keep = (io.init_joints_2d, io.init_joints_3d)
        io.init_joints_2d, io.init_joints_3d = io.gt_joints_2d, io.gt_joints_3d
        try:
            hip.check(hip.lib().ihmr_opt_set_params(C.byref(io), self.final_params.data_ptr(), B, hip.stream_ptr()), "ihmr_opt_set_params")
            ow = hip.OptWeights(w["joints_2d_loss"], w["joints_3d_loss"], 0.0, 0.0, w["collision_loss"], 0.0)
            tw = hip.TrainWeights(w["joints_2d_loss"], w["mano_pose_loss"], w["mano_shape_loss"], w["hand_trans_loss"], w["shape_reg_loss"],
                                  w["shape_residual_loss"])
            mr, ml = core._mano_handles()
            sid = self._stage_id
            if sid not in self._out_cols:               # columns of final_params the stage's sub-network produces, in its output order
                from .mlp_model import COLS
                cols = [c for n in self.strategy[sid]["update_params"] for c in range(COLS[n].start, COLS[n].stop)]
                self._out_cols[sid] = torch.tensor(cols, dtype=torch.int32, device=self.device)
            oc, tr = self._out_cols[sid], self.trainers[sid]
            # (set_input keeps these as slices of the packed 122-vectors: dense copies for the training kernel)
            gt_pose, gt_shape, init_shape = self.gt_pose_params.contiguous(), self.gt_shape_params.contiguous(), self.init_shape_params.contiguous()
            hip.check(hip.lib().ihmr_mlp_train_grad(mr, ml, C.byref(io), B, C.byref(ow), C.byref(tw),
                                                    hip.ptr(gt_pose), hip.ptr(gt_shape), hip.ptr(self.mano_params_weight),
                                                    hip.ptr(init_shape), None, hip.ptr(self._grad122),
                                                    hip.ptr(self._terms5), hip.ptr(oc), oc.numel(), hip.ptr(tr.dy[3]), tr.dy[3].shape[1],
                                                    hip.stream_ptr()), "ihmr_mlp_train_grad")
        finally:
            io.init_joints_2d, io.init_joints_3d = keep
        self._train_w = dict(w)

    def _train_terms(self):
        """The loss terms of the last training compute_loss as device scalars (evaluated on demand: a training step needs
        only the gradient)."""
        lb, t5, w = self._core.buf["loss_batch"], self._terms5.sum(dim=0), self._train_w
        d = OrderedDict(joints_2d_loss=lb[0].mean(), joints_3d_loss=lb[1].mean(), mano_pose_loss=t5[0], mano_shape_loss=t5[1],
                        hand_trans_loss=t5[2], shape_reg_loss=t5[3], shape_residual_loss=t5[4],
                        collision_loss=lb[2].mean() * w["collision_loss"])
        d["loss"] = sum(d.values())
        return d

    @property
    def loss(self):
        return self._train_terms()["loss"]

    # mlp_model.py:586-589
    def optimize_parameters(self):
        tr = self.trainers[self._stage_id]
        tr.backward()                              # dY was written by ihmr_mlp_train_grad
        tr.optimizer_step(self.world_size)

    # mlp_model.py:722-752
    def get_current_errors(self):
        t = self._train_terms()
        d = OrderedDict((n, float(t[n])) for n in TRAIN_LOSS_NAMES)
        d["total_loss"] = float(t["loss"])
        d["hand_type_loss"] = 0.0
        return d

    # mlp_model.py:854-868
    def update_learning_rate(self, epoch, stage_id):
        st = self.strategy[stage_id]
        lr = st["lr"]
        if st.get("lr_decay_type", "none") == "cosine":
            lr = 0.5 * (1.0 + math.cos(math.pi * epoch / self.opt.total_epoch)) * lr
        else:
            assert st.get("lr_decay_type", "none") == "none"
        self.trainers[stage_id].lr = float(lr)
        return lr

    # mlp_model.py:337-356 / 592-637 for the end-of-stage pass of train_mlp.py:125-133
    def save_pred_to_prev(self):
        self._save_prev(self.final_params, self._loss3)

    def select_better_params(self, stage_id):
        from .mlp_model import LOSS_SLOT
        stage = self.strategy[stage_id]
        idx = self.data_idxs
        prev, prev_loss, new, new_loss = self.prev_final[idx], self.prev_loss[idx], self.final_params, self._loss3
        ok = torch.ones(self.batch_size, dtype=torch.bool, device=self.device)
        for name, pct in stage["filter_loss"]:
            c = LOSS_SLOT[name]
            ok &= new_loss[:, c] < prev_loss[:, c] * (1 + float(pct) / 100)
        c = LOSS_SLOT[stage["select_loss"]]
        ok &= new_loss[:, c] <= prev_loss[:, c]
        self.final_params = torch.where(ok[:, None], new, prev)
        self._loss3 = torch.where(ok[:, None], new_loss, prev_loss)
        return ok

    # mlp_model.py:834-839 / base_model.py:23-43: weights + {'epoch', 'optimizer'} in torch's formats
    def save(self, epoch, stage_id):
        import os
        import os.path as osp
        os.makedirs(self.save_dir, exist_ok=True)
        if stage_id in self.trainers:
            self.trainers[stage_id].sync_to_module()
            torch.save(dict(epoch=epoch, optimizer=self.trainers[stage_id].optimizer_state_dict()),
                       osp.join(self.save_dir, f"{epoch}_info_stage_{stage_id:02d}.pth"))
        torch.save({k: v.cpu() for k, v in self.sub_network_list[stage_id].state_dict().items()},
                   osp.join(self.save_dir, f"{epoch}_net_mlp_stage_{stage_id:02d}.pth"))

    def load_checkpoint(self, epoch, stage_id):
        """Resume training of stage `stage_id`: weights (``load``) + the Adam state saved beside them."""
        import os.path as osp
        if not self.load(epoch, stage_id):
            return None
        tr = self.trainers[stage_id]
        tr.load_from_module()
        tr._refresh_transposed_weights()
        info = torch.load(osp.join(self.save_dir, f"{epoch}_info_stage_{stage_id:02d}.pth"), map_location="cpu", weights_only=False)
        tr.load_optimizer_state_dict(info["optimizer"])
        return info["epoch"]
