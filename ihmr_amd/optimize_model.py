"""Seam C -- ``OptimizeModel``: IHMR-OPT on the fused HIP path, with the call surface of the reference's
``src/models/optimize_model.py`` so that a loop written like ``src/optimize.py:61-71`` runs unchanged:

    model = OptimizeModel(opt); model.set_input(data); model.init_optimize()
    model.optimize(iter_id, num_iter); pred = model.get_pred_result()

``opt`` is a plain namespace with the reference's option names (``options/base_options.py``,
``opt_options.py``): ``batchSize, inputSize, num_joints, total_params_dim, cam_params_dim,
pose_params_dim, shape_params_dim, trans_params_dim, model_root, strategy, save_mid_freq, optimizer,
process_rank`` (+ ``opt_epoch`` to re-parameterise the per-stage iteration count).

The whole refinement (MANO LBS forward/backward for 2B hands, joint / translation / finger / shape
losses, sparse voxel-SDF collision, Adam, snapshots, filter + argmin select) runs in the kernels
behind ``ihmr_opt_run_stage`` -- one C call per stage, no host synchronisation inside a stage and no
GPU->CPU bounce like the reference's ``_finger_reg_loss`` (``loss_utils.py:155,164``).
"""
from __future__ import annotations

import ctypes as C
import os.path as osp
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import hip
from . import mano as mano_shim
from .strategies import OPT_DEFAULT_LOSS_WEIGHTS, get_strategy

def _filter_factor(criterion: str) -> float:
    """utils/opt_utils.py:104-114: bar = origin * (1 + (float(c) + 0.1) / 100), evaluated in float32."""
    if not (isinstance(criterion, str) and criterion and criterion[0] in "+-"):
        raise ValueError(f"filter criterion {criterion!r}: the reference requires a signed percentage string such as '+0' or '-10'")
    return float(np.float32(1 + (float(criterion) + 0.1) / 100))


def _loss_id(name: str, field: str) -> int:
    """utils/opt_utils.py:57-67 + optimize_model.py:366-371: a criterion must not be GT-based and needs a `_batch` twin."""
    if name in ("joints_3d_loss", "joints_2d_loss", "hand_trans_loss"):
        raise ValueError(f"stage['{field}']: '{name}' is computed from the ground truth; the reference (opt_utils.check_valid_loss) "
                         "forbids it as a filter / select criterion")
    if name not in hip.LOSS_IDS:
        raise ValueError(f"stage['{field}']: no per-sample loss '{name}_batch' exists in OptimizeModel "
                         f"(available: {sorted(hip.LOSS_IDS)})")
    return hip.LOSS_IDS[name]


def stage_to_args(stage, optimizer="adam", save_mid_freq=1) -> hip.OptStage:
    """A strategy entry (strategies/opt_default.py) + opt.optimizer / opt.save_mid_freq -> the C ABI's ``ihmr_opt_stage``."""
    mask = 0
    for name in stage["update_params"]:
        if name not in hip.PARAM_BLOCKS:
            raise ValueError(f"stage['update_params']: '{name}' is not a refinable parameter of OptimizeModel "
                             f"(leaf tensors of optimize_model.py:235-251: {sorted(hip.PARAM_BLOCKS)})")
        mask |= hip.PARAM_BLOCKS[name][0]
    if mask == 0:
        raise ValueError("stage['update_params'] is empty")
    if optimizer not in hip.OPTIMIZERS:
        raise ValueError(f"opt.optimizer = {optimizer!r}: the reference knows 'adam' and 'sgd' (optimize_model.py:343-347)")
    if len(stage["filter_loss"]) == 0:
        raise ValueError("stage['filter_loss'] is empty (opt_utils.filter_by_losses asserts at least one criterion)")
    sg = hip.OptStage(param_mask=mask, optimizer=hip.OPTIMIZERS[optimizer], lr=float(stage["lr"]), n_iters=int(stage["epoch"]) + 1,
                      save_freq=int(save_mid_freq), select_loss=_loss_id(stage["select_loss"], "select_loss"))
    for l in range(3):
        sg.use_filter[l], sg.filter_factor[l] = 0, 1.0
    for name, criterion in stage["filter_loss"]:
        l, fac = _loss_id(name, "filter_loss"), _filter_factor(criterion)
        # losses are non-negative, so two criteria on one loss keep what the stricter one keeps
        sg.filter_factor[l] = min(sg.filter_factor[l], fac) if sg.use_filter[l] else fac
        sg.use_filter[l] = 1
    return sg


def _stage_key(sg: hip.OptStage):
    return (sg.param_mask, sg.optimizer, sg.lr, sg.n_iters, sg.save_freq, tuple(sg.use_filter), tuple(sg.filter_factor), sg.select_loss, sg.keep_lists)


def _weights(w) -> hip.OptWeights:
    return hip.OptWeights(w["joints_2d_loss"], w["joints_3d_loss"], w["trans_loss_weight"], w["shape_reg_loss_weight"],
                          w["collision_loss_weight"], w["finger_reg_loss_weight"])


class _PendingResult:
    """Handle of :meth:`OptimizeModel.get_pred_result_async`."""

    def __init__(self, pinned, event, batch_size):
        self._pinned, self._event, self._n = pinned, event, batch_size

    def wait(self):
        self._event.synchronize()
        out = OrderedDict((k, v.numpy()) for k, v in self._pinned.items())
        out["do_flip"] = np.zeros(self._n).astype(np.int32)
        out["pred_hand_type"] = np.ones(self._n).astype(np.int32)
        return out


class OptimizeModel:
    name = "OptimizeModel"

    def __init__(self, opt):
        hip.require_gpu()
        self.opt = opt
        self.process_rank = getattr(opt, "process_rank", -1)
        self.inputSize = opt.inputSize
        # opt.fuse_batches = k: one instance (one launch sequence) carries k consecutive batches of opt.batchSize samples;
        # every sample gets exactly the arithmetic of a batchSize call (ihmr_opt_io.norm_batch), only the launches are shared
        self.fuse_batches = int(getattr(opt, "fuse_batches", 1) or 1)
        self.norm_batch = int(opt.batchSize)
        self.batch_size = self.norm_batch * self.fuse_batches
        assert opt.total_params_dim == opt.cam_params_dim + opt.trans_params_dim + opt.pose_params_dim + opt.shape_params_dim
        self.optimizer = getattr(opt, "optimizer", "adam")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.load_mano_model()
        epoch = getattr(opt, "opt_epoch", None)
        self.strategy = get_strategy(opt.strategy, epoch)
        self.default_loss_weights = dict(OPT_DEFAULT_LOSS_WEIGHTS)
        assert abs(self.default_loss_weights["collision_loss_weight"] - 1.0) < 1e-7  # optimize_model.py:93-94
        self.save_mid_freq = getattr(opt, "save_mid_freq", 1)
        self._alloc(self.batch_size)
        self.selected_history = []
        self.use_graphs = bool(getattr(opt, "use_graphs", True))
        self._graphs = {}

    # optimize_model.py:97-117
    def load_mano_model(self):
        root = getattr(self.opt, "model_root", "") or ""
        models = {}
        for hand_type in ["left", "right"]:
            f = osp.join(root, f"MANO_{hand_type.upper()}.pkl")
            models[hand_type] = mano_shim.create(f, "mano", use_pca=False, is_rhand=(hand_type == "right"),
                                                 batch_size=self.batch_size * 2)
        diff = torch.mean(torch.abs(models["left"].shapedirs[:, 0, :] - models["right"].shapedirs[:, 0, :]))
        if diff < 1e-7:
            models["left"].shapedirs[:, 0, :] *= -1
        self.mano_models = {k: m.to(self.device) for k, m in models.items()}

    def _alloc(self, B):
        dev = self.device
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.S_max = max(st["epoch"] // self.save_mid_freq + 1 for st in self.strategy)
        self.buf = dict(
            cam=z(B, 3), trans=z(B, 3), orient=z(2, B, 3), pose=z(2, B, 45), shape=z(2, B, 10),
            init_joints_2d=z(B, 42, 3), init_joints_3d=z(B, 42, 4), init_hand_trans_j=z(B, 4),
            gt_joints_2d=z(B, 42, 3), gt_joints_3d=z(B, 42, 4), gt_hand_trans=z(B, 4), hand_type_array=z(B, 2),
            verts=z(2, B, 778, 3), joints_3d=z(B, 42, 3), joints_2d=z(B, 42, 2), loss_batch=z(8, B),
            coll_per_vert=z(B, 1556), coll_origin_scale=z(B, 1556),
            snap_params=z(self.S_max, B, hip.OPT_NPARAM), snap_loss=z(self.S_max, 3, B),
            selected=torch.zeros(B, device=dev, dtype=torch.int32), adam_m=z(B, hip.OPT_NPARAM), adam_v=z(B, hip.OPT_NPARAM),
            workspace=torch.empty(hip.lib().ihmr_opt_workspace_bytes(B), device=dev, dtype=torch.uint8),
        )
        # conventions of the collision module (include/ihmr_hip.h: ihmr_sdf_options): opt.sdf_align_corners / opt.sdf_loss_divisor / opt.sdf_swap_xz
        # checker switches of the loop's accelerations (include/ihmr_hip.h: ihmr_opt_io): opt.sdf_no_candidate_lists, opt.sdf_no_static_reuse (exact
        # accelerations: bit-identical either way), opt.sdf_no_translated_reuse (the kept grid of a hand that a stage only translates: rounding-level),
        # opt.no_fused_tail
        self.io = hip.OptIO(**{k: v.data_ptr() for k, v in self.buf.items()}, norm_batch=self.norm_batch,
                            sdf_align_corners=int(bool(getattr(self.opt, "sdf_align_corners", False))),
                            sdf_loss_divisor=float(getattr(self.opt, "sdf_loss_divisor", 0.0) or 0.0),
                            sdf_swap_xz=int(bool(getattr(self.opt, "sdf_swap_xz", False))),
                            sdf_no_candidate_lists=int(bool(getattr(self.opt, "sdf_no_candidate_lists", False))),
                            sdf_no_static_reuse=(1 if getattr(self.opt, "sdf_no_static_reuse", False) else (2 if getattr(self.opt, "sdf_no_translated_reuse", False) else 0)),
                            no_fused_tail=int(bool(getattr(self.opt, "no_fused_tail", False))))
        self.mano_params_weight = z(B, 2)
        self.init = {}
        self._lists_live = False          # run_stage: the workspace holds the candidate lists of this batch's previous stage

    # optimize_model.py:120-168
    def set_input(self, input):
        B = self.batch_size
        dev = self.device
        self._lists_live = False
        g = lambda k: input[k].to(dev, dtype=torch.float32, non_blocking=True)
        assert input["init_cam"].shape[0] == B, "batch size is fixed at construction (opt.batchSize x opt.fuse_batches)"
        self.buf["hand_type_array"].copy_(g("hand_type_array"))
        self.buf["gt_joints_2d"].copy_(g("joints_2d"))
        self.buf["gt_joints_3d"].copy_(g("joints_3d"))
        self.buf["gt_hand_trans"].copy_(g("hand_trans").reshape(B, 4))
        self.mano_params_weight.copy_(g("mano_params_weight"))
        self.buf["init_joints_2d"].copy_(g("init_joints_2d"))
        self.buf["init_joints_3d"].copy_(g("init_joints_3d"))
        self.buf["init_hand_trans_j"].copy_(g("init_hand_trans_j").reshape(B, 4))
        self.init = dict(cam=g("init_cam"), pose=g("init_pose_params"), shape=g("init_shape_params"),
                         trans=g("init_hand_trans").reshape(B, -1)[:, :3])
        self.gt_pose_params = g("mano_pose")
        self.gt_shape_params = g("mano_betas")

    # optimize_model.py:235-251
    def init_optimize(self):
        b, i = self.buf, self.init
        self._lists_live = False
        b["cam"].copy_(i["cam"])
        b["trans"].copy_(i["trans"])
        b["orient"][0].copy_(i["pose"][:, 0:3])
        b["orient"][1].copy_(i["pose"][:, 48:51])
        b["pose"][0].copy_(i["pose"][:, 3:48])
        b["pose"][1].copy_(i["pose"][:, 51:96])
        b["shape"][0].copy_(i["shape"][:, :10])
        b["shape"][1].copy_(i["shape"][:, 10:])

    def _mano_handles(self):
        return self.mano_models["right"]._handle().handle, self.mano_models["left"]._handle().handle

    # optimize_model.py:254-330 with explicit weights (forward + __compute_loss)
    def forward_losses(self, loss_weights=None):
        lw = loss_weights or self.default_loss_weights
        w = _weights(lw)
        mr, ml = self._mano_handles()
        self._lists_live = False          # (a single-shot collision launch on the same workspace)
        if self.use_graphs:
            key = ("fwd",) + tuple(sorted(lw.items()))
            if key not in self._graphs:
                g = C.c_void_p()
                hip.check(hip.lib().ihmr_opt_forward_graph_create(mr, ml, C.byref(self.io), self.batch_size, C.byref(w), C.byref(g)),
                          "ihmr_opt_forward_graph_create")
                self._graphs[key] = g
            hip.check(hip.lib().ihmr_graph_launch(self._graphs[key], hip.stream_ptr()), "ihmr_graph_launch")
            return
        hip.check(hip.lib().ihmr_opt_forward_losses(mr, ml, C.byref(self.io), self.batch_size, C.byref(w), hip.stream_ptr()),
                  "ihmr_opt_forward_losses")

    def collect_sdf_stats(self, loss_weights=None):
        """Work counters of one ``sdf_prep_kernel`` + ``sdf_dist_kernel`` launch pair at the current parameters (diagnostics)."""
        w = _weights(loss_weights or self.default_loss_weights)
        mr, ml = self._mano_handles()
        self._lists_live = False
        out = (C.c_ulonglong * 4)()
        hip.check(hip.lib().ihmr_opt_sdf_stats(mr, ml, C.byref(self.io), self.batch_size, C.byref(w), out, hip.stream_ptr()),
                  "ihmr_opt_sdf_stats")
        return dict(ray_tests=int(out[0]), dist_evals=int(out[1]), inside_voxels=int(out[2]), needed_voxels=int(out[3]))

    SDF_COUNTERS = ("ray_tests", "dist_evals", "inside_voxels", "needed_voxels", "sphere_tests", "voxels_from_lists",
                    "voxels_without_list", "voxels_rebuilt", "plane_tests", "lists_refused", "voxels_full_search")

    def _drop_graphs(self):
        """Destroy every captured graph of this instance (they are re-captured on demand)."""
        for g in self._graphs.values():
            hip.lib().ihmr_graph_destroy(g)
        self._graphs = {}

    def sdf_counters_start(self):
        """Zero the collision kernels' work counters and switch them on for everything launched (or captured) from now on.
        The counter switch is baked into a captured launch, so the instance's graphs are dropped here and again in
        :meth:`sdf_counters_stop`: a graph captured with the counters on never survives into a timed run."""
        hip.check(hip.lib().ihmr_opt_sdf_counters(C.byref(self.io), self.batch_size, None, 1), "ihmr_opt_sdf_counters")
        self._drop_graphs()

    def sdf_counters_stop(self):
        """Synchronise, switch the counters off and return their totals since :meth:`sdf_counters_start` (diagnostics)."""
        out = (C.c_ulonglong * 16)()
        hip.check(hip.lib().ihmr_opt_sdf_counters(C.byref(self.io), self.batch_size, out, 0), "ihmr_opt_sdf_counters")
        self._drop_graphs()
        return {k: int(out[i]) for i, k in enumerate(self.SDF_COUNTERS)}

    def sdf_inside_bits(self):
        """Diagnostics (synchronises): the inside-voxel bitmaps (2, B, 1024) uint32 and boxes (2, B, 4) the collision kernels of the
        last launch left in the workspace (``ihmr_opt_sdf_inside_bits``)."""
        B = self.batch_size
        bits, box = np.zeros((2, B, 1024), np.uint32), np.zeros((2, B, 4), np.float32)
        hip.check(hip.lib().ihmr_opt_sdf_inside_bits(C.byref(self.io), B, bits.ctypes.data, box.ctypes.data), "ihmr_opt_sdf_inside_bits")
        return bits, box

    def run_stage(self, stage):
        sg = stage_to_args(stage, self.optimizer, self.save_mid_freq)
        if (sg.n_iters - 1) // sg.save_freq + 1 > self.S_max:
            raise ValueError("stage takes more snapshots than the ring allocated at construction holds")
        # ihmr_opt_stage.keep_lists: the collision kernels' candidate lists of the previous stage stay valid across the stage boundary when
        # THIS instance ran that stage on THIS batch and has launched nothing else on its workspace since (set_input, init_optimize and
        # every other launch on the workspace clear the flag).  An exact acceleration: opt.sdf_no_stage_list_reuse is the checker switch
        sg.keep_lists = int(self._lists_live and not getattr(self.opt, "sdf_no_stage_list_reuse", False))
        self._lists_live = True
        w = _weights(stage["loss_weights"])
        mr, ml = self._mano_handles()
        if self.use_graphs:
            key = ("stage",) + _stage_key(sg) + tuple(sorted(stage["loss_weights"].items()))
            if key not in self._graphs:
                g = C.c_void_p()
                hip.check(hip.lib().ihmr_opt_stage_graph_create(mr, ml, C.byref(self.io), self.batch_size, C.byref(w), C.byref(sg),
                                                                C.byref(g)), "ihmr_opt_stage_graph_create")
                self._graphs[key] = g
            hip.check(hip.lib().ihmr_graph_launch(self._graphs[key], hip.stream_ptr()), "ihmr_graph_launch")
            return
        hip.check(hip.lib().ihmr_opt_run_stage(mr, ml, C.byref(self.io), self.batch_size, C.byref(w), C.byref(sg), hip.stream_ptr()),
                  "ihmr_opt_run_stage")

    # optimize_model.py:390-415
    def optimize(self, iter_id=0, num_iter=1, verbose=False):
        self.selected_history = []
        for stage_id, stage in enumerate(self.strategy):
            self.run_stage(stage)
            self.selected_history.append(self.buf["selected"].clone())
            if verbose and self.process_rank <= 0:
                print(f"iter:{iter_id + 1:04d}/{num_iter:04d}, stage-{stage_id:02d} completes")
                sys.stdout.flush()
        self.forward_losses(self.default_loss_weights)

    def __del__(self):
        try:
            if sys is None or sys.is_finalizing():
                return
            self._drop_graphs()
        except Exception:
            pass

    # reference-named views of the state
    @property
    def pred_cam_params(self): return self.buf["cam"]
    @property
    def pred_hand_trans(self): return self.buf["trans"].view(-1, 1, 3)
    @property
    def pred_shape_params(self): return torch.cat([self.buf["shape"][0], self.buf["shape"][1]], dim=1)
    @property
    def pred_pose_params(self):
        b = self.buf
        return torch.cat([b["orient"][0], b["pose"][0], b["orient"][1], b["pose"][1]], dim=1)
    @property
    def pred_right_hand_verts(self): return self.buf["verts"][0]
    @property
    def pred_left_hand_verts(self): return self.buf["verts"][1]
    @property
    def pred_joints_3d(self): return self.buf["joints_3d"]
    @property
    def pred_joints_2d(self): return self.buf["joints_2d"]
    @property
    def collision_loss_batch(self): return self.buf["loss_batch"][2]
    @property
    def collision_loss_origin_scale(self): return self.buf["coll_origin_scale"]
    @property
    def joints_3d_loss_p_batch(self): return self.buf["loss_batch"][1]
    @property
    def joints_2d_loss_p_batch(self): return self.buf["loss_batch"][0]

    def _export_sources(self):
        return OrderedDict(
            pred_cam_params=self.pred_cam_params, pred_hand_trans=self.pred_hand_trans,
            pred_shape_params=self.pred_shape_params, pred_pose_params=self.pred_pose_params,
            pred_right_hand_verts=self.pred_right_hand_verts, pred_left_hand_verts=self.pred_left_hand_verts,
            mano_params_weight=self.mano_params_weight, pred_joints_3d=self.pred_joints_3d,
            gt_joints_3d=self.buf["gt_joints_3d"], collision_loss=self.collision_loss_batch,
            collision_loss_origin_scale=self.collision_loss_origin_scale)

    # optimize_model.py:418-435
    def get_pred_result(self):
        n = lambda t: t.detach().cpu().numpy()
        out = OrderedDict((k, n(v)) for k, v in self._export_sources().items())
        out["do_flip"] = np.zeros(self.batch_size).astype(np.int32)
        out["pred_hand_type"] = np.ones(self.batch_size).astype(np.int32)
        return out

    def get_pred_result_async(self):
        """``get_pred_result()`` without stalling the host: the device-to-host copies go, on the current stream and behind
        the refinement that produced the values, into pinned host buffers (two alternating sets per instance); the
        returned handle's ``wait()`` blocks until they have landed and hands out the same OrderedDict.  The arrays
        are views of the pinned buffers: valid until the next-but-one export of this instance."""
        if not hasattr(self, "_pinned"):
            self._pinned, self._pin_slot = [None, None], 0
        self._pin_slot ^= 1
        src = self._export_sources()
        if self._pinned[self._pin_slot] is None:
            self._pinned[self._pin_slot] = OrderedDict((k, torch.empty(v.shape, dtype=v.dtype, pin_memory=True)) for k, v in src.items())
        dst = self._pinned[self._pin_slot]
        for k, v in src.items():
            dst[k].copy_(v, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _PendingResult(dst, ev, self.batch_size)

    # optimize_model.py:437-455 (means of the per-sample values)
    def get_current_errors(self):
        lb = self.buf["loss_batch"].mean(dim=1).cpu().numpy()
        return OrderedDict(joints_2d_loss=float(lb[4]), joints_3d_loss=float(lb[5]) * 1000, hand_trans_loss=float(lb[7]) * 10,
                           collision_loss=float(lb[2]), joints_3d_loss_p=float(lb[1]))
