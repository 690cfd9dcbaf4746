"""``MLPModel`` -- IHMR-MLP inference on the HIP path, with the call surface ``src/test_mlp.py:45-68`` uses:
``MLPModel(opt)``, ``set_update_info(strategy, num_data)``, ``add_new_network(i)``, ``load(epoch, i)``,
``eval()``, ``set_input(data)``, ``test()``, ``get_pred_result()`` (``src/models/mlp_model.py``).

``test()`` (``mlp_model.py:683-699``) = backbone prediction, then for each of the 6 stages: residual MLP on
``[img_feat | final_params]`` (fp32 GEMMs on the matrix cores, :mod:`ihmr_amd.networks`), two-hand MANO +
joint / collision terms (ONE captured launch of the fused kernels that also serve IHMR-OPT,
``ihmr_opt_forward_losses``), keep the update per sample only if every filter loss got strictly better-or-equal
as ``select_better_params`` prescribes (``:592-637``), store to the per-dataset-index "prev" tables
(``:337-356``).  The training step (``forward / compute_loss / optimize_parameters``, ``src/train_mlp.py:93-99``) is in
:mod:`ihmr_amd.mlp_train`; ``sync`` (pickle gather) and visualisation are out of scope.
"""
from __future__ import annotations

import copy
import ctypes as C
import os.path as osp
import types
from collections import OrderedDict

import numpy as np
import torch

from . import hip, two_hand
from .networks import InterHandSubNetwork
from .mlp_train import MLPTrainMixin
from .optimize_model import OptimizeModel

PARAM_DIMS = OrderedDict(pred_hand_trans=3, pred_left_orient=3, pred_right_orient=3, pred_left_pose_params=45,
                         pred_right_pose_params=45, pred_left_shape_params=10, pred_right_shape_params=10, pred_cam_params=3)
# columns of the packed prediction vector final_params (B,122) (mlp_model.py:426-439): the whole refinement state
COLS = OrderedDict(pred_cam_params=slice(0, 3), pred_right_orient=slice(3, 6), pred_right_pose_params=slice(6, 51),
                   pred_left_orient=slice(51, 54), pred_left_pose_params=slice(54, 99), pred_right_shape_params=slice(99, 109),
                   pred_left_shape_params=slice(109, 119), pred_hand_trans=slice(119, 122))
LOSS_SLOT = dict(joints_2d_loss_p=0, joints_3d_loss_p=1, collision_loss=2)      # rows of ihmr_opt_io.loss_batch


class MLPModel(MLPTrainMixin):
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
        self._init_train()
        # test() as ONE hipGraph per instance (opt.use_test_graph, default on): ~150 small launches (6 MLPs, 8 fused MANO + collision
        # passes, the per-sample selects, the "prev" table scatters) are captured at the first call and replayed afterwards; the
        # inputs live in static device buffers that set_input() refills.  Eager whenever sub-networks are being trained here.
        self.use_test_graph = bool(getattr(opt, "use_test_graph", True))
        self._in, self._test_graph, self._graph_sig, self._init_packed, self._gt_packed, self._glue_cache = {}, None, None, None, None, None

    # mlp_model.py:297-334
    def set_update_info(self, strategy, num_data):
        self.strategy = copy.deepcopy(strategy)
        self.loss_names = sorted({n for st in strategy for n, _ in st["filter_loss"]} | {st["select_loss"] for st in strategy})
        for n in self.loss_names:
            if n not in LOSS_SLOT:
                raise ValueError(f"unsupported filter/select loss {n}")
        dev = self.device
        # "prev" tables indexed by dataset index (mlp_model.py:337-356), kept packed: one row per sample
        self.data_idxs_all = torch.zeros(num_data, dtype=torch.bool, device=dev)
        self.img_feat_all = torch.zeros(num_data, 1024, device=dev)
        self.prev_final = torch.zeros(num_data, 122, device=dev)
        self.prev_loss = torch.zeros(num_data, 3, device=dev)          # columns = LOSS_SLOT
        self._test_graph = None     # a captured test() holds the OLD tables' addresses, column slices and thresholds

    # mlp_model.py:370-405 (inference part)
    def add_new_network(self, stage_id):
        dim = sum(PARAM_DIMS[p] for p in self.strategy[stage_id]["update_params"])
        self.sub_network_list.append(InterHandSubNetwork(self.opt, self.opt.total_params_dim + 1024, dim).to(self.device))
        self._test_graph = None
        if self.isTrain:                                   # mlp_model.py:402-405: a fresh Adam(lr) for the new sub-network
            self._make_trainer(stage_id)

    def load(self, epoch, stage_id):
        path = osp.join(self.save_dir, f"{epoch}_net_mlp_stage_{stage_id:02d}.pth")
        if not osp.exists(path):
            print(f"{path} does not exist !!!")
            return False
        self.sub_network_list[stage_id].load_state_dict(torch.load(path, map_location="cpu"))
        self._test_graph = None
        return True

    def eval(self):
        for n in self.sub_network_list:
            n.eval()
        return self

    # mlp_model.py:156-216
    def set_input(self, input):
        """The batch's 17 input tensors reach their final device buffers -- the fused kernels' target buffers, the packed 122-vectors
        of the backbone's prediction (mlp_model.py:204-216 order) and of the annotation, the image features -- through ONE launch
        (``ihmr_copy_segments``: a table of 2-D strided copies; round 4: 17 copy launches, 22 % of the GPU time of a batch together
        with the export's).  Host tensors are packed into one pinned staging buffer first and cross PCIe in one copy.  All
        destinations are allocated once (a captured test() replays over the same addresses)."""
        B, dev = self.batch_size, self.device
        c = self._core.buf
        if self._init_packed is None:
            z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
            self._init_packed, self._gt_packed = z(B, 122), z(B, 122)
            self._in = dict(index=z(B, dt=torch.long), img_feat=z(B, 1024), mano_params_weight=z(B, 2))
            c["init_hand_trans_j"].zero_()          # (no 3-D translation target from the backbone at inference, mlp_model.py:180-183)
            self._test_graph = None
        ip, gp = self._init_packed, self._gt_packed
        plan = [(c["hand_type_array"], "hand_type_array"), (c["gt_joints_2d"], "joints_2d"), (c["gt_joints_3d"], "joints_3d"),
                (c["gt_hand_trans"], "hand_trans"), (c["init_joints_2d"], "init_joints_2d"), (c["init_joints_3d"], "init_joints_3d"),
                (ip[:, 0:3], "init_cam"), (ip[:, 3:99], "init_pose_params"), (ip[:, 99:119], "init_shape_params"),
                (ip[:, 119:122], "init_hand_trans"), (gp[:, 3:99], "mano_pose"), (gp[:, 99:119], "mano_betas"),
                (self._in["index"], "index"), (self._in["img_feat"], "img_feat"), (self._in["mano_params_weight"], "mano_params_weight")]
        want = {k: dst.dtype for dst, k in plan}      # a loader may yield an int32 index, float64 joints, fp16 features: converted,
        srcs = self._stage_inputs({k for _, k in plan} | {"hand_trans"},      # as the reference's FloatTensor.copy_() / .long() do
                                  {k: (v if k not in want or v.dtype == want[k] else v.to(want[k])) for k, v in input.items()})
        pairs = [(srcs[k].reshape(dst.shape) if dst.is_contiguous() else srcs[k].reshape(dst.shape[0], -1), dst) for dst, k in plan]
        pairs.append((srcs["hand_trans"].reshape(B, 4)[:, :3], gp[:, 119:122]))
        hip.copy_segments(pairs)
        # the reference's attribute names, as views of those buffers
        self.hand_type_array, self.joints_2d, self.joints_3d = c["hand_type_array"], c["gt_joints_2d"], c["gt_joints_3d"]
        self.hand_trans = c["gt_hand_trans"].view(B, 1, 4)
        self.gt_pose_params, self.gt_shape_params, self.mano_params_weight = gp[:, 3:99], gp[:, 99:119], self._in["mano_params_weight"]
        self.data_idxs, self.img_feat = self._in["index"], self._in["img_feat"]
        self.init_cam, self.init_pose_params, self.init_shape_params, self.init_hand_trans = ip[:, 0:3], ip[:, 3:99], ip[:, 99:119], ip[:, 119:122]

    def _stage_inputs(self, keys, input):
        """Device-resident, contiguous sources for ``keys``: device tensors as they are; host tensors packed into ONE pinned buffer
        (allocated once) and sent in ONE host-to-device copy."""
        out, host = {}, []
        for k in sorted(keys):
            t = input[k]
            if t.is_cuda:
                out[k] = t if t.is_contiguous() else t.contiguous()
            else:
                host.append(k)
        if host:
            sizes = [(k, input[k].numel() * input[k].element_size()) for k in host]
            total = sum((n + 15) // 16 * 16 for _, n in sizes)
            if getattr(self, "_stage_host", None) is None or self._stage_host.numel() < total:
                self._stage_host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
                self._stage_dev = torch.empty(total, dtype=torch.uint8, device=self.device)
            if getattr(self, "_stage_ev", None) is not None:
                self._stage_ev.synchronize()             # the previous batch's copy has left the pinned buffer
            off = 0
            for k, n in sizes:
                self._stage_host[off:off + n].copy_(input[k].contiguous().view(-1).view(torch.uint8))
                out[k] = self._stage_dev[off:off + n].view(input[k].dtype).view(input[k].shape)
                off += (n + 15) // 16 * 16
            self._stage_dev[:total].copy_(self._stage_host[:total], non_blocking=True)
            self._stage_ev = torch.cuda.Event()
            self._stage_ev.record()
        return out

    # reference-named views of the packed state (mlp_model.py:426-439)
    @property
    def pred_pose_params(self): return self.final_params[:, 3:99]
    @property
    def pred_shape_params(self): return self.final_params[:, 99:119]

    def _forward_mano_and_losses(self, final):
        """__forward_mano + the selection-relevant part of compute_loss for the packed state `final` (B,122): one
        scatter kernel + one captured launch of the fused kernels; returns the three per-sample losses (B,3).  (Training path and
        diagnostics; test() goes through the device-side glue below.)"""
        hip.check(hip.lib().ihmr_opt_set_params(C.byref(self._core.io), final.data_ptr(), self.batch_size, hip.stream_ptr()),
                  "ihmr_opt_set_params")
        self._core.forward_losses(self._w)
        return self._core.buf["loss_batch"][:3].t().contiguous()

    def _save_prev(self, final, loss):  # mlp_model.py:337-356
        idx = self.data_idxs
        self.data_idxs_all.index_fill_(0, idx, True)      # (no host scalar: stays capturable in a graph)
        self.img_feat_all[idx] = self.img_feat
        self.prev_final[idx] = final
        self.prev_loss[idx] = loss

    # ---- device-side glue of test(): everything between two fused forward passes is ONE launch (csrc/mlp_infer.h)
    def _glue(self):
        """Static buffers and the ctypes descriptors of the stage launches (built once per capture of test(): they hold the
        addresses of the "prev" tables, the sub-networks' packed weights and the strategy's criteria)."""
        sig = self._test_signature()
        g = getattr(self, "_glue_cache", None)
        if g is not None and g["sig"] == sig:
            return g
        B, dev, L = self.batch_size, self.device, hip.lib()
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
        g = dict(sig=sig, new=z(B, 122), final=z(B, 122), kept=z(len(self.strategy), B, dt=torch.uint8), kept0=z(B, dt=torch.uint8),
                 ws=z(L.ihmr_mlp_workspace_bytes(B), dt=torch.uint8), nets=[], stages=[], tables=[])
        # the annotation's meshes: a second parameter set skinned by the same kernels into its own vertex buffer
        gtb = dict(cam=z(B, 3), trans=z(B, 3), orient=z(2, B, 3), pose=z(2, B, 45), shape=z(2, B, 10), verts=z(2, B, 778, 3))
        io = hip.OptIO.from_buffer_copy(self._core.io)
        for k, v in gtb.items():
            setattr(io, k, v.data_ptr())
        g["gt_buf"], g["gt_io"] = gtb, io
        tab = lambda new, kept: hip.MlpTables(
            idx=self.data_idxs.data_ptr(), data_idxs_all=self.data_idxs_all.data_ptr(), img_feat_all=self.img_feat_all.data_ptr(),
            prev_final=self.prev_final.data_ptr(), prev_loss=self.prev_loss.data_ptr(), img_feat=self.img_feat.data_ptr(),
            new_params=new.data_ptr(), final_params=g["final"].data_ptr(), kept=kept.data_ptr())
        g["first"] = tab(self._init_packed, g["kept0"])
        for sid, stage in enumerate(self.strategy):
            pk = self.sub_network_list[sid].packed(dev)
            net = hip.MlpNet()
            for l in range(4):
                net.w[l], net.b[l], net.ldw[l] = pk[l].w.data_ptr(), pk[l].b.data_ptr(), pk[l].ldw
            cols = [c for n in stage["update_params"] for c in range(COLS[n].start, COLS[n].stop)]     # mlp_model.py:459-472 order
            net.k_out = len(cols)
            for j, c in enumerate(cols):
                net.col[j] = c
            st = hip.MlpStage(n_filter=len(stage["filter_loss"]), select_loss=LOSS_SLOT[stage["select_loss"]])
            for f, (name, pct) in enumerate(stage["filter_loss"]):
                st.filter_loss[f] = LOSS_SLOT[name]
                st.filter_factor[f] = float(np.float32(1 + float(pct) / 100))        # torch multiplies the float32 loss by this scalar
            g["nets"].append(net); g["stages"].append(st); g["tables"].append(tab(g["new"], g["kept"][sid]))
            g.setdefault("keep", []).append(pk)
            # a stage that moves nothing but the camera (mlp_default's last) changes no vertex, no 3-D joint and no penetration depth:
            # judged by ONE small launch on the accepted state's joints (ihmr_mlp_camera_select; opt.mlp_no_camera_shortcut = the
            # reference's full re-evaluation, the checker)
            g.setdefault("cam_only", []).append(all(c < 3 for c in cols) and not getattr(self.opt, "mlp_no_camera_shortcut", False))
            # does the stage touch a column v_posed depends on -- finger pose (6..50, 54..98 of the 122-vector) or shape (99..118)?
            g.setdefault("moves_vposed", []).append(any(6 <= c < 51 or 54 <= c < 119 for c in cols))
        self._glue_cache = g
        return g

    @property
    def kept_history(self):
        """Per stage, which samples kept the stage's update (select_better_params' decisions)."""
        return [k.bool() for k in self._glue_cache["kept"]] if getattr(self, "_glue_cache", None) is not None else []

    # mlp_model.py:683-699
    @torch.no_grad()
    def test(self):
        for tr in self.trainers.values():             # weights trained in this process: refresh the modules' copies
            tr.sync_to_module()
        if not self.use_test_graph or self.trainers:
            return self._test_eager()
        sig = self._test_signature()
        if self._test_graph is None or self._graph_sig != sig:
            self._test_eager()                               # untimed first pass: lazy allocations, the core's own graphs
            torch.cuda.synchronize()
            graph, core_graphs = torch.cuda.CUDAGraph(), self._core.use_graphs
            self._core.use_graphs = False                    # the fused kernels are captured directly, not as a nested graph launch
            try:
                with torch.cuda.graph(graph):
                    self._test_eager()
            finally:
                self._core.use_graphs = core_graphs
            self._test_graph, self._graph_sig = graph, sig
        self._test_graph.replay()

    def _test_signature(self):
        """Everything a captured test() has baked in: the sub-networks (and their packed weights), the addresses of the four "prev"
        tables, the strategy's contents (update columns, filter percentages, select losses) and the loss weights."""
        strat = tuple((tuple(st["update_params"]), tuple((n, str(p)) for n, p in st["filter_loss"]), st["select_loss"]) for st in self.strategy)
        tables = tuple(t.data_ptr() for t in (self.data_idxs_all, self.img_feat_all, self.prev_final, self.prev_loss))
        nets = tuple((id(n), tuple(p.w.data_ptr() for p in n.packed(self.device))) for n in self.sub_network_list)
        inputs = tuple(t.data_ptr() for t in (self.data_idxs, self.img_feat, self._init_packed, self._gt_packed))
        return (strat, tables, tuple(sorted(self._w.items())), nets, inputs)

    def _test_eager(self):
        """test() as 50 kernel launches and nothing else: [unpack + forward/select] for the backbone's prediction, [sub-network head +
        forward/select] per stage, [unpack + forward] of the final state, [unpack + skeleton + skinning] of the annotation."""
        L, B, st = hip.lib(), self.batch_size, hip.stream_ptr()
        g = self._glue()
        core = self._core
        from .optimize_model import _weights
        io, w = C.byref(core.io), _weights(self._w)
        mr, ml = core._mano_handles()
        # mlp_model.py:204-216: [cam | pose 96 | shape 20 | trans] in the reference's order = self._init_packed (filled by set_input)
        hip.check(L.ihmr_opt_set_params(io, self._init_packed.data_ptr(), B, st), "ihmr_opt_set_params")
        hip.check(L.ihmr_mlp_forward_select(mr, ml, io, B, C.byref(w), C.byref(g["first"]), None, 1, g["ws"].data_ptr(), st), "ihmr_mlp_forward_select")
        # v_posed (blend shapes applied, before skinning) depends on finger pose and shape only.  The evaluation that opens test() skins
        # the backbone's prediction in full; until a stage proposes new finger poses or shapes every row of every evaluated state --
        # kept or fallen back -- still has the prediction's, i.e. the workspace's v_posed is the bits a recomputation would give and
        # the skinning launch may skip both blends (mode 3; opt.mlp_no_vposed_reuse: the checker).  From the first such stage on the
        # stored v_posed is a PROPOSAL's, which a sample may have rejected: full skinning.
        vposed_ok = not getattr(self.opt, "mlp_no_vposed_reuse", False)
        for sid in range(len(self.strategy)):
            hip.check(L.ihmr_mlp_stage_head(C.byref(g["nets"][sid]), C.byref(g["tables"][sid]), io, B, g["ws"].data_ptr(), st), "ihmr_mlp_stage_head")
            vposed_ok = vposed_ok and not g["moves_vposed"][sid]
            if g["cam_only"][sid]:
                hip.check(L.ihmr_mlp_camera_select(io, B, C.byref(w), C.byref(g["tables"][sid]), C.byref(g["stages"][sid]), g["ws"].data_ptr(), st),
                          "ihmr_mlp_camera_select")
                continue
            hip.check(L.ihmr_mlp_forward_select(mr, ml, io, B, C.byref(w), C.byref(g["tables"][sid]), C.byref(g["stages"][sid]), 3 if vposed_ok else 2,
                                                g["ws"].data_ptr(), st), "ihmr_mlp_forward_select")
        final = g["final"]
        hip.check(L.ihmr_opt_set_params(io, final.data_ptr(), B, st), "ihmr_opt_set_params")
        hip.check(L.ihmr_mlp_forward_select(mr, ml, io, B, C.byref(w), None, None, 0, g["ws"].data_ptr(), st), "ihmr_mlp_forward_select")
        self.final_params = final
        self.collision_loss_batch = core.buf["loss_batch"][2]
        for n, sl in COLS.items():
            setattr(self, n, final[:, sl])
        c = core.buf
        self.pred_right_hand_verts, self.pred_left_hand_verts = c["verts"][0], c["verts"][1]
        self.pred_joints_3d, self.collision_loss_origin_scale = c["joints_3d"], c["coll_origin_scale"]
        # ground-truth meshes for the export (mlp_model.py:497-501): the annotation's parameters through the same skeleton + skinning kernels
        hip.check(L.ihmr_opt_set_params(C.byref(g["gt_io"]), self._gt_packed.data_ptr(), B, st), "ihmr_opt_set_params")
        hip.check(L.ihmr_opt_forward_verts(mr, C.byref(g["gt_io"]), B, st), "ihmr_opt_forward_verts")
        self.gt_right_hand_verts, self.gt_left_hand_verts = g["gt_buf"]["verts"][0], g["gt_buf"]["verts"][1]

    def _export_sources(self):
        # the reference root-aligns its GT joint buffer in place (no clone at mlp_model.py:530-531) and exports it: one launch
        if getattr(self, "_gt_aligned", None) is None or self._gt_aligned.shape != self.joints_3d.shape:
            self._gt_aligned = torch.empty_like(self.joints_3d)
        hip.check(hip.lib().ihmr_root_align_joints(self.joints_3d.data_ptr(), self._gt_aligned.data_ptr(), self.batch_size, hip.stream_ptr()),
                  "ihmr_root_align_joints")
        gt = self._gt_aligned
        return OrderedDict(
            pred_cam_params=self.pred_cam_params, pred_pose_params=self.pred_pose_params, pred_shape_params=self.pred_shape_params,
            pred_hand_trans=self.pred_hand_trans, gt_right_hand_verts=self.gt_right_hand_verts, gt_left_hand_verts=self.gt_left_hand_verts,
            pred_right_hand_verts=self.pred_right_hand_verts, pred_left_hand_verts=self.pred_left_hand_verts,
            mano_params_weight=self.mano_params_weight, pred_joints_3d=self.pred_joints_3d, gt_joints_3d=gt,
            collision_loss=self.collision_loss_batch, collision_loss_origin_scale=self.collision_loss_origin_scale)

    _KEY_ORDER = ("pred_cam_params", "pred_pose_params", "pred_shape_params", "pred_hand_trans", "gt_right_hand_verts", "gt_left_hand_verts",
                  "pred_right_hand_verts", "pred_left_hand_verts", "mano_params_weight", "pred_joints_3d", "gt_joints_3d", "do_flip",
                  "collision_loss", "collision_loss_origin_scale")

    # mlp_model.py:702-719
    def get_pred_result(self):
        """Arrays the caller OWNS: the reference's loop keeps every batch's rows until it pickles them at the end (test_mlp.py:61-68,
        evaluator.py:38-97 stores row views), so the synchronous export copies out of the two alternating pinned buffers the
        asynchronous one hands out views of."""
        return OrderedDict((k, np.array(v, copy=True)) for k, v in self.get_pred_result_async().wait().items())

    def _packed_export(self, src):
        """The export's 13 tensors gathered into ONE device buffer by one launch (``ihmr_copy_segments``) and sent to the host in ONE
        copy (round 4: 13 copies); returns the pinned host buffer and the (offset, shape, dtype) of every key in it."""
        if not hasattr(self, "_pinned"):
            self._pinned, self._pin_slot, self._pack_dev, self._pack_layout = [None, None], 0, None, None
        layout, off = OrderedDict(), 0
        for k, v in src.items():
            layout[k] = (off, tuple(v.shape), v.dtype)
            off += (v.numel() * v.element_size() + 15) // 16 * 16
        if self._pack_dev is None or self._pack_dev.numel() < off or self._pack_layout != layout:
            self._pack_dev = torch.empty(off, dtype=torch.uint8, device=self.device)
            self._pinned, self._pack_layout = [None, None], layout
        self._pin_slot ^= 1
        if self._pinned[self._pin_slot] is None:
            self._pinned[self._pin_slot] = torch.empty(off, dtype=torch.uint8, pin_memory=True)
        pairs = []
        for k, v in src.items():
            o, shape, dt = layout[k]
            dst = self._pack_dev[o:o + v.numel() * v.element_size()].view(dt).view(shape)
            v = v.detach()
            if not v.is_contiguous():
                if v.dim() == 2 and v.stride(1) == 1:
                    pass                                  # a column slice of the packed parameter matrix: a strided segment
                else:
                    v = v.contiguous()
            pairs.append((v, dst))
        hip.copy_segments(pairs)
        host = self._pinned[self._pin_slot]
        host.copy_(self._pack_dev[:host.numel()], non_blocking=True)
        return host, layout

    def get_pred_result_async(self):
        """``get_pred_result()`` without stalling the host (as :meth:`OptimizeModel.get_pred_result_async`): the copies are
        queued behind ``test()`` on the current stream into pinned buffers (two alternating sets); ``wait()`` on the
        returned handle blocks until they have landed.  The arrays are valid until the next-but-one export."""
        host, layout = self._packed_export(self._export_sources())
        ev = torch.cuda.Event()
        ev.record()
        model = self

        class _Pending:
            def wait(self_inner):
                ev.synchronize()
                out = {}
                for k, (o, shape, dt) in layout.items():
                    n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
                    out[k] = host[o:o + n].view(dt).view(shape).numpy()
                out["do_flip"] = np.zeros(model.batch_size).astype(np.int32)
                return OrderedDict((k, out[k]) for k in model._KEY_ORDER)
        return _Pending()
