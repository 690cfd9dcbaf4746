"""ctypes binding of ``libihmr_hip.so`` (C ABI declared in ``include/ihmr_hip.h``).

The library is the product: every compute entry point of this package goes through it and there is
NO CPU fallback -- if the shared object is missing (and cannot be built with hipcc) or no GPU is
visible, the calls raise.  PyTorch is used only for device memory and streams: tensors are passed as
raw device pointers, the launch stream is ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
import os
import os.path as osp
import subprocess

import numpy as np
import torch

_HERE = osp.dirname(osp.abspath(__file__))
LIB_PATH = osp.join(_HERE, "libihmr_hip.so")
SRC_DIR = osp.join(_HERE, "csrc")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"]

NUM_VERTS, NUM_FACES, NUM_JOINTS, OPT_NPARAM = 778, 1538, 16, 122
# include/ihmr_hip.h: parameter blocks (slot order of the 122-vector), filter / select losses, optimizers
PARAM_BLOCKS = dict(pred_cam_params=(1, 0, 3), pred_hand_trans=(2, 3, 3), pred_right_orient=(4, 6, 3), pred_left_orient=(8, 9, 3),
                    pred_right_pose_params=(16, 12, 45), pred_left_pose_params=(32, 57, 45),
                    pred_right_shape_params=(64, 102, 10), pred_left_shape_params=(128, 112, 10))   # name -> (bit, first slot, size)
LOSS_IDS = dict(joints_2d_loss_p=0, joints_3d_loss_p=1, collision_loss=2)
OPTIMIZERS = dict(adam=0, sgd=1)

# every symbol include/ihmr_hip.h declares (checked by the CPU test-suite against the built library)
EXPORTED_SYMBOLS = [
    "ihmr_mano_create", "ihmr_mano_destroy", "ihmr_mano_update_shapedirs", "ihmr_mano_workspace_bytes", "ihmr_mano_lbs_fwd",
    "ihmr_mano_lbs_bwd",
    "ihmr_sdf_workspace_bytes", "ihmr_sdf_collision", "ihmr_sdf_collision_ex", "ihmr_sdf_dense_grid", "ihmr_opt_workspace_bytes",
    "ihmr_opt_run_stage", "ihmr_opt_forward_losses", "ihmr_opt_sdf_stats", "ihmr_opt_sdf_counters", "ihmr_opt_sdf_inside_bits", "ihmr_opt_stage_graph_create",
    "ihmr_opt_forward_graph_create", "ihmr_graph_launch", "ihmr_graph_destroy", "ihmr_opt_set_params", "ihmr_eval_metrics", "ihmr_eval_mpvpe", "ihmr_conv_igemm", "ihmr_maxpool3x3s2",
    "ihmr_avgpool_relu", "ihmr_preprocess_images", "ihmr_mlp_train_grad", "ihmr_transpose", "ihmr_relu_backward", "ihmr_colsum",
    "ihmr_adam_step", "ihmr_bn_workspace_bytes", "ihmr_bn_train_forward", "ihmr_bn_train_backward", "ihmr_conv_wgrad",
    "ihmr_dilate2", "ihmr_interleave2", "ihmr_pack_dgrad_weight", "ihmr_maxpool3x3s2_backward", "ihmr_avgpool_relu_backward", "ihmr_set_kernel_timer", "ihmr_flush_kernel_timer",
    "ihmr_mlp_workspace_bytes", "ihmr_mlp_stage_head", "ihmr_mlp_forward_select", "ihmr_mlp_camera_select", "ihmr_opt_forward_verts",
    "ihmr_debug_force_lbs_bwd2_streaming", "ihmr_debug_force_full_skin", "ihmr_version", "ihmr_copy_segments", "ihmr_root_align_joints",
]


class ManoArrays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents",
                                          "hands_mean", "faces", "tip_ids")]


class OptIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "cam", "trans", "orient", "pose", "shape",
        "init_joints_2d", "init_joints_3d", "init_hand_trans_j", "gt_joints_2d", "gt_joints_3d", "gt_hand_trans",
        "hand_type_array",
        "verts", "joints_3d", "joints_2d", "loss_batch", "coll_per_vert", "coll_origin_scale",
        "snap_params", "snap_loss", "selected", "adam_m", "adam_v", "workspace")] + [
        ("norm_batch", C.c_int), ("sdf_align_corners", C.c_int), ("sdf_loss_divisor", C.c_float), ("sdf_swap_xz", C.c_int),
        ("sdf_no_candidate_lists", C.c_int), ("sdf_no_static_reuse", C.c_int),
        ("no_fused_tail", C.c_int)]


class CopySeg(C.Structure):
    """``ihmr_copy_seg``: one 2-D strided copy of 32-bit words (rows, width, leading dimensions in dwords)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("width", C.c_int), ("src_ld", C.c_int), ("dst_ld", C.c_int)]


COPY_MAX_SEGS = 32


def copy_segments(pairs):
    """``pairs``: (src, dst) device tensors of equal shape and 4- or 8-byte element type -- dst may be a strided 2-D view (a column
    slice of a packed matrix), src too; everything else contiguous.  One launch (``ihmr_copy_segments``) per 32 pairs."""
    segs = []
    for src, dst in pairs:
        assert src.shape == dst.shape and src.dtype == dst.dtype and src.is_cuda and dst.is_cuda, (src.shape, dst.shape, src.dtype, dst.dtype)
        k = src.element_size() // 4
        assert k in (1, 2)

        def geom(t):
            if t.is_contiguous():
                return 1, t.numel() * k, t.numel() * k
            assert t.dim() == 2 and t.stride(1) == 1, "a strided operand must be a column slice of a row-major matrix"
            return t.shape[0], t.shape[1] * k, t.stride(0) * k
        (rs, ws, ls), (rd, wd, ld) = geom(src), geom(dst)
        if (rs, ws) != (rd, wd):                 # one side flat, the other a 2-D slice: describe both as the slice's rows
            rows, width = (rd, wd) if rd > 1 else (rs, ws)
            ls = ls if rs > 1 else width
            ld = ld if rd > 1 else width
        else:
            rows, width = rs, ws
        segs.append(CopySeg(src.data_ptr(), dst.data_ptr(), rows, width, ls, ld))
    L, st = lib(), stream_ptr()
    for i in range(0, len(segs), COPY_MAX_SEGS):
        chunk = segs[i:i + COPY_MAX_SEGS]
        arr = (CopySeg * len(chunk))(*chunk)
        check(L.ihmr_copy_segments(arr, len(chunk), st), "ihmr_copy_segments")


class SdfOptions(C.Structure):
    _fields_ = [("align_corners", C.c_int), ("loss_divisor", C.c_float), ("swap_xz", C.c_int)]


class OptStage(C.Structure):
    _fields_ = [("param_mask", C.c_int), ("optimizer", C.c_int), ("lr", C.c_float), ("n_iters", C.c_int), ("save_freq", C.c_int),
                ("use_filter", C.c_int * 3), ("filter_factor", C.c_float * 3), ("select_loss", C.c_int), ("keep_lists", C.c_int)]


class OptWeights(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("joints_2d", "joints_3d", "trans", "shape_reg", "collision", "finger_reg")]


class MlpNet(C.Structure):
    """``ihmr_mlp_net``: one IHMR-MLP sub-network -- packed weights / biases of its four Linear layers, output -> column map."""
    _fields_ = [("w", C.c_void_p * 4), ("b", C.c_void_p * 4), ("ldw", C.c_int * 4), ("k_out", C.c_int), ("col", C.c_int * 122)]


class MlpTables(C.Structure):
    """``ihmr_mlp_tables``: the "prev" tables of MLPModel (mlp_model.py:297-356) + the batch's working rows."""
    _fields_ = [(n, C.c_void_p) for n in ("idx", "data_idxs_all", "img_feat_all", "prev_final", "prev_loss", "img_feat", "new_params",
                                          "final_params", "kept")]


class MlpStage(C.Structure):
    """``ihmr_mlp_stage``: filter / select criteria of one stage (select_better_params, mlp_model.py:592-637)."""
    _fields_ = [("n_filter", C.c_int), ("filter_loss", C.c_int * 4), ("filter_factor", C.c_float * 4), ("select_loss", C.c_int)]


class TrainWeights(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("joints_2d", "mano_pose", "mano_shape", "hand_trans", "shape_reg", "shape_residual")]


TIMED_SDF_PREP, TIMED_SDF_DIST, TIMED_OPT_TAIL, TIMED_KERNELS = 0, 1, 2, 4


class KernelTimer(C.Structure):
    """``ihmr_kernel_timer`` (include/ihmr_hip.h): summed HIP-event time and launch count per timed kernel + the empty event pairs."""
    _fields_ = [("ms", C.c_double * TIMED_KERNELS), ("n", C.c_long * TIMED_KERNELS), ("ms_event_pair", C.c_double), ("n_event_pair", C.c_long)]

    def launch_ms(self, k):
        """Mean duration of kernel slot k's launches (event time minus the cost of an empty event pair)."""
        if self.n[k] <= 0:
            return None
        return self.ms[k] / self.n[k] - (self.ms_event_pair / self.n_event_pair if self.n_event_pair > 0 else 0.0)


def sources():
    return sorted(osp.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith((".hip", ".h"))) + [
        osp.join(osp.dirname(_HERE), "include", "ihmr_hip.h")]


HASH_PATH = osp.join(_HERE, "libihmr_hip.srchash")


def loaded_source_hash() -> str | None:
    """Source hash of the library this process would load, from its build record (None for an IHMR_HIP_LIBRARY override or a
    missing record).  bench.py quotes a committed profile only when the profile's recorded hash equals this."""
    if os.environ.get("IHMR_HIP_LIBRARY") or not osp.isfile(HASH_PATH):
        return None
    with open(HASH_PATH) as fh:
        rec = fh.read().split()
    return rec[0] if len(rec) == 2 else None


def _source_hash() -> str:
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in sources():
        h.update(osp.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _file_hash(path: str) -> str:
    import hashlib
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def is_stale() -> bool:
    """The library is current iff the record written at build time matches BOTH the hash of the sources (+ flags) and the hash
    of the shared object itself -- content, not mtimes: the tree is copied to the GPU box, several ranks may import at once,
    and a library written by anything but :func:`build` (an experiment's hipcc line) must not pass for the product."""
    if not osp.isfile(LIB_PATH) or not osp.isfile(HASH_PATH):
        return True
    with open(HASH_PATH) as fh:
        rec = fh.read().split()
    return len(rec) != 2 or rec[0] != _source_hash() or rec[1] != _file_hash(LIB_PATH)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles for gfx950 without a GPU (seconds).  Safe against concurrent callers (one rank per GPU): built
    under a lock file into a temporary name, then renamed into place."""
    if not force and not is_stale():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or is_stale():             # another process may have built it while this one waited
                tmp = f"{LIB_PATH}.tmp{os.getpid()}"
                cmd = ["hipcc"] + HIPCC_FLAGS + [osp.join(SRC_DIR, "ihmr_hip.hip"), "-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, LIB_PATH)
                with open(HASH_PATH + ".tmp", "w") as fh:
                    fh.write(_source_hash() + " " + _file_hash(LIB_PATH))
                os.replace(HASH_PATH + ".tmp", HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_LIB = None


def _resolve_library() -> str:
    """The product library.  `IHMR_HIP_LIBRARY` selects another build (profiling / A-B scripts use it instead of
    overwriting the product .so); otherwise the in-tree library, rebuilt when it is missing or older than its sources
    (hipcc present) -- a stale .so is never loaded silently."""
    override = os.environ.get("IHMR_HIP_LIBRARY")
    if override:
        if not osp.isfile(override):
            raise FileNotFoundError(f"IHMR_HIP_LIBRARY={override} does not exist")
        return override
    import shutil
    if shutil.which("hipcc"):
        return build()
    if not osp.isfile(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing and hipcc is not available to build it")
    if is_stale():
        raise RuntimeError(f"{LIB_PATH} does not match its sources (or its build record {HASH_PATH} is missing) and hipcc is not "
                           "available to rebuild it: the ctypes structs of this package would not match the binary.  Rebuild where "
                           "hipcc exists, or name a library explicitly with IHMR_HIP_LIBRARY")
    return LIB_PATH


def lib():
    """Load (building on demand when hipcc is present).  Raises if unavailable -- never falls back."""
    global _LIB
    if _LIB is None:
        L = C.CDLL(_resolve_library())
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.ihmr_mano_create.argtypes = [C.POINTER(ManoArrays), C.POINTER(vp)]
        L.ihmr_mano_destroy.argtypes = [vp]
        L.ihmr_mano_update_shapedirs.argtypes = [vp, vp]
        L.ihmr_mano_workspace_bytes.argtypes = [i]
        L.ihmr_mano_workspace_bytes.restype = C.c_size_t
        L.ihmr_mano_lbs_fwd.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp]
        L.ihmr_mano_lbs_bwd.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, i, vp]
        L.ihmr_sdf_workspace_bytes.argtypes = [i]
        L.ihmr_sdf_workspace_bytes.restype = C.c_size_t
        L.ihmr_sdf_collision.argtypes = [vp, vp, vp, i, f, vp, vp, vp, vp, vp, vp]
        L.ihmr_sdf_collision_ex.argtypes = [vp, vp, vp, i, f, C.POINTER(SdfOptions), vp, vp, vp, vp, vp, vp]
        L.ihmr_sdf_dense_grid.argtypes = [vp, vp, vp, i, vp, vp, vp]
        L.ihmr_opt_workspace_bytes.argtypes = [i]
        L.ihmr_opt_workspace_bytes.restype = C.c_size_t
        L.ihmr_opt_run_stage.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(OptStage), vp]
        L.ihmr_opt_forward_losses.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), vp]
        L.ihmr_opt_stage_graph_create.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(OptStage), C.POINTER(vp)]
        L.ihmr_opt_forward_graph_create.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(vp)]
        L.ihmr_graph_launch.argtypes = [vp, vp]
        L.ihmr_opt_set_params.argtypes = [C.POINTER(OptIO), vp, i, vp]
        L.ihmr_mlp_workspace_bytes.argtypes = [i]
        L.ihmr_mlp_workspace_bytes.restype = C.c_size_t
        L.ihmr_mlp_stage_head.argtypes = [C.POINTER(MlpNet), C.POINTER(MlpTables), C.POINTER(OptIO), i, vp, vp]
        L.ihmr_mlp_forward_select.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(MlpTables), C.POINTER(MlpStage), i, vp, vp]
        L.ihmr_mlp_camera_select.argtypes = [C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(MlpTables), C.POINTER(MlpStage), vp, vp]
        L.ihmr_opt_forward_verts.argtypes = [vp, C.POINTER(OptIO), i, vp]
        L.ihmr_eval_metrics.argtypes = [vp, vp, vp, vp, vp, i, vp, vp]
        L.ihmr_eval_mpvpe.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, vp]
        L.ihmr_graph_destroy.argtypes = [vp]
        L.ihmr_conv_igemm.argtypes = [vp, vp, vp, vp, vp] + [i] * 16 + [vp, C.c_size_t, vp]
        L.ihmr_maxpool3x3s2.argtypes = [vp, vp, i, i, i, i, i, i, vp]
        L.ihmr_avgpool_relu.argtypes = [vp, vp, i, i, i, i, vp]
        L.ihmr_mlp_train_grad.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), C.POINTER(TrainWeights)] + [vp] * 8 + [i, vp, i, vp]
        L.ihmr_transpose.argtypes = [vp, vp, i, i, i, i, vp]
        L.ihmr_relu_backward.argtypes = [vp, vp, i, i, i, i, vp]
        L.ihmr_colsum.argtypes = [vp, vp, i, i, i, vp]
        L.ihmr_adam_step.argtypes = [vp, vp, vp, vp, C.c_size_t, f, f, f, f, f, i, vp]
        L.ihmr_bn_workspace_bytes.argtypes = [i]
        L.ihmr_bn_workspace_bytes.restype = C.c_size_t
        L.ihmr_bn_train_forward.argtypes = [vp, C.c_long, i, vp, vp, vp, i, f, vp, vp, vp, vp, vp, vp, f, vp, vp]
        L.ihmr_bn_train_backward.argtypes = [vp, vp, C.c_long, i, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.ihmr_conv_wgrad.argtypes = [vp, vp, vp] + [i] * 14 + [vp, C.c_size_t, vp]
        L.ihmr_dilate2.argtypes = [vp, vp, i, i, i, i, vp]
        L.ihmr_interleave2.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
        L.ihmr_pack_dgrad_weight.argtypes = [vp, vp, i, i, i, i, i, i, vp]
        L.ihmr_maxpool3x3s2_backward.argtypes = [vp, vp, vp, i, i, i, i, i, i, vp]
        L.ihmr_avgpool_relu_backward.argtypes = [vp, vp, vp, i, i, i, i, vp]
        L.ihmr_preprocess_images.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp, vp, vp]
        L.ihmr_opt_sdf_stats.argtypes = [vp, vp, C.POINTER(OptIO), i, C.POINTER(OptWeights), vp, vp]
        L.ihmr_opt_sdf_counters.argtypes = [C.POINTER(OptIO), i, vp, i]
        L.ihmr_opt_sdf_inside_bits.argtypes = [C.POINTER(OptIO), i, vp, vp]
        L.ihmr_set_kernel_timer.argtypes = [C.POINTER(KernelTimer)]
        L.ihmr_flush_kernel_timer.argtypes = []
        L.ihmr_copy_segments.argtypes = [C.POINTER(CopySeg), i, vp]
        L.ihmr_root_align_joints.argtypes = [vp, vp, i, vp]
        L.ihmr_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libihmr_hip: {what} failed with code {rc}")


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("ihmr_amd needs an MI355X (gfx950) GPU: the hot path has no CPU fallback")


def ptr(t: torch.Tensor | None):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ManoHandle:
    """Device-resident MANO constants (``ihmr_mano_create``)."""

    def __init__(self, arrays: dict, tip_ids=(744, 320, 443, 554, 671)):
        require_gpu()
        self._keep = dict(
            v_template=np.ascontiguousarray(arrays["v_template"], np.float32),
            shapedirs=np.ascontiguousarray(arrays["shapedirs"], np.float32),
            posedirs=np.ascontiguousarray(arrays["posedirs"], np.float32),
            J_regressor=np.ascontiguousarray(arrays["J_regressor"], np.float32),
            lbs_weights=np.ascontiguousarray(arrays["lbs_weights"], np.float32),
            parents=np.ascontiguousarray(arrays["parents"], np.int32),
            hands_mean=np.ascontiguousarray(arrays["hands_mean"], np.float32),
            faces=np.ascontiguousarray(arrays["faces"], np.int32),
            tip_ids=np.ascontiguousarray(tip_ids, np.int32),
        )
        a = ManoArrays(**{k: v.ctypes.data for k, v in self._keep.items()})
        h = C.c_void_p()
        check(lib().ihmr_mano_create(C.byref(a), C.byref(h)), "ihmr_mano_create")
        self.handle = h

    def update_shapedirs(self, shapedirs: np.ndarray):
        sd = np.ascontiguousarray(shapedirs, np.float32)
        check(lib().ihmr_mano_update_shapedirs(self.handle, sd.ctypes.data), "ihmr_mano_update_shapedirs")

    def __del__(self):
        try:
            import sys
            if sys is None or sys.is_finalizing():
                return  # the HIP runtime may already be gone at interpreter shutdown
            if getattr(self, "handle", None):
                lib().ihmr_mano_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
