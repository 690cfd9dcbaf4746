"""ihmr_amd -- MI355X-native (gfx950) implementation of the IHMR hot path.

Host side mirrors the reference's Python interface for this path (``smplx.create``-style MANO layer in
:mod:`ihmr_amd.mano`, ``sdf.SDFLoss``-style collision module in :mod:`ihmr_amd.sdf`, model classes with
``set_input / optimize / test / get_pred_result``); all arithmetic lives in hand-written HIP kernels
behind the C ABI of ``include/ihmr_hip.h`` (:mod:`ihmr_amd.hip`).  There is no CPU fallback.
"""
__version__ = "0.1.0"
