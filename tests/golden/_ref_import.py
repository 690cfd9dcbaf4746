"""Import the reference (``/root/reference/src``) on CPU inside the BUILD CONTAINER ONLY.

Used exclusively by ``make_golden.py`` (fixture generation) and by the optional drop-in tests that
skip when ``/root/reference`` is absent (it never exists on the GPU box).  Nothing here is copied
from the reference: we only register empty stand-in modules for its absent third-party imports,
make ``.cuda()`` an identity, and inject this build's CPU restatements at the two third-party seams
(``smplx.create`` and ``sdf.SDFLoss``).
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True   # never write __pycache__ next to the (read-only) reference sources


def reference_available() -> bool:
    return os.path.isdir(REF_SRC)


_DONE = False


def import_reference(smplx_create=None, sdf_loss_cls=None):
    """Returns a namespace of reference modules, imported with CPU stubs."""
    global _DONE
    assert reference_available(), "reference sources are only present in the build container"
    if not _DONE:
        repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        if repo not in sys.path:
            sys.path.insert(0, repo)
        # the reference uses top-level package names `models`, `utils`, `strategies`, `data`, `options`
        sys.path.insert(0, REF_SRC)
        for name in ["cv2", "ry_utils", "smplx", "sdf", "torchgeometry", "visdom", "dominate", "dominate.tags",
                     "opendr", "opendr.camera", "opendr.renderer", "opendr.lighting", "torchvision",
                     "torchvision.transforms", "PIL", "PIL.Image", "PIL.ImageDraw", "PIL.ImageFont"]:
            if name not in sys.modules:
                try:
                    __import__(name)
                except Exception:
                    m = types.ModuleType(name)
                    m.__path__ = []
                    sys.modules[name] = m
        for attr in ["ProjectPoints", "ColoredRenderer", "LambertianPointLight"]:
            for mod in ["opendr.camera", "opendr.renderer", "opendr.lighting"]:
                setattr(sys.modules[mod], attr, object)
        ry = sys.modules["ry_utils"]
        if not hasattr(ry, "load_pkl"):
            import pickle

            ry.load_pkl = lambda p: pickle.load(open(p, "rb"))
            ry.save_pkl = lambda p, o: pickle.dump(o, open(p, "wb"))
            ry.build_dir = lambda p: os.makedirs(p, exist_ok=True)
        # CPU-ify
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.FloatTensor = torch.FloatTensor
        _DONE = True

    from oracle import mano_ref, sdf_ref

    sys.modules["smplx"].create = smplx_create or mano_ref.create
    sys.modules["sdf"].SDFLoss = sdf_loss_cls or sdf_ref.SDFLossRef
    sys.modules["sdf"].SDFLoss_Single = sdf_ref.SDFLoss_Single

    import importlib

    ns = types.SimpleNamespace()
    ns.transform_utils = importlib.import_module("models.transform_utils")
    ns.loss_utils = importlib.import_module("models.loss_utils")
    ns.opt_utils = importlib.import_module("utils.opt_utils")
    ns.strategies = importlib.import_module("strategies")
    ns.optimize_model = importlib.import_module("models.optimize_model")
    ns.networks = importlib.import_module("models.networks")
    ns.resnet = importlib.import_module("models.resnet")
    ns.metric_utils = importlib.import_module("utils.metric_utils")
    # models imported after the seams are set read the injected classes at import time
    # (``from sdf import SDFLoss`` binds the name inside loss_utils):
    ns.loss_utils.SDFLoss = sys.modules["sdf"].SDFLoss
    return ns


def make_opt(batch_size, strategy="opt_default", save_mid_freq=1, model_root="", is_train=False):
    """Namespace with the option fields the reference models read (options/base_options.py,
    opt_options.py defaults)."""
    return types.SimpleNamespace(
        isTrain=is_train, dist=False, process_rank=-1, batchSize=batch_size, inputSize=224, input_nc=3,
        checkpoints_dir="./checkpoints", model_root=model_root, num_joints=42, total_params_dim=122,
        cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3,
        mean_param_file="mean_mano_params.pkl", main_encoder="resnet50", strategy=strategy,
        save_mid_freq=save_mid_freq, optimizer="adam", sdf_robustifier=None, use_hand_rotation=False)
