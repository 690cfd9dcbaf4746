"""Minimal stand-in for the reference's external helper package ``ry_utils`` (penincillin/Tools,
``docs/install.md:40-45``): only the helpers the hot-path callers use (pickle / directory / OBJ)."""
import os
import pickle

import numpy as np


def load_pkl(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def save_pkl(path, obj):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def build_dir(path):
    os.makedirs(path, exist_ok=True)


def save_mesh_to_obj(path, verts, faces):
    """OBJ with 1-based face indices; for two hands the caller passes
    ``faces = concat(right, left + 778)`` (reference ``utils/opt_utils.py:48-54``)."""
    verts, faces = np.asarray(verts), np.asarray(faces).astype(np.int64)
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for t in faces + 1:
            f.write(f"f {t[0]} {t[1]} {t[2]}\n")


def save_pred_obj(res_dir, pred_result, mano_models, iter_id, data_id, opt_iter, sample=0):
    """The reference's ``utils/opt_utils.py:45-54``: the two predicted hands of one sample as ONE mesh -- right-hand
    vertices first, left-hand face indices shifted by the 778 right-hand vertices (``concat(faces_r, faces_l + 778)``,
    integer work, bit-exact) -- written to ``iter_XXXX_stage_XX_opt_iter_XXXX.obj``.  Returns the path."""
    right_verts = np.asarray(pred_result["pred_right_hand_verts"][sample])
    left_verts = np.asarray(pred_result["pred_left_hand_verts"][sample])
    right_faces = np.asarray(mano_models["right"].faces).astype(np.int64)
    left_faces = np.asarray(mano_models["left"].faces).astype(np.int64)
    verts = np.concatenate([right_verts, left_verts], axis=0)
    faces = np.concatenate([right_faces, left_faces + right_verts.shape[0]], axis=0)
    path = os.path.join(res_dir, f"iter_{iter_id:04d}_stage_{data_id:02d}_opt_iter_{opt_iter:04d}.obj")
    save_mesh_to_obj(path, verts, faces)
    return path
