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
