"""MANO model arrays: deterministic synthetic "MANO-shaped" assets + loader for real MANO_*.pkl.

The real MANO files (``MANO_LEFT.pkl`` / ``MANO_RIGHT.pkl``, read by the reference through
``smplx.create`` at ``src/models/optimize_model.py:103-106``) are licence-gated and absent from this
build, so every test and benchmark uses :func:`synthetic_mano`, which produces arrays with exactly the
MANO sizes and structure:

* 778 vertices, 1538 faces, one open 16-edge boundary loop at the wrist (Euler: F = 2V - 2 - b),
* 16 joints with the MANO kinematic tree ``[-1,0,1,2,0,4,5,0,7,8,0,10,11,0,13,14]``,
* ``J_regressor`` (16,778) and ``lbs_weights`` (778,16) with non-negative rows summing to 1,
* ``shapedirs`` (778,3,10), ``posedirs`` (135,2334) [row = (joint-1)*9 + 3*r + c, col = 3*v + k],
  ``hands_mean`` (45),
* the five fingertip vertex ids the reference hard-codes (``optimize_model.py:99``:
  ``[744, 320, 443, 554, 671]``) sit at the ends of the five finger chains.

If a user supplies the real files, :func:`load_mano_pkl` reads them through the same dict layout.
"""
from __future__ import annotations

import io
import os.path as osp
import pickle
from typing import Dict

import numpy as np

NUM_VERTS = 778
NUM_FACES = 1538
NUM_JOINTS = 16
MANO_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int32)
# reference optimize_model.py:99 / baseline_model.py:136 (thumb, index, middle, ring, pinky)
TIP_VERTEX_IDS = np.array([744, 320, 443, 554, 671], dtype=np.int32)


def _mitten_mesh(rng: np.random.RandomState):
    """pole + 48 rings x 16 verts (V=769, F=16+32*47=1520, b=16), then 9 centroid splits."""
    n_ring, n_seg = 48, 16
    verts = [np.array([0.185, 0.0, 0.0])]  # pole = finger-end of the mitten
    for r in range(n_ring):
        t = (r + 1) / n_ring  # 0 -> pole, 1 -> wrist
        x = 0.185 * (1.0 - t) ** 1.0
        # half-width / half-thickness profile (metres): rounded tip, palm bulge, narrower wrist
        prof = np.sin(np.clip(t * 1.25, 0, 1) * np.pi / 2) ** 0.6
        wy = 0.043 * prof * (1.0 - 0.25 * t ** 3)
        wz = 0.014 * prof * (1.0 + 0.3 * np.sin(np.pi * t))
        for s in range(n_seg):
            a = 2 * np.pi * (s + 0.5 * (r % 2)) / n_seg
            verts.append(np.array([x, wy * np.cos(a), wz * np.sin(a)]))
    verts = np.array(verts)
    faces = []
    for s in range(n_seg):  # pole fan
        faces.append([0, 1 + s, 1 + (s + 1) % n_seg])
    for r in range(n_ring - 1):
        a0 = 1 + r * n_seg
        b0 = 1 + (r + 1) * n_seg
        for s in range(n_seg):
            s1 = (s + 1) % n_seg
            faces.append([a0 + s, b0 + s, b0 + s1])
            faces.append([a0 + s, b0 + s1, a0 + s1])
    faces = np.array(faces, dtype=np.int64)
    assert verts.shape[0] == 769 and faces.shape[0] == 1520
    # 9 centroid splits of well-separated palm faces: +1 V, +2 F each
    split_ids = [200 + 97 * k for k in range(9)]
    vl, fl = list(verts), [list(f) for f in faces]
    for fid in split_ids:
        a, b, c = fl[fid]
        cen = (vl[a] + vl[b] + vl[c]) / 3.0
        cen = cen * np.array([1.0, 1.03, 1.03])  # slightly off-plane so no degenerate triangles
        n = len(vl)
        vl.append(cen)
        fl[fid] = [a, b, n]
        fl.append([b, c, n])
        fl.append([c, a, n])
    verts = np.array(vl)
    faces = np.array(fl, dtype=np.int64)
    assert verts.shape == (NUM_VERTS, 3) and faces.shape == (NUM_FACES, 3)
    # small smooth perturbation so that nothing is axis-degenerate
    verts = verts + 0.0004 * rng.standard_normal(verts.shape)
    return verts, faces


# ----------------------------------------------------------------------------- second synthetic asset: a hand WITH fingers
# The mitten above is a blob: a penetrating hand meets one convex-ish volume (~118 inside voxels per sample of the benchmark batch).
# Real MANO hands have five fingers that interlock, i.e. many thin penetration volumes, long candidate lists between neighbouring
# fingers, more voxels that are refused a list.  `synthetic_mano(kind="fingers")` is the same MANO-shaped model (778 vertices, 1538
# faces, one open 16-edge wrist loop, same tree / weights / blend-shape construction) on a mesh with a palm and five finger tubes, so
# that throughput and list statistics can be quoted for that geometry too (bench.py --asset fingers).
# finger slots along the knuckle line from +y to -y: thumb, index, middle, ring, pinky; MANO chain of each slot (joint order
# index, middle, pinky, ring, thumb: loss_utils.py:139-145)
_FINGER_SLOT_CHAIN = [4, 0, 1, 3, 2]
_FINGER_BASE = np.array([[0.058, 0.043, 0.0], [0.096, 0.027, 0.0], [0.100, 0.009, 0.0], [0.096, -0.009, 0.0], [0.088, -0.027, 0.0]])
_FINGER_DIR = np.array([[0.62, 0.78, 0.05], [1.0, 0.10, 0.0], [1.0, 0.02, 0.0], [1.0, -0.07, 0.0], [1.0, -0.16, 0.0]])
_FINGER_LEN = np.array([0.058, 0.074, 0.080, 0.074, 0.060])
_FINGER_RAD = np.array([0.0095, 0.0080, 0.0082, 0.0078, 0.0070])
_FINGER_RINGS = [8, 10, 10, 9, 8]            # rings beyond the base loop (+ a pole each)
_PALM_RINGS = [16, 17, 19, 21, 24, 28, 32, 36, 40, 44]   # wrist (the open boundary) ... last ring before the knuckle loop


def _loop_strip(A, B):
    """Triangles between two closed vertex loops of the same circulation (lengths may differ): len(A) + len(B) faces."""
    na, nb = len(A), len(B)
    faces, i, j = [], 0, 0
    while i < na or j < nb:
        if j >= nb or (i < na and (i + 1) * nb <= (j + 1) * na):
            faces.append([A[i % na], A[(i + 1) % na], B[j % nb]]); i += 1
        else:
            faces.append([A[i % na], B[(j + 1) % nb], B[j % nb]]); j += 1
    return faces


def _fingers_mesh(rng: np.random.RandomState):
    """Palm tube (wrist ring of 16 -> rings growing to 44) -> knuckle loop whose five lobes are the base loops of five finger tubes
    (neighbouring lobes meet in ONE web vertex) -> finger rings -> poles.  A triangulated disk with one boundary loop of 16 edges:
    F = 2 V - 2 - 16 whatever the ring sizes, which are chosen so that V = 778."""
    verts, faces = [], []
    add = lambda p: (verts.append(np.asarray(p, np.float64)), len(verts) - 1)[1]
    ang = lambda n, k: 2 * np.pi * k / n
    # palm rings (planar ellipses, circulation +y -> +z -> -y -> -z)
    x_k = 0.088
    rings = []
    for r, n in enumerate(_PALM_RINGS):
        u = r / len(_PALM_RINGS)
        x = x_k * u
        wy = 0.027 + 0.020 * np.sin(u * np.pi / 2) ** 0.8
        wz = 0.0125 + 0.0035 * np.sin(np.pi * u)
        yc = 0.004 * u          # the thumb side bulges
        rings.append([add([x, yc + wy * np.cos(ang(n, k)), wz * np.sin(ang(n, k))]) for k in range(n)])
    # fingers: orthonormal frame per finger (d = axis, e1 ~ +y side, e2 ~ +z dorsal)
    a = 4                        # vertices per half loop: a finger loop has 2 a + 2 = 10
    frames = []
    for i in range(5):
        d = _FINGER_DIR[i] / np.linalg.norm(_FINGER_DIR[i])
        e2 = np.array([0.0, 0.0, 1.0]); e2 = e2 - (e2 @ d) * d; e2 /= np.linalg.norm(e2)
        e1 = np.cross(e2, d)      # points to +y for d = +x
        frames.append((d, e1, e2))
    # web vertices: outer side of the thumb, between neighbours, outer side of the pinky (on the finger bases' mid-plane)
    web = []
    for i in range(6):
        if i == 0:
            p = _FINGER_BASE[0] + frames[0][1] * _FINGER_RAD[0]
        elif i == 5:
            p = _FINGER_BASE[4] - frames[4][1] * _FINGER_RAD[4]
        else:
            p = 0.5 * ((_FINGER_BASE[i - 1] - frames[i - 1][1] * _FINGER_RAD[i - 1]) + (_FINGER_BASE[i] + frames[i][1] * _FINGER_RAD[i]))
            p = p - np.array([0.004, 0.0, 0.0])     # the web sits a little proximal of the knuckles
        web.append(add(p))
    dors, palm_arcs = [], []
    for i in range(5):
        d, e1, e2 = frames[i]
        th = [np.pi * (q + 1) / (a + 1) for q in range(a)]           # from the +y side over the top to the -y side
        dors.append([add(_FINGER_BASE[i] + _FINGER_RAD[i] * (np.cos(t) * e1 + np.sin(t) * e2)) for t in th])
        # palmar arc in the knuckle loop's direction: from the -y side under the finger back to the +y side
        palm_arcs.append([add(_FINGER_BASE[i] + _FINGER_RAD[i] * (np.cos(np.pi + t) * e1 + np.sin(np.pi + t) * e2)) for t in th])
    K = []
    for i in range(5):
        K += [web[i]] + dors[i]
    K += [web[5]]
    for i in range(4, -1, -1):
        K += palm_arcs[i] + ([web[i]] if i > 0 else [])
    # palm strips
    for r in range(len(rings) - 1):
        faces += _loop_strip(rings[r], rings[r + 1])
    faces += _loop_strip(rings[-1], K)
    # finger tubes
    for i in range(5):
        d, e1, e2 = frames[i]
        n = 2 * a + 2
        prev = [web[i]] + dors[i] + [web[i + 1]] + palm_arcs[i]
        for q in range(_FINGER_RINGS[i]):
            t = (q + 1) / (_FINGER_RINGS[i] + 0.6)
            rad = _FINGER_RAD[i] * (1.0 - 0.30 * t ** 2)
            c = _FINGER_BASE[i] + d * (_FINGER_LEN[i] * t)
            cur = [add(c + rad * (np.cos(ang(n, k)) * e1 + np.sin(ang(n, k)) * e2)) for k in range(n)]
            faces += _loop_strip(prev, cur)
            prev = cur
        pole = add(_FINGER_BASE[i] + d * _FINGER_LEN[i])
        for k in range(n):
            faces.append([prev[k], prev[(k + 1) % n], pole])
    verts = np.array(verts)
    faces = np.array(faces, dtype=np.int64)
    assert verts.shape == (NUM_VERTS, 3) and faces.shape == (NUM_FACES, 3), (verts.shape, faces.shape)
    # orientation: outward (positive signed volume of the surface closed by the wrist disk is enough to tell)
    vol = np.einsum("ij,ij->i", verts[faces[:, 0]], np.cross(verts[faces[:, 1]], verts[faces[:, 2]])).sum()
    if vol < 0:
        faces = faces[:, ::-1].copy()
    verts = verts + 0.0003 * rng.standard_normal(verts.shape)
    return verts, faces


def _joint_layout_fingers():
    """16 rest joints inside the palm / finger tubes of `_fingers_mesh` + the five tips in the reference's order."""
    J = np.zeros((16, 3))
    J[0] = [0.012, 0.0, 0.0]
    tips = np.zeros((5, 3))
    for slot, chain in enumerate(_FINGER_SLOT_CHAIN):
        d = _FINGER_DIR[slot] / np.linalg.norm(_FINGER_DIR[slot])
        for k, frac in enumerate((0.02, 0.40, 0.70)):
            J[1 + 3 * chain + k] = _FINGER_BASE[slot] + d * (_FINGER_LEN[slot] * frac)
        tips[chain] = _FINGER_BASE[slot] + d * _FINGER_LEN[slot]
    return J, tips[[4, 0, 1, 3, 2]]          # reference tip order (optimize_model.py:99): thumb, index, middle, ring, pinky


def _joint_layout():
    """16 rest joints (metres) inside the mitten: wrist + 5 chains of 3 (MANO order:
    index, middle, pinky, ring, thumb -- see the finger table at reference loss_utils.py:139-145)."""
    J = np.zeros((16, 3))
    J[0] = [0.012, 0.0, 0.0]
    # lateral position (y) of each chain in MANO joint order: index, middle, little, ring, thumb
    ys = [0.022, 0.007, -0.026, -0.010, 0.034]
    x0 = [0.095, 0.100, 0.085, 0.095, 0.045]
    seg = [0.028, 0.030, 0.022, 0.027, 0.026]
    for c in range(5):
        for k in range(3):
            J[1 + 3 * c + k] = [x0[c] + seg[c] * k, ys[c] * (1 - 0.12 * k), 0.001 * (c - 2)]
    tips = np.array([[x0[c] + seg[c] * 3 - 0.004, ys[c] * 0.62, 0.0] for c in range(5)])
    # reference tip order (optimize_model.py:99): thumb, index, middle, ring, pinky
    tips_ref_order = tips[[4, 0, 1, 3, 2]]
    return J, tips_ref_order


def synthetic_mano(is_rhand: bool = True, seed: int = 0, kind: str = "mitten") -> Dict[str, np.ndarray]:
    """Deterministic MANO-shaped arrays. The left model is the x-mirror of the right one (same
    ``shapedirs`` x-sign as the right, so that the reference's sign fix at
    ``optimize_model.py:109-113`` triggers exactly as it does for the real files).
    ``kind``: "mitten" (the default asset of every test and benchmark: a pole-and-rings blob) or "fingers" (palm + five finger tubes:
    the geometry-sensitivity asset, same sizes and construction of everything but the mesh)."""
    rng = np.random.RandomState(seed)
    if kind == "fingers":
        verts, faces = _fingers_mesh(rng)
        J, tips = _joint_layout_fingers()
    else:
        assert kind == "mitten", kind
        verts, faces = _mitten_mesh(rng)
        J, tips = _joint_layout()

    # put the reference's hard-coded tip vertex ids at the ends of the finger chains
    perm = np.arange(NUM_VERTS)
    used = set()
    for tip_id, tip_pos in zip(TIP_VERTEX_IDS, tips):
        d = np.linalg.norm(verts[perm] - tip_pos[None], axis=1)
        for u in used:
            d[u] = np.inf
        src = int(np.argmin(d))
        perm[[tip_id, src]] = perm[[src, tip_id]]
        used.add(int(tip_id))
    inv = np.empty_like(perm)
    inv[perm] = np.arange(NUM_VERTS)
    verts = verts[perm]
    faces = inv[faces]

    # skinning weights: softmax of -dist^2 to the bone segments, top-4, renormalised
    def seg_dist(p, a, b):
        ab = b - a
        t = np.clip(((p - a) @ ab) / max(ab @ ab, 1e-12), 0, 1)
        return np.linalg.norm(p - (a[None] + t[:, None] * ab[None]), axis=1)

    child_of = {j: [c for c in range(16) if MANO_PARENTS[c] == j] for j in range(16)}
    dist = np.zeros((NUM_VERTS, 16))
    for j in range(16):
        if j == 0:
            a, b = J[0], J[0] + np.array([0.06, 0.0, 0.0])
        elif child_of[j]:
            a, b = J[j], J[child_of[j][0]]
        else:
            a, b = J[j], J[j] + (J[j] - J[MANO_PARENTS[j]])
        dist[:, j] = seg_dist(verts, a, b)
    logit = -(dist / 0.012) ** 2
    logit -= logit.max(axis=1, keepdims=True)
    w = np.exp(logit)
    kth = np.sort(w, axis=1)[:, -4][:, None]
    w = np.where(w >= kth, w, 0.0)
    w /= w.sum(axis=1, keepdims=True)

    # joint regressor: sparse, non-negative, rows sum to 1, reproduces J approximately
    Jreg = np.zeros((16, NUM_VERTS))
    for j in range(16):
        d = np.linalg.norm(verts - J[j][None], axis=1)
        idx = np.argsort(d)[:24]
        ww = np.exp(-(d[idx] / 0.02) ** 2) + 1e-3
        Jreg[j, idx] = ww / ww.sum()

    # shape blend shapes: smooth low-frequency displacement fields, a few mm per unit beta
    shapedirs = np.zeros((NUM_VERTS, 3, 10))
    for l in range(10):
        freq = rng.uniform(10, 40, size=(3, 3))
        phase = rng.uniform(0, 2 * np.pi, size=(3,))
        amp = 0.004 / (1 + 0.35 * l)
        for k in range(3):
            shapedirs[:, k, l] = amp * np.sin(verts @ freq[k] + phase[k])
        # global scale component on the first betas (like MANO's first PCs)
        if l < 2:
            shapedirs[:, :, l] += (0.05 if l == 0 else 0.02) * (verts - J[0][None]) * ([1, 1, 1] if l == 0 else [1, -0.5, -0.5])
    # pose blend shapes: small, localised around the driving joint
    posedirs = np.zeros((135, NUM_VERTS * 3))
    for j in range(1, 16):
        near = np.exp(-(np.linalg.norm(verts - J[j][None], axis=1) / 0.02) ** 2)  # (778,)
        for e in range(9):
            field = 0.0025 * rng.standard_normal((NUM_VERTS, 3)) * near[:, None]
            posedirs[(j - 1) * 9 + e] = field.reshape(-1)
    hands_mean = 0.12 * rng.standard_normal(45)
    hands_mean[36:45] *= 1.5  # thumb has the largest mean rotation in MANO too

    out = dict(
        v_template=verts, faces=faces, shapedirs=shapedirs, posedirs=posedirs,
        J_regressor=Jreg, lbs_weights=w, parents=MANO_PARENTS.copy(), hands_mean=hands_mean,
    )
    if not is_rhand:
        out = _mirror_to_left(out)
    return _as_f32(out)


def _mirror_to_left(m: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Left-hand arrays as the x-mirror of the right ones. ``shapedirs`` keeps the RIGHT x-sign on
    purpose: the reference detects that (``shape_diff < 1e-7``) and flips it in place
    (optimize_model.py:109-113, baseline_model.py:145-149, mlp_model.py:112-116)."""
    out = {k: np.array(v, copy=True) for k, v in m.items()}
    out["v_template"][:, 0] *= -1
    out["faces"] = out["faces"][:, ::-1].copy()  # keep outward orientation
    pd = out["posedirs"].reshape(15, 3, 3, NUM_VERTS, 3).copy()
    # mirror across x: R' = S R S (S = diag(-1,1,1)); displacement x-component flips
    sgn = np.array([-1.0, 1.0, 1.0])
    pd = pd * (sgn[None, :, None, None, None] * sgn[None, None, :, None, None] * sgn[None, None, None, None, :])
    out["posedirs"] = pd.reshape(135, NUM_VERTS * 3)
    hm = out["hands_mean"].reshape(15, 3).copy()
    hm[:, 1:] *= -1
    out["hands_mean"] = hm.reshape(45)
    return out


def _as_f32(m):
    out = {}
    for k, v in m.items():
        if k == "faces":
            out[k] = np.ascontiguousarray(v, dtype=np.int64)
        elif k == "parents":
            out[k] = np.ascontiguousarray(v, dtype=np.int32)
        else:
            out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class _Stub:
    """Stand-in for classes pickled from packages that are not installed (chumpy, scipy sparse...)."""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (_Stub,), {"__module__": module})


def _to_array(x):
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    for attr in ("x", "r", "_state"):
        if hasattr(x, attr):
            return np.asarray(getattr(x, attr))
    return np.asarray(x)


def load_mano_pkl(path: str) -> Dict[str, np.ndarray]:
    """Read a real ``MANO_{LEFT,RIGHT}.pkl`` (chumpy-pickled) into the same dict as
    :func:`synthetic_mano`, following what smplx 0.1.28 ``MANO.__init__`` extracts (``f``,
    ``v_template``, ``shapedirs``, ``posedirs`` reshaped to (135, V*3), ``J_regressor``,
    ``kintree_table[0]`` with parents[0] = -1, ``weights``, ``hands_mean``).
    The real files are licence-gated and absent here; the layout is exercised by tests/test_host_cpu.py with a file
    written in that layout (chumpy-pickled arrays, scipy-sparse J_regressor, uint32 faces).  numpy-only, no chumpy needed."""
    with open(path, "rb") as f:
        data = _TolerantUnpickler(io.BytesIO(f.read()), encoding="latin1").load()
    v_template = _to_array(data["v_template"]).astype(np.float64)
    shapedirs = _to_array(data["shapedirs"]).astype(np.float64)[:, :, :10]
    posedirs = _to_array(data["posedirs"]).astype(np.float64)
    posedirs = posedirs.reshape(-1, posedirs.shape[-1]).T  # (135, V*3)
    parents = _to_array(data["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    out = dict(
        v_template=v_template, faces=_to_array(data["f"]).astype(np.int64),
        shapedirs=shapedirs, posedirs=posedirs,
        J_regressor=_to_array(data["J_regressor"]).astype(np.float64),
        lbs_weights=_to_array(data["weights"]).astype(np.float64),
        parents=parents.astype(np.int32), hands_mean=_to_array(data["hands_mean"]).astype(np.float64).reshape(-1),
    )
    assert out["v_template"].shape == (NUM_VERTS, 3) and out["faces"].shape == (NUM_FACES, 3)
    return _as_f32(out)


def get_mano_arrays(model_path: str | None, is_rhand: bool, seed: int = 0) -> Dict[str, np.ndarray]:
    """Real file if it exists, else the synthetic asset (the only case exercised in this build).  A model root of
    ``synthetic:fingers`` (``opt.model_root``; the path then reads ``synthetic:fingers/MANO_RIGHT.pkl``) selects the finger asset."""
    if model_path and osp.isfile(model_path):
        return load_mano_pkl(model_path)
    kind = "fingers" if model_path and "synthetic:fingers" in model_path else "mitten"
    return synthetic_mano(is_rhand=is_rhand, seed=seed, kind=kind)
