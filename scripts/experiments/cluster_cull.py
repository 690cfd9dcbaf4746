#!/usr/bin/env python3
"""CPU experiment (numpy): the full search of sdf_dist_kernel walks all 25 blocks of 64 triangles for every voxel pair.  If the faces are
ordered so that a block is a spatial cluster (kd split of the template's triangle centroids into leaves of 64), how many blocks can hold
a triangle within the list bound |p - m| - R <= ub + 2 * slack of an inside voxel?  A block k has a bounding sphere (c_k = the circle
centre of its first triangle, rho_k = max |m_f - c_k| + R_f); ub0 = min_k |p - c_k| bounds ub from above, so a block with
|p - c_k| - rho_k > ub0 + widen holds no candidate.  Prints the mean share of blocks that survive, native order against kd order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "experiments"))
import numpy as np, torch
from cull_bounds import two_hand
from oracle.sdf_ref import hand_boxes, sdf_grid


def kd_order(cent, leaf=64):
    """permutation of the triangles: recursive split along the longest axis at a multiple of `leaf`"""
    def rec(idx):
        if len(idx) <= leaf:
            return [idx]
        c = cent[idx]
        ax = int(np.argmax(c.max(0) - c.min(0)))
        o = idx[np.argsort(c[:, ax], kind="stable")]
        nl = -(-len(idx) // leaf)
        cut = (nl // 2) * leaf
        return rec(o[:cut]) + rec(o[cut:])
    return np.concatenate(rec(np.arange(len(cent))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    widen = 0.08
    hv, right, left = two_hand(B, 1234)
    centre, scale = hand_boxes(hv, 0.2)
    vn = ((hv - centre) / scale).numpy().astype(np.float64)
    res = {"native": [], "kd": []}
    tri = {"native": [], "kd": []}
    for h, asset in ((0, right), (1, left)):
        faces = np.asarray(asset["faces"]).astype(np.int64)
        vt = np.asarray(asset["v_template"], np.float64).reshape(-1, 3)
        perm = kd_order(vt[faces].mean(1))
        phi = sdf_grid(torch.from_numpy(vn[:, h].astype(np.float32)).contiguous(), torch.from_numpy(faces.astype(np.int32)), 32).numpy()
        for b in range(B):
            q = (hv[b, 1 - h].numpy() - centre[b, h].numpy()) / scale[b, h].numpy()
            f0 = np.floor(((q + 1) * 32 - 1) / 2).astype(int)
            need = set()
            for d in range(8):
                ijk = f0 + np.array([d & 1, (d >> 1) & 1, d >> 2])
                ok = ((ijk >= 0) & (ijk < 32)).all(1)
                for i, j, k in ijk[ok]: need.add((k, j, i))
            vox = np.array([v for v in need if phi[b, v[0], v[1], v[2]] > 0])
            if len(vox) == 0: continue
            p = np.stack([(2 * vox[:, 2] + 1) / 32 - 1, (2 * vox[:, 1] + 1) / 32 - 1, (2 * vox[:, 0] + 1) / 32 - 1], 1)
            for name, order in (("native", np.arange(len(faces))), ("kd", perm)):
                f = faces[order]
                A, Bv, C = vn[b, h][f[:, 0]], vn[b, h][f[:, 1]], vn[b, h][f[:, 2]]
                m = (A + Bv + C) / 3
                R = np.sqrt(np.maximum.reduce([((X - m) ** 2).sum(1) for X in (A, Bv, C)]))
                nb = -(-len(f) // 64)
                ck = np.stack([m[64 * k] for k in range(nb)])
                rho = np.array([(np.linalg.norm(m[64 * k:64 * k + 64] - ck[k], axis=1) + R[64 * k:64 * k + 64]).max() for k in range(nb)])
                dk = np.linalg.norm(p[:, None] - ck[None], axis=2)                # (V, nb)
                ub0 = dk.min(1, keepdims=True)
                keep = dk - rho[None] <= ub0 + widen
                res[name].append(keep.mean(1))
                # what the list really needs (per triangle), for reference
                dm = np.linalg.norm(p[:, None] - m[None], axis=2)
                ub = dm.min(1, keepdims=True)
                tri[name].append(((dm - R[None]) <= ub + widen).mean(1))
    for name in res:
        r = np.concatenate(res[name]); t = np.concatenate(tri[name])
        print(f"{name:7s}: {len(r)} inside voxels; blocks that survive: mean {100 * r.mean():.1f} % (p90 {100 * np.percentile(r, 90):.0f} %) of 25; "
              f"triangles within the list bound {100 * t.mean():.1f} %")


if __name__ == "__main__":
    main()
