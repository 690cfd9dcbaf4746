#!/usr/bin/env python3
"""CPU experiment (numpy): how many triangles survive different conservative lower bounds for the inside voxels of the
benchmark geometry?  (a) bounding sphere about the centroid (the shipped test), (b) plane distance + in-plane circle about the
centroid, (c) the same from fp16-rounded records with the error margins of DESIGN.md.  Upper bound = the exact minimum distance
(what the list search has: the last nearest triangle is almost always the answer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ihmr_amd.assets import synthetic_mano
from oracle.mano_ref import ManoRef
from oracle.sdf_ref import hand_boxes, sdf_grid

def two_hand(B, seed):
    from ihmr_amd.synthetic import synthetic_opt_batch
    right, left = synthetic_mano(True), synthetic_mano(False)
    mr, ml = ManoRef(right), ManoRef(left)
    def fwd_all(pose, shape, trans):
        o = {}
        for name, mm, ps, bs in (("right", mr, 0, 0), ("left", ml, 48, 10)):
            r = mm(global_orient=pose[:, ps:ps + 3], hand_pose=pose[:, ps + 3:ps + 48], betas=shape[:, bs:bs + 10])
            o[name] = (r.vertices, r.joints)
        shift = trans.reshape(-1, 1, 3) + (o["right"][1][:, 0:1] - o["left"][1][:, 0:1])
        return o["right"][0], o["left"][0] + shift, torch.cat([o["right"][1], o["left"][1] + shift], 1)
    b = synthetic_opt_batch(B, lambda p, s, t: torch.zeros(p.shape[0], 42, 3), seed=seed)
    rv, lv, _ = fwd_all(b["init_pose_params"], b["init_shape_params"], b["init_hand_trans"][:, 0, :3])
    return torch.stack([rv, lv], 1), right, left

def point_tri_d(p, a, b, c):
    # vectorised closest-point distance (float64): p (V,3), tri (F,3) -> (V,F)
    p = p[:, None, :]; a = a[None]; b = b[None]; c = c[None]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b; d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c; d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty(d1.shape); done = np.zeros(d1.shape, bool)
    def put(mask, q):
        m = mask & ~done
        out[m] = np.linalg.norm((p - q)[m], axis=-1) if m.any() else 0; done[m] = True
    put((d1 <= 0) & (d2 <= 0), a + 0 * p)
    put((d3 >= 0) & (d4 <= d3), b + 0 * p)
    with np.errstate(all="ignore"):
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[..., None] * ab)
        put((d6 >= 0) & (d5 <= d6), c + 0 * p)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[..., None] * ac)
        put((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[..., None] * (c - b))
        den = 1.0 / (va + vb + vc)
        put(np.ones(d1.shape, bool), a + ab * (vb * den)[..., None] + ac * (vc * den)[..., None])
    return out

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    hv, right, left = two_hand(B, 1234)
    centre, scale = hand_boxes(hv, 0.2)
    vn = ((hv - centre) / scale).numpy().astype(np.float64)
    tot = dict(vox=0, sph=0, pc=0, pc16=0, sph16=0, mec=0)
    for h, faces in ((0, right["faces"]), (1, left["faces"])):
        faces = np.asarray(faces).astype(np.int64)
        phi = sdf_grid(torch.from_numpy(vn[:, h].astype(np.float32)).contiguous(), torch.from_numpy(faces.astype(np.int32)), 32).numpy()
        for b in range(B):
            # voxels the other hand's vertices read (8 corners), inside only
            q = (hv[b, 1 - h].numpy() - centre[b, h].numpy()) / scale[b, h].numpy()
            ix = ((q + 1) * 32 - 1) / 2
            f0 = np.floor(ix).astype(int)
            need = set()
            for d in range(8):
                ijk = f0 + np.array([d & 1, (d >> 1) & 1, d >> 2])
                ok = ((ijk >= 0) & (ijk < 32)).all(1)
                for i, j, k in ijk[ok]: need.add((k, j, i))
            vox = np.array([v for v in need if phi[b, v[0], v[1], v[2]] > 0])
            if len(vox) == 0: continue
            p = np.stack([(2 * vox[:, 2] + 1) / 32 - 1, (2 * vox[:, 1] + 1) / 32 - 1, (2 * vox[:, 0] + 1) / 32 - 1], 1)
            A, Bv, C = vn[b, h][faces[:, 0]], vn[b, h][faces[:, 1]], vn[b, h][faces[:, 2]]
            d = point_tri_d(p, A, Bv, C)
            dmin = d.min(1, keepdims=True)
            ub = dmin * 1.0001 + 1e-6
            cen = (A + Bv + C) / 3
            R = np.sqrt(np.maximum.reduce([((X - cen) ** 2).sum(1) for X in (A, Bv, C)])) * 1.0001 + 1e-6
            n = np.cross(Bv - A, C - A); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
            pc = p[:, None, :] - cen[None]
            dc = np.linalg.norm(pc, axis=-1)
            lb_s = dc - R[None]
            hh = np.abs((pc * n[None]).sum(-1))
            rho = np.sqrt(np.maximum(dc ** 2 - hh ** 2, 0))
            lb_pc = np.sqrt(hh ** 2 + np.maximum(rho - R[None], 0) ** 2)
            assert (lb_pc <= d + 1e-9).all() and (lb_s <= d + 1e-9).all()
            # fp16 records: centroid, R rounded up, normal; margins
            c16 = cen.astype(np.float16).astype(np.float64); n16 = n.astype(np.float16).astype(np.float64)
            ec = np.linalg.norm(c16 - cen, axis=1)
            R16 = (R + 9e-4).astype(np.float16).astype(np.float64); R16 = np.where(R16 < R + 9e-4, np.nextafter(R16.astype(np.float16), np.float16(np.inf)).astype(np.float64), R16)
            pc2 = p[:, None, :] - c16[None]; dc2 = np.linalg.norm(pc2, axis=-1)
            h2 = np.abs((pc2 * n16[None]).sum(-1)); eh = 1.7e-3 * dc2 + 9e-4
            hlo = np.maximum(h2 - eh, 0); hhi = h2 + eh
            rlo = np.maximum(np.sqrt(np.maximum(dc2 ** 2 - hhi ** 2, 0)) - R16[None], 0)
            lb16 = np.sqrt(hlo ** 2 + rlo ** 2)
            assert (lb16 <= d + 1e-9).all(), float((lb16 - d).max())
            lb_s16 = dc2 - R16[None]
            # minimum enclosing circle instead of the centroid circle: circumcircle of an acute triangle, else the longest edge's
            la, lb_, lc = ((Bv - C) ** 2).sum(1), ((A - C) ** 2).sum(1), ((A - Bv) ** 2).sum(1)      # squared edge lengths opposite A, B, C
            obt = np.stack([la >= lb_ + lc, lb_ >= la + lc, lc >= la + lb_], 1)
            mid = np.stack([(Bv + C) / 2, (A + C) / 2, (A + Bv) / 2], 1)
            wa, wb, wc = la * (lb_ + lc - la), lb_ * (la + lc - lb_), lc * (la + lb_ - lc)
            circ = (wa[:, None] * A + wb[:, None] * Bv + wc[:, None] * C) / np.maximum(wa + wb + wc, 1e-30)[:, None]
            m = np.where(obt.any(1)[:, None], mid[np.arange(len(A)), obt.argmax(1)], circ)
            Rm = np.sqrt(np.maximum.reduce([((X - m) ** 2).sum(1) for X in (A, Bv, C)])) * 1.0001 + 1e-6
            pm = p[:, None, :] - m[None]; dm = np.linalg.norm(pm, axis=-1)
            hm = np.abs((pm * n[None]).sum(-1))
            lb_m = np.sqrt(hm ** 2 + np.maximum(np.sqrt(np.maximum(dm ** 2 - hm ** 2, 0)) - Rm[None], 0) ** 2)
            assert (lb_m <= d + 1e-9).all()
            tot["mec"] += int((lb_m <= ub).sum()); tot.setdefault("mec_s", 0); tot["mec_s"] += int((dm - Rm[None] <= ub).sum())
            tot.setdefault("Rratio", []).append(float((Rm / R).mean()))
            # list sizes at build time: bound = nearest-centre distance + 2 * slack
            for S in (0.04, 0.02):
                ubc = dc.min(1, keepdims=True) * 1.0001 + 1e-6 + 2 * S
                ubm = dm.min(1, keepdims=True) * 1.0001 + 1e-6 + 2 * S
                ube = ub + 2 * S
                tot.setdefault(f"L_sph_{S}", 0); tot[f"L_sph_{S}"] += int((lb_s <= ubc).sum())
                tot.setdefault(f"L_mec_{S}", 0); tot[f"L_mec_{S}"] += int((dm - Rm[None] <= ubm).sum())
                tot.setdefault(f"L_mecpc_{S}", 0); tot[f"L_mecpc_{S}"] += int((lb_m <= ubm).sum())
                tot.setdefault(f"L_mecpc_exact_{S}", 0); tot[f"L_mecpc_exact_{S}"] += int((lb_m <= ube).sum())
            tot["vox"] += len(vox); tot["sph"] += int((lb_s <= ub).sum()); tot["pc"] += int((lb_pc <= ub).sum())
            tot["pc16"] += int((lb16 <= ub).sum()); tot["sph16"] += int((lb_s16 <= ub).sum())
    v = tot["vox"]
    print({k: round(x / v, 1) for k, x in tot.items() if k.startswith("L_")})
    print(f"B={B}: inside voxels {v} ({v / (2 * B):.1f} per hand); survivors per voxel with ub = exact minimum: "
          f"sphere {tot['sph'] / v:.1f}, plane+circle {tot['pc'] / v:.1f}, fp16 sphere {tot['sph16'] / v:.1f}, fp16 plane+circle {tot['pc16'] / v:.1f}, min-enclosing-circle sphere {tot['mec_s'] / v:.1f} plane+circle {tot['mec'] / v:.1f} (R ratio {np.mean(tot['Rratio']):.3f})")

main()
