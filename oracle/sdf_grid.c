/*
 * ORACLE (test infrastructure, never shipped / never on the product path).
 *
 * CPU restatement of the voxel signed-distance grid that the reference's collision loss builds
 * through the third-party CUDA extension `sdf` (penincillin/SDF_ihmr, unpinned; parent project
 * JiangWenPL/multiperson `sdf`).  Call site: reference src/models/loss_utils.py:13,38,181-182
 * (`SDFLoss(faces_right, faces_left, robustifier)` / `self.sdf_loss(hand_verts, ...)`).
 *
 * PARITY UNPINNED: the extension's source is not under /root/reference and no reference test pins
 * its output, so the semantics are fixed HERE (DESIGN.md "SDF arithmetic spec"):
 *
 *   grid G^3 (G = 32), voxel (k,j,i) has centre p = (-1 + (2i+1)/G, -1 + (2j+1)/G, -1 + (2k+1)/G)
 *   (the sample positions of torch grid_sample with align_corners=False), memory order phi[k][j][i];
 *   inside(p)  <=>  the ray from p along +x crosses an odd number of triangles
 *                   (Moeller-Trumbore, hit iff |det| >= 1e-12, 0<=u<=1, v>=0, u+v<=1, t>0);
 *   phi(p) = inside ? min over triangles of the Euclidean point-triangle distance : 0.
 *
 * Dense and plain on purpose: one voxel at a time, every triangle, no acceleration structure.
 * All dot products are evaluated as fmaf(x2,y2,fmaf(x1,y1,x0*y0)) and the file is compiled with
 * -ffp-contract=off so that the HIP kernels (which use the same operation order) can be compared
 * bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define DOT3(ax, ay, az, bx, by, bz) fmaf((az), (bz), fmaf((ay), (by), (ax) * (bx)))

/* 1 if the +x ray from p crosses triangle (a,b,c) at t > 0 */
static inline int ray_hit_px(const float* a, const float* b, const float* c, float px, float py, float pz) {
    const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const float det = fmaf(e1z, e2y, -(e1y * e2z));
    if (fabsf(det) < 1e-12f) return 0;
    const float inv = 1.0f / det;
    const float sy = py - a[1], sz = pz - a[2];
    const float u = fmaf(sz, e2y, -(sy * e2z)) * inv;
    if (u < 0.0f || u > 1.0f) return 0;
    const float qx = fmaf(sy, e1z, -(sz * e1y));
    const float v = qx * inv;
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float sx = px - a[0];
    const float qy = fmaf(sz, e1x, -(sx * e1z));
    const float qz = fmaf(sx, e1y, -(sy * e1x));
    const float t = DOT3(e2x, e2y, e2z, qx, qy, qz) * inv;
    return t > 0.0f;
}

/* squared distance from p to triangle (a,b,c): closest-point by Voronoi regions */
static inline float point_tri_dist2(const float* a, const float* b, const float* c, float px, float py, float pz) {
    const float abx = b[0] - a[0], aby = b[1] - a[1], abz = b[2] - a[2];
    const float acx = c[0] - a[0], acy = c[1] - a[1], acz = c[2] - a[2];
    const float apx = px - a[0], apy = py - a[1], apz = pz - a[2];
    const float d1 = DOT3(abx, aby, abz, apx, apy, apz);
    const float d2 = DOT3(acx, acy, acz, apx, apy, apz);
    float cx, cy, cz;
    if (d1 <= 0.0f && d2 <= 0.0f) {
        cx = a[0]; cy = a[1]; cz = a[2];
    } else {
        const float bpx = px - b[0], bpy = py - b[1], bpz = pz - b[2];
        const float d3 = DOT3(abx, aby, abz, bpx, bpy, bpz);
        const float d4 = DOT3(acx, acy, acz, bpx, bpy, bpz);
        const float vc = fmaf(d1, d4, -(d3 * d2));
        const float cpx = px - c[0], cpy = py - c[1], cpz = pz - c[2];
        const float d5 = DOT3(abx, aby, abz, cpx, cpy, cpz);
        const float d6 = DOT3(acx, acy, acz, cpx, cpy, cpz);
        const float vb = fmaf(d5, d2, -(d1 * d6));
        const float va = fmaf(d3, d6, -(d5 * d4));
        if (d3 >= 0.0f && d4 <= d3) {
            cx = b[0]; cy = b[1]; cz = b[2];
        } else if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
            const float v = d1 / (d1 - d3);
            cx = fmaf(v, abx, a[0]); cy = fmaf(v, aby, a[1]); cz = fmaf(v, abz, a[2]);
        } else if (d6 >= 0.0f && d5 <= d6) {
            cx = c[0]; cy = c[1]; cz = c[2];
        } else if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
            const float w = d2 / (d2 - d6);
            cx = fmaf(w, acx, a[0]); cy = fmaf(w, acy, a[1]); cz = fmaf(w, acz, a[2]);
        } else if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
            const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
            cx = fmaf(w, c[0] - b[0], b[0]); cy = fmaf(w, c[1] - b[1], b[1]); cz = fmaf(w, c[2] - b[2], b[2]);
        } else {
            const float denom = 1.0f / (va + vb + vc);
            const float v = vb * denom, w = vc * denom;
            cx = fmaf(acx, w, fmaf(abx, v, a[0]));
            cy = fmaf(acy, w, fmaf(aby, v, a[1]));
            cz = fmaf(acz, w, fmaf(abz, v, a[2]));
        }
    }
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    return DOT3(dx, dy, dz, dx, dy, dz);
}

/*
 * verts_n : (H, V, 3) vertices already normalised into [-1,1]^3 (per hand)
 * faces   : (F, 3) int32 vertex ids (shared by all H meshes)
 * phi     : (H, G, G, G) out, phi[h][k][j][i]
 */
void ihmr_oracle_sdf_grid(const float* verts_n, const int32_t* faces, int H, int V, int F, int G, float* phi) {
    const long nvox = (long)G * G * G;
#pragma omp parallel for schedule(dynamic, 64) collapse(2)
    for (int h = 0; h < H; ++h) {
        for (long vox = 0; vox < nvox; ++vox) {
            const float* vb = verts_n + (long)h * V * 3;
            const int i = (int)(vox % G), j = (int)((vox / G) % G), k = (int)(vox / ((long)G * G));
            const float px = (float)(2 * i + 1) / (float)G - 1.0f;
            const float py = (float)(2 * j + 1) / (float)G - 1.0f;
            const float pz = (float)(2 * k + 1) / (float)G - 1.0f;
            int hits = 0;
            for (int f = 0; f < F; ++f) {
                const float* a = vb + 3 * faces[3 * f + 0];
                const float* b = vb + 3 * faces[3 * f + 1];
                const float* c = vb + 3 * faces[3 * f + 2];
                hits += ray_hit_px(a, b, c, px, py, pz);
            }
            float out = 0.0f;
            if (hits & 1) {
                float best = INFINITY;
                for (int f = 0; f < F; ++f) {
                    const float* a = vb + 3 * faces[3 * f + 0];
                    const float* b = vb + 3 * faces[3 * f + 1];
                    const float* c = vb + 3 * faces[3 * f + 2];
                    const float d2 = point_tri_dist2(a, b, c, px, py, pz);
                    if (d2 < best) best = d2;
                }
                out = sqrtf(best);
            }
            phi[(long)h * nvox + vox] = out;
        }
    }
}

/* single-point probes used by the known-answer tests */
float ihmr_oracle_point_tri_dist2(const float* a, const float* b, const float* c, const float* p) {
    return point_tri_dist2(a, b, c, p[0], p[1], p[2]);
}
int ihmr_oracle_ray_hit_px(const float* a, const float* b, const float* c, const float* p) {
    return ray_hit_px(a, b, c, p[0], p[1], p[2]);
}
