// Pure arithmetic helpers of libihmr_hip -- the functions whose bits the parity claims rest on -- compilable for the HOST as well:
// under hipcc they are `__host__ __device__` (the kernels inline them exactly as before), under a plain C++ compiler they are ordinary
// inline functions, so that the GPU-less container can unit-test them under AddressSanitizer / UBSan and compare them bit for bit with
// the oracle (tests/test_pure_host_cpu.py builds tests/pure_host_driver.cpp with g++ -fsanitize=address,undefined).
// Nothing here touches memory other than its arguments; no wave intrinsics, no LDS, no globals.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/ihmr_hip.h"

#if defined(__HIPCC__)
#define IHMR_PURE __host__ __device__ __forceinline__
#else
#define IHMR_PURE static inline
#endif
#ifndef SDF_G
#define SDF_G IHMR_SDF_GRID    // 32
#endif

// dot product in the fixed order of the SDF arithmetic spec: fma(z, fma(y, x*x'))
#define DOT3(ax, ay, az, bx, by, bz) __builtin_fmaf((az), (bz), __builtin_fmaf((ay), (by), (ax) * (bx)))
IHMR_PURE int ihmr_imax(int a, int b) { return a > b ? a : b; }
IHMR_PURE int ihmr_imin(int a, int b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------- collision: normalisation, rays, distances
// grid_sample un-normalisation: align_corners = False: ((x + 1) * G - 1) / 2;  True: (x + 1) / 2 * (G - 1)   (torch's expressions)
IHMR_PURE float sdf_unnorm(float x, int align_corners) {
    return align_corners ? ((x + 1.0f) / 2.0f) * (float)(SDF_G - 1) : ((x + 1.0f) * (float)SDF_G - 1.0f) / 2.0f;
}

// a / b, correctly rounded (the bits of the IEEE division the oracle's `(v - c) / s` performs), for MANY numerators over ONE divisor:
// with y = RN(1 / b) (one IEEE division per hand), q0 = RN(a y), r = a - b q0 (exact in one fma), RN(q0 + r y) is the correctly rounded
// quotient (Markstein's theorem; checked against exact rational arithmetic on 2 x 10^5 random pairs, scripts/experiments/markstein_division.py)
// -- three instructions instead of the ~10 of the division expansion, six times per vertex pair.  Needs b and the results in the normal
// range: the caller takes this path only for a box scale in [1e-6, 1e6] (else the plain division: degenerate hands keep the oracle's infs / NaNs).
struct SdfDivisor { float b, y; bool fast; };
IHMR_PURE SdfDivisor sdf_divisor(float b) { return SdfDivisor{b, 1.0f / b, b >= 1e-6f && b <= 1e6f}; }
IHMR_PURE float sdf_div(float a, const SdfDivisor& d) {
    if (!d.fast) return a / d.b;             // (uniform per hand)
    const float q0 = a * d.y;
    const float r = __builtin_fmaf(-d.b, q0, a);
    return __builtin_fmaf(r, d.y, q0);
}

IHMR_PURE void tri_col_range(float y0, float y1, float y2, float z0, float z1, float z2, int& j0, int& j1,
                                              int& k0, int& k1) {
    // columns whose ray (py, pz) can cross the triangle: centres inside the yz bounding box (+ margin)
    const float ymin = fminf(y0, fminf(y1, y2)) - 1e-4f, ymax = fmaxf(y0, fmaxf(y1, y2)) + 1e-4f;
    const float zmin = fminf(z0, fminf(z1, z2)) - 1e-4f, zmax = fmaxf(z0, fmaxf(z1, z2)) + 1e-4f;
    j0 = ihmr_imax(0, (int)ceilf((ymin + 1.0f) * 16.0f - 0.5f));
    j1 = ihmr_imin(SDF_G - 1, (int)floorf((ymax + 1.0f) * 16.0f - 0.5f));
    k0 = ihmr_imax(0, (int)ceilf((zmin + 1.0f) * 16.0f - 0.5f));
    k1 = ihmr_imin(SDF_G - 1, (int)floorf((zmax + 1.0f) * 16.0f - 0.5f));
}

// t of the +x ray from voxel centre i of a column against one triangle (already known to pass the (u,v) test).
// Same operation order as oracle/sdf_grid.c ray_hit_px.
IHMR_PURE float sdf_ray_t(int i, float ax, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                           float inv, float sy, float sz, float qx) {
    const float px = (float)(2 * i + 1) / (float)SDF_G - 1.0f;
    const float sx = px - ax;
    const float qy = __builtin_fmaf(sz, e1x, -(sx * e1z));
    const float qz = __builtin_fmaf(sx, e1y, -(sy * e1x));
    return DOT3(e2x, e2y, e2z, qx, qy, qz) * inv;
}

// t > 0 test of the needed voxels of one column; returns the hit mask, bit for bit what evaluating sdf_ray_t for
// every needed voxel gives -- without the per-voxel loop.  In real arithmetic t_i = x* - px_i: it falls by exactly
// 1/16 per voxel (up to the rounding of det and 1/det, delta below), so one evaluation at the lowest needed
// voxel places the crossing index ic = lo + 16 t_lo.  Voxels a whole index away from ic have |t| >= 1/16, far
// above the rounding error E of the float expression (bounded per triangle by sdf_ray_tri_safe), so their sign is known;
// the (at most two) voxels next to the crossing are evaluated with the exact expression.  Whenever the bound
// does not hold (near-degenerate triangles: huge 1/det) every needed voxel is evaluated.
// Per-triangle part of the error bound of sdf_ray_hits: true iff, for EVERY column and voxel of the grid, the float
// expression of t differs from the real one by well under a voxel step.  |float t - real t| <= E = 16 u S |1/det| with
// S the sum of the magnitudes of the terms (|s| <= 2 anywhere in the [-1,1]^3 grid), and the real slope of t
// along x is -(1 + delta)/16 with |delta| <= 4 u (|e1z e2y| + |e1y e2z|) |1/det| + 4 u  (rounding of det and 1/det).
IHMR_PURE bool sdf_ray_tri_safe(float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, float inv) {
    const float U = 5.9604645e-8f;   // 2^-24
    const float ainv = fabsf(inv), smax = 2.0f;
    const float qx_max = smax * (fabsf(e1z) + fabsf(e1y));
    const float S = fabsf(e2x) * qx_max + fabsf(e2y) * smax * (fabsf(e1x) + fabsf(e1z)) + fabsf(e2z) * smax * (fabsf(e1y) + fabsf(e1x));
    const float E = 16.0f * U * S * ainv;
    const float delta = 4.0f * U * (fabsf(e1z * e2y) + fabsf(e1y * e2z)) * ainv + 4.0f * U;
    return E + 2.0f * delta < (1.0f / 64.0f);
}

IHMR_PURE unsigned sdf_ray_hits(unsigned need, bool tri_safe, float ax, float e1x, float e1y, float e1z, float e2x,
                                                 float e2y, float e2z, float inv, float sy, float sz, float qx) {
    const int lo = __builtin_ffs((int)need) - 1;
    const float t_lo = sdf_ray_t(lo, ax, e1x, e1y, e1z, e2x, e2y, e2z, inv, sy, sz, qx);
    if (!tri_safe) {
        unsigned hits = 0, rem = need;
        while (rem) {
            const int i = __builtin_ffs((int)rem) - 1;
            rem &= rem - 1;
            if (sdf_ray_t(i, ax, e1x, e1y, e1z, e2x, e2y, e2z, inv, sy, sz, qx) > 0.0f) hits |= 1u << i;
        }
        return hits;
    }
    const float ic = fminf(fmaxf((float)lo + 16.0f * t_lo, -2.0f), 34.0f);
    const int i1 = (int)floorf(ic), i2 = i1 + 1;
    unsigned hits = need & (i1 <= 0 ? 0u : (i1 >= 32 ? 0xffffffffu : ((1u << i1) - 1u)));   // voxels below the crossing: t > 0
    if (i1 >= 0 && i1 < SDF_G && ((need >> i1) & 1u) &&
        sdf_ray_t(i1, ax, e1x, e1y, e1z, e2x, e2y, e2z, inv, sy, sz, qx) > 0.0f) hits |= 1u << i1;
    if (i2 >= 0 && i2 < SDF_G && ((need >> i2) & 1u) &&
        sdf_ray_t(i2, ax, e1x, e1y, e1z, e2x, e2y, e2z, inv, sy, sz, qx) > 0.0f) hits |= 1u << i2;
    return hits;
}

// The +x rays of ONE grid column (k, j) = col against one triangle: the (u, v) test is shared by the column's 32 voxels (uv_pass), the
// t > 0 test of the needed ones is sdf_ray_hits' loop-free mask.  The caller has already dropped triangles that are degenerate in yz
// (|det| < 1e-12: sdf_tri_yz_det).  Expression for expression oracle/sdf_grid.c ray_hit_px.
IHMR_PURE float sdf_tri_yz_det(float ay, float az, float by, float bz, float cy, float cz) {
    const float e1y = by - ay, e1z = bz - az, e2y = cy - ay, e2z = cz - az;
    return __builtin_fmaf(e1z, e2y, -(e1y * e2z));
}
IHMR_PURE unsigned sdf_ray_column_hits(const float* a, const float* b, const float* c, int col, unsigned need, bool& uv_pass) {
    const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const float det = __builtin_fmaf(e1z, e2y, -(e1y * e2z));
    const float inv = 1.0f / det;
    const int j = col & (SDF_G - 1), k = col >> 5;
    const float py = (float)(2 * j + 1) / (float)SDF_G - 1.0f;
    const float pz = (float)(2 * k + 1) / (float)SDF_G - 1.0f;
    const float sy = py - a[1], sz = pz - a[2];
    const float uu = __builtin_fmaf(sz, e2y, -(sy * e2z)) * inv;
    const float qx = __builtin_fmaf(sy, e1z, -(sz * e1y));
    const float vv = qx * inv;
    uv_pass = (uu >= 0.0f) && (uu <= 1.0f) && (vv >= 0.0f) && (uu + vv <= 1.0f);
    if (!uv_pass) return 0u;
    const bool tri_safe = sdf_ray_tri_safe(e1x, e1y, e1z, e2x, e2y, e2z, inv);
    return sdf_ray_hits(need, tri_safe, a[0], e1x, e1y, e1z, e2x, e2y, e2z, inv, sy, sz, qx);
}

// squared distance point -> triangle, closest point by Voronoi region.  Same values, operation for
// operation, as oracle/sdf_grid.c point_tri_dist2 -- but branch-free: the region is a priority select, the
// (at most one) quotient every region needs goes through ONE IEEE division, and the closest point is
// q = fma(dir2, t2, fma(dir1, t1, base)) with zeros where a region has fewer terms (fma(x, 0, y) == y exactly).
IHMR_PURE float sdf_point_tri_dist2(const float* a, const float* b, const float* c, float px, float py,
                                                     float pz) {
    const float abx = b[0] - a[0], aby = b[1] - a[1], abz = b[2] - a[2];
    const float acx = c[0] - a[0], acy = c[1] - a[1], acz = c[2] - a[2];
    const float apx = px - a[0], apy = py - a[1], apz = pz - a[2];
    const float d1 = DOT3(abx, aby, abz, apx, apy, apz);
    const float d2 = DOT3(acx, acy, acz, apx, apy, apz);
    const float bpx = px - b[0], bpy = py - b[1], bpz = pz - b[2];
    const float d3 = DOT3(abx, aby, abz, bpx, bpy, bpz);
    const float d4 = DOT3(acx, acy, acz, bpx, bpy, bpz);
    const float vc = __builtin_fmaf(d1, d4, -(d3 * d2));
    const float cpx = px - c[0], cpy = py - c[1], cpz = pz - c[2];
    const float d5 = DOT3(abx, aby, abz, cpx, cpy, cpz);
    const float d6 = DOT3(acx, acy, acz, cpx, cpy, cpz);
    const float vb = __builtin_fmaf(d5, d2, -(d1 * d6));
    const float va = __builtin_fmaf(d3, d6, -(d5 * d4));
    const float d43 = d4 - d3, d56 = d5 - d6;
    const bool r0 = (d1 <= 0.0f) && (d2 <= 0.0f);
    const bool r1 = !r0 && (d3 >= 0.0f) && (d4 <= d3);
    const bool r2 = !r0 && !r1 && (vc <= 0.0f) && (d1 >= 0.0f) && (d3 <= 0.0f);
    const bool r3 = !r0 && !r1 && !r2 && (d6 >= 0.0f) && (d5 <= d6);
    const bool r4 = !r0 && !r1 && !r2 && !r3 && (vb <= 0.0f) && (d2 >= 0.0f) && (d6 <= 0.0f);
    const bool r5 = !r0 && !r1 && !r2 && !r3 && !r4 && (va <= 0.0f) && (d43 >= 0.0f) && (d56 >= 0.0f);
    const bool r6 = !(r0 || r1 || r2 || r3 || r4 || r5);
    const float num = r2 ? d1 : (r4 ? d2 : (r5 ? d43 : (r6 ? 1.0f : 0.0f)));
    const float den = r2 ? (d1 - d3) : (r4 ? (d2 - d6) : (r5 ? (d43 + d56) : (r6 ? (va + vb + vc) : 1.0f)));
    const float t = num / den;
    const float t1 = r6 ? vb * t : ((r2 || r4 || r5) ? t : 0.0f);
    const float t2 = r6 ? vc * t : 0.0f;
    const bool base_b = r1 || r5, base_c = r3;
    const float bx = base_b ? b[0] : (base_c ? c[0] : a[0]);
    const float by = base_b ? b[1] : (base_c ? c[1] : a[1]);
    const float bz = base_b ? b[2] : (base_c ? c[2] : a[2]);
    const float d1x = r4 ? acx : (r5 ? c[0] - b[0] : abx);
    const float d1y = r4 ? acy : (r5 ? c[1] - b[1] : aby);
    const float d1z = r4 ? acz : (r5 ? c[2] - b[2] : abz);
    const float qx = __builtin_fmaf(acx, t2, __builtin_fmaf(d1x, t1, bx));
    const float qy = __builtin_fmaf(acy, t2, __builtin_fmaf(d1y, t1, by));
    const float qz = __builtin_fmaf(acz, t2, __builtin_fmaf(d1z, t1, bz));
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    return DOT3(dx, dy, dz, dx, dy, dz);
}

IHMR_PURE void sdf_vox_centre(int id, float& x, float& y, float& z) {
    x = (float)(2 * (id & 31) + 1) / (float)SDF_G - 1.0f;
    y = (float)(2 * ((id >> 5) & 31) + 1) / (float)SDF_G - 1.0f;
    z = (float)(2 * (id >> 10) + 1) / (float)SDF_G - 1.0f;
}

// ------------------------------------------------------------------------------------- MANO: Rodrigues, kinematic chain
// smplx batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1 - cos) K^2, K = skew(r / angle)
IHMR_PURE void rodrigues_fwd(const float* r, float* R) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float nx = r[0] / a, ny = r[1] / a, nz = r[2] / a;
    const float s = sinf(a), c1 = 1.0f - cosf(a);
    const float nn = nx * nx + ny * ny + nz * nz;
    // K^2 = n n^T - (n.n) I
    R[0] = 1.0f + c1 * (nx * nx - nn);
    R[1] = -s * nz + c1 * (nx * ny);
    R[2] = s * ny + c1 * (nx * nz);
    R[3] = s * nz + c1 * (ny * nx);
    R[4] = 1.0f + c1 * (ny * ny - nn);
    R[5] = -s * nx + c1 * (ny * nz);
    R[6] = -s * ny + c1 * (nz * nx);
    R[7] = s * nx + c1 * (nz * ny);
    R[8] = 1.0f + c1 * (nz * nz - nn);
}

// gradient of rodrigues_fwd: dR (3x3 row-major) -> dr (3)
IHMR_PURE void rodrigues_bwd(const float* r, const float* dR, float* dr) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inva = 1.0f / a;
    const float n[3] = {r[0] * inva, r[1] * inva, r[2] * inva};
    const float s = sinf(a), c = cosf(a), c1 = 1.0f - c;
    const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const float K[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
    float ds = 0.f, dc1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ds += dR[3 * i + j] * K[3 * i + j];
            const float k2 = n[i] * n[j] - (i == j ? nn : 0.f);
            dc1 += dR[3 * i + j] * k2;
        }
    // d/dn of  s*K(n) + c1*(n n^T - (n.n) I)
    float dn[3];
    dn[0] = s * (dR[7] - dR[5]);
    dn[1] = s * (dR[2] - dR[6]);
    dn[2] = s * (dR[3] - dR[1]);
    const float tr = dR[0] + dR[4] + dR[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += (dR[3 * i + j] + dR[3 * j + i]) * n[j];
        dn[i] += c1 * (acc - 2.0f * tr * n[i]);
    }
    const float ndn = n[0] * dn[0] + n[1] * dn[1] + n[2] * dn[2];
    const float da = c * ds + s * dc1 - ndn * inva;
    dr[0] = dn[0] * inva + da * ex * inva;
    dr[1] = dn[1] * inva + da * ey * inva;
    dr[2] = dn[2] * inva + da * ez * inva;
}

// One element (row r, column c of the 3 x 4 matrix) of a joint's global transform G_j = G_parent . [R_j | J_j - J_parent] (smplx
// batch_rigid_transform), and of its skinning matrix A_j = [G_j(3x3) | t_j - G_j(3x3) J_j]: the expressions of lbs_skel_hand, one
// output per thread there.
IHMR_PURE float lbs_chain_elem(const float* Gp /* parent, [3][4] */, const float* Rj /* [3][3] */, const float* Jj, const float* Jp, int r, int c) {
    if (c < 3) return Gp[4 * r + 0] * Rj[c] + Gp[4 * r + 1] * Rj[3 + c] + Gp[4 * r + 2] * Rj[6 + c];
    const float rx = Jj[0] - Jp[0], ry = Jj[1] - Jp[1], rz = Jj[2] - Jp[2];
    return Gp[4 * r + 0] * rx + Gp[4 * r + 1] * ry + Gp[4 * r + 2] * rz + Gp[4 * r + 3];
}
IHMR_PURE float lbs_rel_elem(const float* G /* [3][4] */, const float* Jj, int r, int c) {
    return c < 3 ? G[4 * r + c] : G[4 * r + 3] - (G[4 * r + 0] * Jj[0] + G[4 * r + 1] * Jj[1] + G[4 * r + 2] * Jj[2]);
}

// ------------------------------------------------------------------------------------- losses, optimizer
IHMR_PURE void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// root of loss_utils.py:90-98: weight > 0.5 -> joint 0, weight < 1e-7 -> joint 21, else no alignment
IHMR_PURE int align_root(float w) { return w > 0.5f ? 0 : (w < 1e-7f ? 21 : -1); }

// torch.optim.Adam's single-tensor update of ONE element, in torch's operation order (step_size = lr / (1 - beta1^t), bc2_sqrt =
// sqrt(1 - beta2^t), eps 1e-8, betas (0.9, 0.999): optimize_model.py:343-347); returns the new parameter, m / v updated in place
IHMR_PURE float opt_adam_update(float x, float g, float& m, float& v, float step_size, float bc2_sqrt) {
    m = m + 0.1f * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
    v = v * 0.999f;                          // exp_avg_sq.mul_(beta2)
    v = v + (0.001f * g) * g;                //            .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
    return x + (-step_size) * (m / denom);   // param.addcdiv_(exp_avg, denom, value = -step_size)
}
// torch.optim.SGD(momentum 0.9): buf.mul_(momentum).add_(grad) (the first step's buf = grad: the state starts at zero); param.add_(buf, alpha = -lr)
IHMR_PURE float opt_sgd_update(float x, float g, float& m, float lr) {
    m = m * 0.9f;
    m = m + g;
    return x + (-lr) * m;
}
