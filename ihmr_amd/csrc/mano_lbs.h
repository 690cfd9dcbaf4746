// MANO linear blend skinning -- forward and analytic backward, one workgroup per hand.
//
// Replaces smplx 0.1.28 `MANO.forward` + `lbs()` as the reference calls it
// (models/optimize_model.py:194-198) and, in TWO_HAND mode, the whole of
// `OptimizeModel.get_mano_output` (:171-232): mirror of the left axis-angles (:180-188), right-hand
// model on 2B hands, 5 fingertip vertices appended (:201-202), x-negation of the left outputs
// (:210-211), left hand shifted by hand_trans + (right wrist - left wrist) (:222-228).
//
// Layout: the skeleton state (16 rotations, rest joints, chain transforms) lives in LDS; posedirs
// (1.26 MB) / shapedirs are streamed from L2 with lane <-> consecutive (vertex,coord) so every
// wave-load is a contiguous 768 B run.  J = J_template + J_shapedirs.beta is precomputed
// algebra (J_regressor is linear), which removes the 16x778 regression from the per-iteration path.
#pragma once
#include "ihmr_common.h"

#define LBS_THREADS 256

struct LbsShared {
    float pose[48];     // full pose (+ mean), mirrored for left hands in TWO_HAND mode
    float beta[10];
    float R[NJ][9];
    float J[NJ][3];
    float G[NJ][12];    // world transform rows [R | t]
    float A[NJ][12];    // skinning transform [G.R | G.t - G.R J]
    float pf[NPF];
    float tip[IHMR_NUM_TIPS][3];
    float shift[3];
    float red[LBS_THREADS];
};

// smplx batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1 - cos) K^2, K = skew(r / angle)
__device__ __forceinline__ void rodrigues_fwd(const float* r, float* R) {
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
__device__ __forceinline__ void rodrigues_bwd(const float* r, const float* dR, float* dr) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inva = 1.0f / a;
    const float n[3] = {r[0] * inva, r[1] * inva, r[2] * inva};
    const float s = sinf(a), c = cosf(a), c1 = 1.0f - c;
    const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    // K = skew(n)
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
    // from K: dn_x = dK[2][1] - dK[1][2], dn_y = dK[0][2] - dK[2][0], dn_z = dK[1][0] - dK[0][1]
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

// Skeleton set-up shared by forward and backward: fills sh.pose/beta/R/J/pf/G/A.
// `mirror`: negate the y,z components of every axis-angle triple (left hand through the right model).
__device__ __forceinline__ void lbs_skeleton(const ihmr_mano& m, LbsShared& sh, const float* orient, const float* pose,
                                             const float* betas, bool mirror) {
    const int tid = threadIdx.x;
    if (tid < 48) {
        float v = tid < 3 ? orient[tid] : pose[tid - 3];
        if (mirror && (tid % 3) != 0) v = -v;
        sh.pose[tid] = v + m.pose_mean[tid];
    }
    if (tid >= 64 && tid < 74) sh.beta[tid - 64] = betas[tid - 64];
    __syncthreads();
    if (tid < NJ) rodrigues_fwd(&sh.pose[3 * tid], sh.R[tid]);
    if (tid >= 64 && tid < 64 + 48) {
        const int e = tid - 64;
        float acc = m.J_template[e];
#pragma unroll
        for (int l = 0; l < 10; ++l) acc = __builtin_fmaf(m.J_shapedirs[e * 10 + l], sh.beta[l], acc);
        sh.J[e / 3][e % 3] = acc;
    }
    __syncthreads();
    if (tid < NPF) {
        const int j = 1 + tid / 9, e = tid % 9;
        sh.pf[tid] = sh.R[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    // kinematic chain, level by level (MANO: depth <= 3); 12 lanes per joint
    if (tid < 12) {
        const int r = tid / 4, c = tid % 4;
        sh.G[0][tid] = c < 3 ? sh.R[0][3 * r + c] : sh.J[0][r];
    }
    __syncthreads();
    for (int d = 1; d <= m.max_depth; ++d) {
        if (tid < NJ * 12) {
            const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
            if (m.depth[j] == d) {
                const int p = m.parents[j];
                const float* Gp = sh.G[p];
                float acc;
                if (c < 3) {
                    acc = Gp[4 * r + 0] * sh.R[j][c] + Gp[4 * r + 1] * sh.R[j][3 + c] + Gp[4 * r + 2] * sh.R[j][6 + c];
                } else {
                    const float rx = sh.J[j][0] - sh.J[p][0], ry = sh.J[j][1] - sh.J[p][1], rz = sh.J[j][2] - sh.J[p][2];
                    acc = Gp[4 * r + 0] * rx + Gp[4 * r + 1] * ry + Gp[4 * r + 2] * rz + Gp[4 * r + 3];
                }
                sh.G[j][e] = acc;
            }
        }
        __syncthreads();
    }
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        const float* G = sh.G[j];
        sh.A[j][e] = c < 3 ? G[e]
                           : G[4 * r + 3] - (G[4 * r + 0] * sh.J[j][0] + G[4 * r + 1] * sh.J[j][1] + G[4 * r + 2] * sh.J[j][2]);
    }
    __syncthreads();
}

// skinning transform of vertex v: T (3x4 row-major) = sum_j W[v][j] A_j
__device__ __forceinline__ void lbs_blend(const ihmr_mano& m, const LbsShared& sh, int v, float* T) {
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    const float4* w4 = reinterpret_cast<const float4*>(m.weights + v * NJ);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = w4[q];
        const float ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float wj = ws[i];
            if (wj != 0.f) {
                const float* A = sh.A[4 * q + i];
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = __builtin_fmaf(wj, A[e], T[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------- forward
// grid = N hands, block = 256.  TWO_HAND: hands [0,B) right, [B,2B) left of sample (h - B);
// verts out is (2,B,778,3) == (N,778,3); joints out is (B,42,3).
template <bool TWO_HAND>
__global__ __launch_bounds__(LBS_THREADS) void lbs_fwd_kernel(ihmr_mano m, const float* __restrict__ orient,
                                                              const float* __restrict__ pose,
                                                              const float* __restrict__ betas,
                                                              const float* __restrict__ trans, int B,
                                                              float* __restrict__ verts, float* __restrict__ joints,
                                                              float* __restrict__ v_posed_ws) {
    __shared__ LbsShared sh;
    const int h = blockIdx.x, tid = threadIdx.x;
    const bool left = TWO_HAND && h >= B;
    lbs_skeleton(m, sh, orient + h * 3, pose + h * 45, betas + h * 10, left);

    if (TWO_HAND) {
        if (left && tid < 3) {
            // right wrist of the same sample: J_r[0] = J_template[0] + J_shapedirs[0] . beta_right
            const float* br = betas + (h - B) * 10;
            float jr = m.J_template[tid];
#pragma unroll
            for (int l = 0; l < 10; ++l) jr = __builtin_fmaf(m.J_shapedirs[tid * 10 + l], br[l], jr);
            const float jl = tid == 0 ? -sh.J[0][0] : sh.J[0][tid];  // mirrored left wrist
            sh.shift[tid] = trans[(h - B) * 3 + tid] + (jr - jl);
        }
        __syncthreads();
    }

    const float* vt = m.v_template;
    const float* sd = m.shapedirs_t;
    const float* pd = m.posedirs;
    for (int v = tid; v < NV; v += LBS_THREADS) {
        float vs[3], vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float acc = vt[3 * v + k];
#pragma unroll
            for (int l = 0; l < 10; ++l) acc = __builtin_fmaf(sd[l * NV3 + 3 * v + k], sh.beta[l], acc);
            vs[k] = acc;
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll 5
        for (int e = 0; e < NPF; ++e) {
            const float f = sh.pf[e];
            const float* row = pd + e * NV3 + 3 * v;
            o0 = __builtin_fmaf(f, row[0], o0);
            o1 = __builtin_fmaf(f, row[1], o1);
            o2 = __builtin_fmaf(f, row[2], o2);
        }
        vp[0] = vs[0] + o0; vp[1] = vs[1] + o1; vp[2] = vs[2] + o2;
        float T[12];
        lbs_blend(m, sh, v, T);
        float out[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) out[r] = T[4 * r + 0] * vp[0] + T[4 * r + 1] * vp[1] + T[4 * r + 2] * vp[2] + T[4 * r + 3];
        float* ws = v_posed_ws + ((size_t)h * NV + v) * 3;
        ws[0] = vp[0]; ws[1] = vp[1]; ws[2] = vp[2];
        if (left) {
            out[0] = -out[0] + sh.shift[0];
            out[1] = out[1] + sh.shift[1];
            out[2] = out[2] + sh.shift[2];
        }
        float* dst = verts + ((size_t)h * NV + v) * 3;
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2];
        if (TWO_HAND) {
#pragma unroll
            for (int t = 0; t < IHMR_NUM_TIPS; ++t)
                if (v == m.tip_ids[t]) { sh.tip[t][0] = out[0]; sh.tip[t][1] = out[1]; sh.tip[t][2] = out[2]; }
        }
    }
    if (!TWO_HAND) {
        if (tid < NJ * 3) joints[(size_t)h * NJ * 3 + tid] = sh.G[tid / 3][4 * (tid % 3) + 3];
    } else {
        __syncthreads();
        if (tid < 21 * 3) {
            const int j = tid / 3, k = tid % 3;
            float val;
            if (j < NJ) {
                val = sh.G[j][4 * k + 3];
                if (left) val = (k == 0 ? -val : val) + sh.shift[k];
            } else {
                val = sh.tip[j - NJ][k];
            }
            const int b = left ? h - B : h;
            joints[((size_t)b * 42 + (left ? 21 : 0) + j) * 3 + k] = val;
        }
    }
}

// ------------------------------------------------------------------------------------- backward
// Inputs are gradients w.r.t. the kernel's OUTPUTS (final verts / joints).  Outputs:
//   d_orient (N,3), d_pose (N,45), d_betas (N,10) in the caller's (un-mirrored) parametrisation,
//   TWO_HAND: d_trans (B,3) written by the left-hand workgroup.
// need_mask: bit0 orient, bit1 pose, bit2 betas, bit3 trans.
struct LbsBwdShared {
    float dvp[NV3];       // d L / d v_posed
    float g[NV3];         // d L / d verts (raw hand frame)
    float dA[NJ][12];
    float dG[NJ][12];
    float dR[NJ][9];
    float dJ[NJ][3];
    float dpf[NPF];
    float gsum[3];        // TWO_HAND: sum of the left-hand output gradients (= d L / d shift)
    float gj[21][3];      // joint gradients (raw hand frame)
};

template <bool TWO_HAND>
__global__ __launch_bounds__(LBS_THREADS) void lbs_bwd_kernel(ihmr_mano m, const float* __restrict__ orient,
                                                              const float* __restrict__ pose,
                                                              const float* __restrict__ betas, int B,
                                                              const float* __restrict__ v_posed_ws,
                                                              const float* __restrict__ d_verts,
                                                              const float* __restrict__ d_joints,
                                                              float* __restrict__ d_orient, float* __restrict__ d_pose,
                                                              float* __restrict__ d_betas, float* __restrict__ d_trans,
                                                              int need_mask) {
    __shared__ LbsShared sh;
    __shared__ LbsBwdShared bw;
    const int h = blockIdx.x, tid = threadIdx.x;
    const bool left = TWO_HAND && h >= B;
    const int b = TWO_HAND ? (left ? h - B : h) : 0;
    const bool need_orient = need_mask & 1, need_pose = need_mask & 2, need_betas = need_mask & 4, need_trans = need_mask & 8;

    // ---- TWO_HAND: d L / d shift = sum over the LEFT hand's vertex and joint gradients of this sample
    if (TWO_HAND) {
        const float* gl = d_verts + ((size_t)(B + b) * NV) * 3;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int v = tid; v < NV; v += LBS_THREADS) { s0 += gl[3 * v]; s1 += gl[3 * v + 1]; s2 += gl[3 * v + 2]; }
        if (tid < 21) {
            const float* gj = d_joints + ((size_t)b * 42 + 21 + tid) * 3;
            s0 += gj[0]; s1 += gj[1]; s2 += gj[2];
        }
        s0 = block_reduce_sum(s0, sh.red);
        s1 = block_reduce_sum(s1, sh.red);
        s2 = block_reduce_sum(s2, sh.red);
        if (tid == 0) { bw.gsum[0] = s0; bw.gsum[1] = s1; bw.gsum[2] = s2; }
        if (left && need_trans && tid < 3) d_trans[b * 3 + tid] = tid == 0 ? s0 : (tid == 1 ? s1 : s2);
        __syncthreads();
        if ((need_mask & 7) == 0) return;  // stage 0: only the translation moves
    }

    lbs_skeleton(m, sh, orient + h * 3, pose + h * 45, betas + h * 10, left);

    // ---- load output gradients into the raw hand frame
    for (int i = tid; i < NV3; i += LBS_THREADS) {
        float gv = d_verts[(size_t)h * NV3 + i];
        if (left && (i % 3) == 0) gv = -gv;
        bw.g[i] = gv;
    }
    if (tid < 21 * 3) {
        const int j = tid / 3, k = tid % 3;
        float gv;
        if (TWO_HAND) {
            gv = d_joints[((size_t)b * 42 + (left ? 21 : 0) + j) * 3 + k];
            if (left && k == 0) gv = -gv;
        } else {
            gv = j < NJ ? d_joints[((size_t)h * NJ + j) * 3 + k] : 0.f;
        }
        bw.gj[j][k] = gv;
    }
    __syncthreads();
    if (TWO_HAND && tid < IHMR_NUM_TIPS * 3) {  // fingertip joints are vertices
        const int t = tid / 3, k = tid % 3;
        bw.g[3 * m.tip_ids[t] + k] += bw.gj[NJ + t][k];
    }
    __syncthreads();

    // ---- per vertex: d v_posed = T.R^T g
    for (int v = tid; v < NV; v += LBS_THREADS) {
        float T[12];
        lbs_blend(m, sh, v, T);
        const float g0 = bw.g[3 * v], g1 = bw.g[3 * v + 1], g2 = bw.g[3 * v + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) bw.dvp[3 * v + c] = T[c] * g0 + T[4 + c] * g1 + T[8 + c] * g2;
    }
    __syncthreads();

    // ---- dA[j][e] = sum_v W[v][j] * [g (x) v_posed | g][e]  (CSR by joint, fixed order)
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        const float* vp = v_posed_ws + (size_t)h * NV3;
        float acc = 0.f;
        for (int q = m.wj_start[j]; q < m.wj_start[j + 1]; ++q) {
            const int v = m.wj_vert[q];
            const float gr = bw.g[3 * v + r];
            const float x = c < 3 ? gr * vp[3 * v + c] : gr;
            acc = __builtin_fmaf(m.wj_w[q], x, acc);
        }
        bw.dA[j][e] = acc;
    }
    // ---- d pose_feature[e] = posedirs[e] . dvp   (one wave per row, lanes across the 2334 columns)
    if (need_pose) {
        const int wave = tid / WAVE, lane = tid % WAVE;
        for (int e = wave; e < NPF; e += LBS_THREADS / WAVE) {
            const float* row = m.posedirs + (size_t)e * NV3;
            float acc = 0.f;
            for (int i = lane; i < NV3; i += WAVE) acc = __builtin_fmaf(row[i], bw.dvp[i], acc);
            acc = wave_reduce_sum(acc);
            if (lane == 0) bw.dpf[e] = acc;
        }
    } else if (tid < NPF) {
        bw.dpf[tid] = 0.f;
    }
    __syncthreads();

    // ---- chain backward (joint-serial, level by level from the leaves)
    // dG_j = [dA_j.R - dA_j.t (x) J_j | dA_j.t + d posed_joint_j];  dJ_j(direct) = -G_j.R^T dA_j.t
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        const float dat = bw.dA[j][4 * r + 3];
        bw.dG[j][e] = c < 3 ? bw.dA[j][e] - dat * sh.J[j][c] : dat + bw.gj[j][r];
    }
    if (tid >= 192 && tid < 192 + NJ * 3) {
        const int j = (tid - 192) / 3, c = (tid - 192) % 3;
        const float* G = sh.G[j];
        bw.dJ[j][c] = -(G[c] * bw.dA[j][3] + G[4 + c] * bw.dA[j][7] + G[8 + c] * bw.dA[j][11]);
    }
    __syncthreads();
    for (int d = m.max_depth; d >= 1; --d) {
        // children at depth d push into their parents; siblings share a parent, so one lane per parent
        if (tid < NJ) {
            const int p = tid;
            for (int j = 1; j < NJ; ++j) {
                if (m.parents[j] != p || m.depth[j] != d) continue;
                const float* dGj = bw.dG[j];
                const float* Rj = sh.R[j];
                const float rel[3] = {sh.J[j][0] - sh.J[p][0], sh.J[j][1] - sh.J[p][1], sh.J[j][2] - sh.J[p][2]};
                const float* Gp = sh.G[p];
                // dR_j = Gp.R^T dG_j.R ; drel_j = Gp.R^T dG_j.t
                float drel[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2)
                        bw.dR[j][3 * c + c2] = Gp[c] * dGj[c2] + Gp[4 + c] * dGj[4 + c2] + Gp[8 + c] * dGj[8 + c2];
                    drel[c] = Gp[c] * dGj[3] + Gp[4 + c] * dGj[7] + Gp[8 + c] * dGj[11];
                }
                // dGp.R += dG_j.R R_j^T + dG_j.t (x) rel ; dGp.t += dG_j.t
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        bw.dG[p][4 * r + c] += dGj[4 * r] * Rj[3 * c] + dGj[4 * r + 1] * Rj[3 * c + 1] +
                                               dGj[4 * r + 2] * Rj[3 * c + 2] + dGj[4 * r + 3] * rel[c];
                    bw.dG[p][4 * r + 3] += dGj[4 * r + 3];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) { bw.dJ[j][c] += drel[c]; bw.dJ[p][c] -= drel[c]; }
            }
        }
        __syncthreads();
    }
    if (tid < 9) bw.dR[0][tid] = bw.dG[0][4 * (tid / 3) + (tid % 3)];
    if (tid >= 64 && tid < 67) bw.dJ[0][tid - 64] += bw.dG[0][4 * (tid - 64) + 3];
    __syncthreads();

    // ---- pose gradients through Rodrigues (pose-feature term enters R_1..15 directly)
    if (tid < NJ && ((tid == 0 && need_orient) || (tid > 0 && need_pose))) {
        float dR[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) dR[e] = bw.dR[tid][e] + (tid > 0 ? bw.dpf[(tid - 1) * 9 + e] : 0.f);
        float dr[3];
        rodrigues_bwd(&sh.pose[3 * tid], dR, dr);
        if (left) { dr[1] = -dr[1]; dr[2] = -dr[2]; }
        float* dst = tid == 0 ? d_orient + h * 3 : d_pose + h * 45 + 3 * (tid - 1);
        dst[0] = dr[0]; dst[1] = dr[1]; dst[2] = dr[2];
    }

    // ---- shape gradients: d beta_l = shapedirs_l . d v_shaped + J_shapedirs_l . dJ   (d v_shaped = d v_posed)
    if (need_betas) {
        if (TWO_HAND) {
            __syncthreads();
            // d shift reaches the right wrist (+) and the mirrored left wrist (-S)
            if (tid < 3) {
                if (!left) bw.dJ[0][tid] += bw.gsum[tid];
                else bw.dJ[0][tid] += tid == 0 ? bw.gsum[0] : -bw.gsum[tid];
            }
            __syncthreads();
        }
        const int wave = tid / WAVE, lane = tid % WAVE;
        for (int l = wave; l < 10; l += LBS_THREADS / WAVE) {
            const float* row = m.shapedirs_t + (size_t)l * NV3;
            float acc = 0.f;
            for (int i = lane; i < NV3; i += WAVE) acc = __builtin_fmaf(row[i], bw.dvp[i], acc);
            if (lane < 48) acc = __builtin_fmaf(m.J_shapedirs[lane * 10 + l], bw.dJ[lane / 3][lane % 3], acc);
            acc = wave_reduce_sum(acc);
            if (lane == 0) d_betas[h * 10 + l] = acc;
        }
    }
}
