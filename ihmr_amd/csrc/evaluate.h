// Evaluation metrics on the device: the four numbers the reference prints after a run (optimize.py:98-102) --
// mpjpe_3d, inter_mpjpe_3d, collision_ave, collision_max -- as per-sample partial results that the host adds up
// in float64 (and all-reduces over ranks, ihmr_amd/dist.py).
//
// Reference: utils/metric_utils.py:23-38 (get_single_joints_error: per-hand MPJPE with the root subtraction
// applied CUMULATIVELY to the same copies), :107-117 (calc_transform_no_rot: per-axis mean / std alignment),
// :120-143 (get_single_pa_inter_joints_error, use_rot=False), utils/evaluator.py:149-181 (collision_ave / _max =
// mean / max of the 1556 per-vertex depths x 1000, over samples whose hand_type is 'interacting').
#pragma once
#include "ihmr_common.h"

// grid = B, block = 64: lane j < 42 owns joint j.  out (B,6) doubles:
//   [0] sum of per-joint errors, [1] number of them, [2] sum of aligned ("inter") errors, [3] number of them,
//   [4] mean penetration depth [mm], [5] max penetration depth [mm]  ([4],[5] = 0 and not counted unless interacting)
__global__ __launch_bounds__(64) void eval_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ coll, const float* __restrict__ scale,
                                                          const unsigned char* __restrict__ interacting, int B,
                                                          double* __restrict__ out) {
    const int b = blockIdx.x, j = threadIdx.x;
    const bool act = j < 42;
    float a[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f}, w = 0.f;
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[k] = pred[((size_t)b * 42 + j) * 3 + k]; g[k] = gt[((size_t)b * 42 + j) * 4 + k]; }
        w = gt[((size_t)b * 42 + j) * 4 + 3];
    }
    const float sc = scale ? scale[b] : 1.0f;
    const float p0[3] = {a[0], a[1], a[2]}, g0[3] = {g[0], g[1], g[2]};   // un-shifted copies for the aligned error

    // ---- per-hand MPJPE (metric_utils.py:23-38): root 0 for joints 0..20, then (on the already shifted copies) root 21
    float err = 0.f, cnt = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int root = 21 * h;
        const float wr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), root));
        if (wr > 0.f) {     // uniform over the wave
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a[k] -= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[k]), root));
                g[k] -= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[k]), root));
            }
            if (act && j >= root && j < root + 21 && w > 0.f) {
                const float dx = a[0] - g[0], dy = a[1] - g[1], dz = a[2] - g[2];
                err += sqrtf(dx * dx + dy * dy + dz * dz) / sc;
                cnt += 1.f;
            }
        }
    }
    const double err_sum = (double)wave_reduce_sum(err), err_n = (double)wave_reduce_sum(cnt);

    // ---- aligned error over all valid joints (metric_utils.py:107-143, no rotation): p' = (p - mean p) / std p * std g + mean g
    const bool valid = act && w > 0.f;
    const float n = wave_reduce_sum(valid ? 1.f : 0.f);
    float ierr = 0.f;
    if (n >= 2.f) {
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mp = wave_reduce_sum(valid ? p0[k] : 0.f) / n, mg = wave_reduce_sum(valid ? g0[k] : 0.f) / n;
            const float dp = p0[k] - mp, dg = g0[k] - mg;
            const float sp = sqrtf(wave_reduce_sum(valid ? dp * dp : 0.f) / n), sg = sqrtf(wave_reduce_sum(valid ? dg * dg : 0.f) / n);
            d[k] = (dp / sp * sg + mg) - g0[k];
        }
        if (valid) ierr = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / sc;
    }
    const double ierr_sum = (double)wave_reduce_sum(ierr);

    // ---- penetration depth statistics of the 1556 per-vertex values [mm]
    double csum = 0.0;
    float cmax = -INFINITY;
    const bool inter = interacting ? interacting[b] != 0 : true;
    if (inter) {
        for (int v = j; v < 2 * NV; v += 64) {
            const float x = coll[(size_t)b * 2 * NV + v];
            csum += (double)x;
            cmax = fmaxf(cmax, x);
        }
    }
    double cs = csum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
    const float cm = wave_reduce_max(cmax);
    if (j == 0) {
        double* o = out + (size_t)b * 6;
        o[0] = err_sum; o[1] = err_n;
        o[2] = n >= 2.f ? ierr_sum : 0.0; o[3] = n >= 2.f ? (double)n : 0.0;
        o[4] = inter ? cs / (double)(2 * NV) * 1000.0 : 0.0;
        o[5] = inter ? (double)cm * 1000.0 : 0.0;
    }
}

// MPVPE partial sums (BASELINE.json's metric list; the reference exports the meshes -- baseline_model.py:365-368,
// mlp_model.py:708-711 -- and has no vertex metric of its own).  Same convention as the MPJPE above: per hand,
// root-relative, L2 per point, / scale; the root of a mesh is its wrist regressed with row 0 of the MANO joint
// regressor (root_w (2,778): right, left).  A hand counts when mano_params_weight[b][h] > 0 (a GT mesh exists).
// grid = (B, 2), block = 256.  out (B,2,2) doubles: per hand [sum of the per-vertex errors, number of them].
__global__ __launch_bounds__(256) void eval_mpvpe_kernel(const float* __restrict__ pred_r, const float* __restrict__ pred_l,
                                                         const float* __restrict__ gt_r, const float* __restrict__ gt_l,
                                                         const float* __restrict__ root_w, const float* __restrict__ params_weight,
                                                         const float* __restrict__ scale, int B, double* __restrict__ hand_out) {
    __shared__ float part[4][8];
    __shared__ float root[6];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    double* o = hand_out + ((size_t)b * 2 + h) * 2;
    if (!(params_weight[b * 2 + h] > 0.f)) {
        if (tid == 0) { o[0] = 0.0; o[1] = 0.0; }
        return;
    }
    const float* P = (h ? pred_l : pred_r) + (size_t)b * NV3;
    const float* G = (h ? gt_l : gt_r) + (size_t)b * NV3;
    const float* W = root_w + (size_t)h * NV;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int v = tid; v < NV; v += 256) {
        const float w = W[v];
#pragma unroll
        for (int k = 0; k < 3; ++k) { acc[k] += w * P[3 * v + k]; acc[3 + k] += w * G[3 * v + k]; }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float s = wave_reduce_sum(acc[k]);
        if (lane == 0) part[wave][k] = s;
    }
    __syncthreads();
    if (tid < 6) root[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    const float sc = scale ? scale[b] : 1.0f;
    float err = 0.f;
    for (int v = tid; v < NV; v += 256) {
        const float dx = (P[3 * v] - root[0]) - (G[3 * v] - root[3]);
        const float dy = (P[3 * v + 1] - root[1]) - (G[3 * v + 1] - root[4]);
        const float dz = (P[3 * v + 2] - root[2]) - (G[3 * v + 2] - root[5]);
        err += sqrtf(dx * dx + dy * dy + dz * dz) / sc;
    }
    const float s = wave_reduce_sum(err);
    __syncthreads();
    if (lane == 0) part[wave][0] = s;
    __syncthreads();
    if (tid == 0) { o[0] = ((double)part[0][0] + (double)part[1][0]) + ((double)part[2][0] + (double)part[3][0]); o[1] = (double)NV; }
}
