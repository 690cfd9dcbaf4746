// IHMR-MLP training step: gradient of the reference's training objective w.r.t. the packed
// prediction vector, and the small dense pieces of a Linear-ReLU head's backward pass + Adam.
//
// Reference: models/mlp_model.py:514-583 (compute_loss with a stage's weights), :586-589 (optimize_parameters),
// models/loss_utils.py:46-78,114-135 (_mano_pose_loss with the reference's own batch_rodrigues,
// models/transform_utils.py:23-44; _mano_shape_loss; _hand_trans_loss; _shape_reg_loss; _shape_residual_loss),
// torch.optim.Adam as created at mlp_model.py:403-405.
//
// The mesh-dependent terms (2-D / 3-D joints vs the annotation, collision) run through the fused forward + LBS
// backward that IHMR-OPT uses (the `init_*` target pointers of ihmr_opt_io point at the annotation); this file adds
// the terms that act on the parameters directly and gathers everything into d loss / d final_params (B,122).
#pragma once
#include "refine.h"

// the reference's batch_rodrigues for ONE axis-angle vector: R = cos I + (1 - cos) r r^T + sin [r]x,
// angle = ||theta + 1e-8||, r = theta / angle (transform_utils.py:23-44).  If dk >= 0 also d R / d theta[dk].
__device__ __forceinline__ void ref_rodrigues(const float* th, float* R, int dk, float* dR) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float a = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float r[3] = {th[0] / a, th[1] / a, th[2] / a};
    const float c = cosf(a), s = sinf(a), oc = 1.0f - c;
    const float K[9] = {0.f, -r[2], r[1], r[2], 0.f, -r[0], -r[1], r[0], 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (i == j ? c : 0.f) + oc * r[i] * r[j] + s * K[3 * i + j];
    if (dk < 0) return;
    const float e = (dk == 0 ? t0 : (dk == 1 ? t1 : t2)) / a;                 // d angle / d theta_k
    float dr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dr[i] = ((i == dk ? 1.0f : 0.0f) - r[i] * e) / a;
    const float dK[9] = {0.f, -dr[2], dr[1], dr[2], 0.f, -dr[0], -dr[1], dr[0], 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            dR[3 * i + j] = (i == j ? -s * e : 0.f) + s * e * r[i] * r[j] + oc * (dr[i] * r[j] + r[i] * dr[j]) + c * e * K[3 * i + j] +
                            s * dK[3 * i + j];
}

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// grid = B, block = 128: thread e < 122 owns entry e of final_params = [cam 3 | R orient 3 | R pose 45 | L orient 3 |
// L pose 45 | R shape 10 | L shape 10 | trans 3] (mlp_model.py:426-439).
// grad (B,122) = d loss / d final_params;  terms (B,5) = per-sample shares of [mano_pose, mano_shape, hand_trans,
// shape_reg, shape_residual] (already weighted; their sum over the batch is the reference's scalar).
__global__ __launch_bounds__(128) void mlp_train_grad_kernel(ihmr_opt_io io, OptWork wk, int B, ihmr_train_weights tw,
                                                             const float* __restrict__ gt_pose, const float* __restrict__ gt_shape,
                                                             const float* __restrict__ params_weight,
                                                             const float* __restrict__ init_shape,
                                                             const float* __restrict__ trans_weight_mean, float* __restrict__ grad,
                                                             float* __restrict__ terms, const int32_t* __restrict__ out_cols, int n_out,
                                                             float* __restrict__ d_out, int ld_out) {
    __shared__ float part[5][128];
    __shared__ float gsh[128];
    const int b = blockIdx.x, e = threadIdx.x;
    const float Bn = (float)(io.norm_batch > 0 ? io.norm_batch : B);
    float g = 0.f, l_pose = 0.f, l_shape = 0.f, l_trans = 0.f, l_reg = 0.f, l_res = 0.f;
    if (e < 3) {
        // camera: 2-D loss mean(|t - p| w) * W, p = (X + cam[1:3]) * cam[0] on the un-aligned joints (transform_utils.py:47-54)
        const float cs = io.cam[b * 3], ctx = io.cam[b * 3 + 1], cty = io.cam[b * 3 + 2];
        const float s2 = tw.joints_2d / (Bn * 84.0f);
        float acc = 0.f;
        for (int j = 0; j < 42; ++j) {
            const float X = wk.joints_raw[(b * 42 + j) * 3], Y = wk.joints_raw[(b * 42 + j) * 3 + 1];
            const float tx = io.init_joints_2d[(b * 42 + j) * 3], ty = io.init_joints_2d[(b * 42 + j) * 3 + 1],
                        w = io.init_joints_2d[(b * 42 + j) * 3 + 2];
            const float gx = -s2 * sgnf(tx - (X + ctx) * cs) * w, gy = -s2 * sgnf(ty - (Y + cty) * cs) * w;
            acc += e == 0 ? gx * (X + ctx) + gy * (Y + cty) : (e == 1 ? gx * cs : gy * cs);
        }
        g = acc;
    } else if (e < 99) {
        const int hnd = e >= 51, q = e - (hnd ? 51 : 3);               // q in [0,48): 0..2 orient, 3..47 finger pose
        if (q < 3) {
            g = wk.g_orient[((size_t)hnd * B + b) * 3 + q];
        } else {
            const int d = q - 3, jn = d / 3, k = d % 3;
            g = wk.g_pose[((size_t)hnd * B + b) * 45 + d];
            // _mano_pose_loss on the 15 finger joints: mean((R_gt - R_pred)^2 * w_hand) over (B, 135)
            const float w = params_weight[b * 2 + hnd];
            const float* tp = io.pose + ((size_t)hnd * B + b) * 45 + 3 * jn;
            const float* tg = gt_pose + (size_t)b * 96 + 48 * hnd + 3 + 3 * jn;
            const float thp[3] = {tp[0], tp[1], tp[2]}, thg[3] = {tg[0], tg[1], tg[2]};
            float Rp[9], Rg[9], dR[9];
            ref_rodrigues(thp, Rp, k, dR);
            ref_rodrigues(thg, Rg, -1, nullptr);
            const float sp = tw.mano_pose / (Bn * 135.0f);
            float acc = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i) { const float df = Rg[i] - Rp[i]; acc += df * dR[i]; sq += df * df; }
            g += -2.0f * sp * w * acc;
            if (k == 0) l_pose = sp * w * sq;
        }
    } else if (e < 119) {
        const int hnd = e >= 109, d = e - (hnd ? 109 : 99);
        const float x = io.shape[((size_t)hnd * B + b) * 10 + d];
        g = wk.g_shape[((size_t)hnd * B + b) * 10 + d];
        // _shape_reg_loss: mean((beta_r - beta_l)^2) over (B, 10)
        const float diff = io.shape[(size_t)b * 10 + d] - io.shape[((size_t)B + b) * 10 + d];
        const float sr = tw.shape_reg / (Bn * 10.0f);
        g += (hnd == 0 ? 2.0f : -2.0f) * sr * diff;
        if (hnd == 0) l_reg = sr * diff * diff;
        // _mano_shape_loss: mean(|gt - pred| * w_hand) over (B, 10), per hand
        const float w = params_weight[b * 2 + hnd], dg = gt_shape[(size_t)b * 20 + 10 * hnd + d] - x;
        const float ss = tw.mano_shape / (Bn * 10.0f);
        g += -ss * w * sgnf(dg);
        l_shape = ss * w * fabsf(dg);
        // _shape_residual_loss: mean(|pred - init|) over (B, 10), per hand
        const float di = x - init_shape[(size_t)b * 20 + 10 * hnd + d];
        const float sd = tw.shape_residual / (Bn * 10.0f);
        g += sd * sgnf(di);
        l_res = sd * fabsf(di);
    } else if (e < 122) {
        const int d = e - 119;
        g = wk.g_trans[b * 3 + d] + wk.g_trans_direct[b * 3 + d];
        // _hand_trans_loss with a (B,3) difference and a (B,1,1) weight: the reference's broadcast makes it
        // mean_i(w_i) * mean_{j,k}(d_jk^2) (mlp_model.py:557-558, loss_utils.py:114-118)
        const float df = io.gt_hand_trans[b * 4 + d] - io.trans[b * 3 + d];
        float wmean;
        if (trans_weight_mean) {
            wmean = trans_weight_mean[0];
        } else {                                            // mean of the batch this sample belongs to, rows in order
            const int nb = io.norm_batch > 0 ? io.norm_batch : B, b0 = (b / nb) * nb;
            float sw = 0.f;
            for (int q = 0; q < nb; ++q) sw += io.gt_hand_trans[(b0 + q) * 4 + 3];
            wmean = sw / (float)nb;
        }
        const float st = tw.hand_trans * wmean / (Bn * 3.0f);
        g += -2.0f * st * df;
        l_trans = st * df * df;
    }
    if (e < 122) grad[(size_t)b * 122 + e] = g;
    gsh[e] = g;
    part[0][e] = l_pose; part[1][e] = l_shape; part[2][e] = l_trans; part[3][e] = l_reg; part[4][e] = l_res;
    __syncthreads();
    if (e < 5) {
        float s = 0.f;
        for (int q = 0; q < 122; ++q) s += part[e][q];
        terms[(size_t)b * 5 + e] = s;
    }
    // optional: the columns the current stage's sub-network produces, straight into the head's dY operand
    if (d_out)
        for (int c = e; c < n_out; c += 128) d_out[(size_t)b * ld_out + c] = gsh[out_cols[c]];
}

// ------------------------------------------------------------------------------------- dense helpers of the head
// y[c][r] = x[r][c]   (rows x cols, row strides ldx / ldy); 32 x 32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols,
                                                        int ldx, int ldy) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = x[(size_t)(r0 + i) * ldx + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) y[(size_t)(c0 + i) * ldy + r0 + tx] = tile[tx][i];
}

// ReLU backward in place: dx[r][c] = y[r][c] > 0 ? dx[r][c] : 0
__global__ __launch_bounds__(256) void relu_backward_kernel(float* __restrict__ dx, const float* __restrict__ y, int rows, int cols,
                                                            int ld_dx, int ld_y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols;
    if (!(y[(size_t)r * ld_y + c] > 0.f)) dx[(size_t)r * ld_dx + c] = 0.f;
}

// the same for dense matrices (ld == cols) whose element count is a multiple of 4: one 16-byte load of y and of dx and one store per thread
// (the encoder's block outputs: up to 51 M elements per call)
__global__ __launch_bounds__(256) void relu_backward4_kernel(float4* __restrict__ dx, const float4* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 yv = y[i];
    float4 d = dx[i];
    if (!(yv.x > 0.f)) d.x = 0.f;
    if (!(yv.y > 0.f)) d.y = 0.f;
    if (!(yv.z > 0.f)) d.z = 0.f;
    if (!(yv.w > 0.f)) d.w = 0.f;
    dx[i] = d;
}

// out[c] = sum_r x[r][c] (bias gradient): a workgroup owns 64 columns, its 4 waves take rows w, w + 4, ... (coalesced over the
// columns), the four partial sums are added in wave order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols, int ldx) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64, c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < cols)
        for (int r = wave; r < rows; r += 4) s += x[(size_t)r * ldx + c];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) out[c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// torch.optim.Adam (no weight decay, no amsgrad) on a flat buffer; the same update expression as opt_adam_apply
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n, float grad_scale, float beta1, float beta2,
                                                        float eps, float step_size, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * grad_scale;
    float mm = m[i], vv = v[i];
    mm = mm + (1.0f - beta1) * (gr - mm);
    vv = vv * beta2;
    vv = vv + ((1.0f - beta2) * gr) * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] + (-step_size) * (mm / denom);
}
