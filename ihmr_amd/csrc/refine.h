// IHMR-OPT refinement step pieces: joint / translation / finger losses with their analytic
// gradients, the Adam update (+ snapshot), and the per-stage snapshot filter / argmin select.
//
// Reference: models/optimize_model.py:276-330 (__compute_loss), models/loss_utils.py:82-171,
// models/transform_utils.py:47-54, torch.optim.Adam as created at optimize_model.py:344,
// utils/opt_utils.py:70-153 (gather / filter / select).
#pragma once
#include "ihmr_common.h"

#define LOSS_THREADS 64
#define OPT_NPARAM IHMR_OPT_NPARAM  // parameter slots of a sample, see ihmr_hip.h (block order of IHMR_PB_*)

struct OptWork {  // carved from ihmr_opt_io.workspace
    LbsWork lbs;        // skeleton records, v_posed, bwd scratch for 2B hands
    float* joints_raw;  // (B,42,3)
    float* g_verts;     // (2,B,778,3)
    float* g_joints;    // (B,42,3)
    float* g_trans_direct;  // (B,3)
    float* g_cam;       // (B,3)  camera gradient of the 2-D term (only written when the stage refines the camera)
    float* g_orient;    // (2,B,3)
    float* g_pose;      // (2,B,45)
    float* g_shape;     // (2,B,10)
    float* g_trans;     // (B,3)
    void* sdf_ws;
};

static inline size_t opt_ws_bytes(int B) {
    size_t n = 0;
    n += (size_t)2 * B * NV3 * 4;         // g_verts
    n += lbs_ws_bytes(2 * B);
    n += (size_t)B * 42 * 3 * 4 * 2;      // joints_raw, g_joints
    n += (size_t)B * (3 + 6 + 90 + 20 + 3 + 3 + 1) * 4;
    n = (n + 255) & ~(size_t)255;
    n += 8192;
    return n + sdf_ws_bytes(2 * B, true);
}

static inline OptWork opt_carve(void* ws, int B) {
    OptWork w;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    w.lbs = lbs_carve(take(lbs_ws_bytes(2 * B)), 2 * B);
    w.g_verts = (float*)take((size_t)2 * B * NV3 * 4);
    w.joints_raw = (float*)take((size_t)B * 42 * 3 * 4);
    w.g_joints = (float*)take((size_t)B * 42 * 3 * 4);
    w.g_trans_direct = (float*)take((size_t)B * 3 * 4);
    w.g_orient = (float*)take((size_t)B * 6 * 4);
    w.g_pose = (float*)take((size_t)B * 90 * 4);
    w.g_shape = (float*)take((size_t)B * 20 * 4);
    w.g_trans = (float*)take((size_t)B * 3 * 4);
    w.g_cam = (float*)take((size_t)B * 3 * 4);
    w.sdf_ws = (void*)p;
    return w;
}

// finger table of loss_utils.py:139-145: [a, b, c, tip] for index, middle, little, ring, thumb
__constant__ int c_finger_ids[20] = {1, 2, 3, 17, 4, 5, 6, 18, 7, 8, 9, 20, 10, 11, 12, 19, 13, 14, 15, 16};

// ONE wave per sample: lane j < 42 owns joint j (the steps are separated by wave-level barriers only, so the
// wave can run beside the collision sampling of the same workgroup, see opt_sample_loss_kernel).
struct LossShared {
    float raw[42][3], p1[42][3], p2[42][3], g2[42][3], acc[8][LOSS_THREADS], gsum[2][3];
};
// LDS hand-off between the lanes of one wave: DS operations of a wave execute in order, the fences only stop the
// compiler from moving them
#define LOSS_SYNC()                                                  \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       \
    } while (0)

// gj_lds (fused tail): the joint gradients go straight to the LBS backward's LDS records -- bw[hand].gj, raw hand frame: x of the left
// hand negated, what lbs_bwd1_hand's staging does with the values it reads back from global memory -- instead of wk.g_joints
__device__ __forceinline__ void opt_loss_wave(const ihmr_opt_io& io, const OptWork& wk, int B, const ihmr_opt_weights& w,
                                              LossShared& sh, int b, int j, int need_cam, LbsBwdShared* gj_lds = nullptr) {
    const bool act = j < 42;
    const int Bn = io.norm_batch > 0 ? io.norm_batch : B;   // the batch the reference's means run over
    // ---- every global input of this sample first, in one batch (the stores below may alias them as far as the
    //      compiler knows, so left in place each load would wait for its own round trip)
    const int jj = act ? j : 0;
    const float cs = io.cam[b * 3], ctx = io.cam[b * 3 + 1], cty = io.cam[b * 3 + 2];
    float r[3], t2[3], tg2[3], tg3[4], ti3[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r[k] = wk.joints_raw[(b * 42 + jj) * 3 + k];
        t2[k] = io.init_joints_2d[(b * 42 + jj) * 3 + k];
        tg2[k] = io.gt_joints_2d[(b * 42 + jj) * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { tg3[k] = io.gt_joints_3d[(b * 42 + jj) * 4 + k]; ti3[k] = io.init_joints_3d[(b * 42 + jj) * 4 + k]; }
    float tp4[4] = {0.f, 0.f, 0.f, 0.f}, tgt4[4] = {0.f, 0.f, 0.f, 0.f}, tr3[3] = {0.f, 0.f, 0.f};
    if (j == 5) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { tp4[k] = io.init_hand_trans_j[b * 4 + k]; tgt4[k] = io.gt_hand_trans[b * 4 + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) tr3[k] = io.trans[b * 3 + k];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sh.raw[j][k] = r[k];
    }
    LOSS_SYNC();

    // ---- 2D: projection of the un-aligned joints (transform_utils.py:47-54, optimize_model.py:263)
    float l2d_p = 0.f, l2d_gt = 0.f, g_raw[3] = {0.f, 0.f, 0.f};
    if (act) {
        const float px = (r[0] + ctx) * cs, py = (r[1] + cty) * cs;
        io.joints_2d[(b * 42 + j) * 2] = px;
        io.joints_2d[(b * 42 + j) * 2 + 1] = py;
        const float dx = t2[0] - px, dy = t2[1] - py;
        l2d_p = (fabsf(dx) + fabsf(dy)) * t2[2];
        l2d_gt = (fabsf(tg2[0] - px) + fabsf(tg2[1] - py)) * tg2[2];
        const float s2 = w.joints_2d / (float)(Bn * 42 * 2);
        // d|t - p|/dp = -sign(t - p); dp/dX = cam scale
        const float sx = dx > 0.f ? -1.f : (dx < 0.f ? 1.f : 0.f), sy = dy > 0.f ? -1.f : (dy < 0.f ? 1.f : 0.f);
        g_raw[0] = s2 * sx * t2[2] * cs;
        g_raw[1] = s2 * sy * t2[2] * cs;
        if (need_cam) {   // p = (X + cam[1:3]) * cam[0]: d/d scale, d/d tx, d/d ty of this joint's term
            const float gx = s2 * sx * t2[2], gy = s2 * sy * t2[2];
            sh.acc[5][j] = gx * (r[0] + ctx) + gy * (r[1] + cty);
            sh.acc[6][j] = gx * cs;
            sh.acc[7][j] = gy * cs;
        }
    }

    // ---- 3D: two successive in-place root alignments (GT weights first, then init weights); lane 0 holds joint 0,
    //      whose weight picks the root; the root joint's target row comes from its lane (uniform index)
    const int root1 = align_root(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(tg3[3]), 0)));
    const int root2 = align_root(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ti3[3]), 0)));
    float tg0[3], ti0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tg0[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tg3[k]), root1 >= 0 ? root1 : 0));
        ti0[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ti3[k]), root2 >= 0 ? root2 : 0));
    }
    float a1[3], a2[3], l3d_gt = 0.f, l3d_p = 0.f;
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a1[k] = root1 >= 0 ? r[k] - sh.raw[root1][k] : r[k]; sh.p1[j][k] = a1[k]; }
    }
    LOSS_SYNC();
    if (act) {
        const float s3 = w.joints_3d / (float)(Bn * 42 * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a2[k] = root2 >= 0 ? a1[k] - sh.p1[root2][k] : a1[k];
            sh.p2[j][k] = a2[k];
            const float gg = (root1 >= 0 ? tg3[k] - tg0[k] : tg3[k]) - a1[k];
            l3d_gt += gg * gg * tg3[3];
            const float gi = (root2 >= 0 ? ti3[k] - ti0[k] : ti3[k]) - a2[k];
            l3d_p += gi * gi * ti3[3];
            sh.g2[j][k] = -2.0f * s3 * gi * ti3[3];
            io.joints_3d[(b * 42 + j) * 3 + k] = a2[k];
        }
    }
    LOSS_SYNC();

    // ---- finger regulariser on the aligned joints (loss_utils.py:138-171); thread f < 10 owns a finger
    float lfin = 0.f;
    if (j < 10) {
        const int off = j < 5 ? 0 : 21, fi = j % 5;
        const int ia = c_finger_ids[4 * fi] + off, ib = c_finger_ids[4 * fi + 1] + off, ic = c_finger_ids[4 * fi + 2] + off,
                  it = c_finger_ids[4 * fi + 3] + off;
        float f0[3], f1[3], f2[3], n1[3], n2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { f0[k] = sh.p2[ia][k] - sh.p2[ib][k]; f1[k] = sh.p2[ib][k] - sh.p2[ic][k]; f2[k] = sh.p2[ic][k] - sh.p2[it][k]; }
        cross3(f0, f1, n1);
        cross3(f1, f2, n2);
        const float C1 = f2[0] * n1[0] + f2[1] * n1[1] + f2[2] * n1[2];
        const float C2 = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
        lfin = fabsf(C1) - fminf(0.f, C2);
        if (w.finger_reg != 0.f) {
            const float sf = w.finger_reg / (float)Bn;
            const float k1 = sf * (C1 > 0.f ? 1.f : (C1 < 0.f ? -1.f : 0.f));
            const float k2 = C2 < 0.f ? -sf : 0.f;
            // dC1/df0 = f1 x f2 = n2, dC1/df1 = f2 x f0, dC1/df2 = n1
            // dC2/df0 = f1 x n2, dC2/df1 = n2 x f0 + f2 x n1, dC2/df2 = n1 x f1
            float t1[3], t2[3], t3[3], t4[3], t5[3];
            cross3(f2, f0, t1);
            cross3(f1, n2, t2);
            cross3(n2, f0, t3);
            cross3(f2, n1, t4);
            cross3(n1, f1, t5);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d0 = k1 * n2[k] + k2 * t2[k];
                const float d1 = k1 * t1[k] + k2 * (t3[k] + t4[k]);
                const float d2 = k1 * n1[k] + k2 * t5[k];
                // every joint belongs to exactly one finger, so these writes never collide
                sh.g2[ia][k] += d0;
                sh.g2[ib][k] += d1 - d0;
                sh.g2[ic][k] += d2 - d1;
                sh.g2[it][k] += -d2;
            }
        }
    }
    LOSS_SYNC();

    // ---- back through the two alignments: g_in = g_out - [j == root] * sum_j g_out
    if (j < 3) {
        float s = 0.f;
        for (int q = 0; q < 42; ++q) s += sh.g2[q][j];
        sh.gsum[0][j] = s;
    }
    LOSS_SYNC();
    float g1[3];
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g1[k] = sh.g2[j][k] - ((root2 >= 0 && j == root2) ? sh.gsum[0][k] : 0.f);
            sh.p1[j][k] = g1[k];  // reuse as scratch for the second sum
        }
    }
    LOSS_SYNC();
    if (j < 3) {
        float s = 0.f;
        for (int q = 0; q < 42; ++q) s += sh.p1[q][j];
        sh.gsum[1][j] = s;
    }
    LOSS_SYNC();
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float g0 = g1[k] - ((root1 >= 0 && j == root1) ? sh.gsum[1][k] : 0.f);
            const float gv = g0 + g_raw[k];
            if (gj_lds) gj_lds[j / 21].gj[j % 21][k] = (j >= 21 && k == 0) ? -gv : gv;
            else wk.g_joints[(b * 42 + j) * 3 + k] = gv;
        }
    }

    // ---- per-sample reductions (fixed order: thread 0 sums 42 entries)
    sh.acc[0][j] = l2d_p; sh.acc[1][j] = l3d_p; sh.acc[2][j] = lfin; sh.acc[3][j] = l2d_gt; sh.acc[4][j] = l3d_gt;
    LOSS_SYNC();
    if (j < 5) {
        float s = 0.f;
        const int n = j == 2 ? 10 : 42;
        for (int q = 0; q < n; ++q) s += sh.acc[j][q];
        if (j == 0) io.loss_batch[0 * B + b] = s / 84.f * w.joints_2d;
        if (j == 1) io.loss_batch[1 * B + b] = s / 126.f * w.joints_3d;
        if (j == 2) io.loss_batch[3 * B + b] = s;
        if (j == 3) io.loss_batch[4 * B + b] = s / 84.f;
        if (j == 4) io.loss_batch[5 * B + b] = s / 126.f;
    } else if (need_cam && j < 8) {
        float s = 0.f;
        for (int q = 0; q < 42; ++q) s += sh.acc[j][q];
        wk.g_cam[b * 3 + (j - 5)] = s;
    }
    // ---- translation loss (loss_utils.py:114-118) and the collision gradient scale
    if (j == 5) {
        float lp = 0.f, lg = 0.f;
        const float st = w.trans / (float)(Bn * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = tp4[k] - tr3[k];
            lp += d * d * tp4[3];
            wk.g_trans_direct[b * 3 + k] = -2.0f * st * d * tp4[3];
            const float dg = tgt4[k] - tr3[k];
            lg += dg * dg * tgt4[3];
        }
        io.loss_batch[6 * B + b] = lp / 3.f;
        io.loss_batch[7 * B + b] = lg / 3.f;
    }
}

// Collision sampling + joint / translation / finger losses of sample b in one launch.  grid = B, block = 512:
// waves 0-6 sample the two distance grids at the other hand's vertices (value, gradient -> g_verts), wave 7
// evaluates the joint losses and their gradient (-> g_joints); the two halves share nothing but the launch.
#define OPT_SAMPLE_WORKERS (SDF_SAMPLE_THREADS - WAVE)
__global__ __launch_bounds__(SDF_SAMPLE_THREADS) void opt_sample_loss_kernel(ihmr_opt_io io, OptWork wk, int B, ihmr_opt_weights w,
                                                                             VertLayout vl, SdfWorkspace ws, int need_cam, MlpSelect sel) {
    TL_SCOPE(10);
    __shared__ LossShared sh;
    __shared__ float red16[SDF_SAMPLE_THREADS / WAVE];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid >= OPT_SAMPLE_WORKERS) opt_loss_wave(io, wk, B, w, sh, b, tid - OPT_SAMPLE_WORKERS, need_cam);
    // collision gradient scale: weight * [two-hand sample] / (loss divisor [num_hands^2] * B)   (loss_utils.py:186-188)
    const float mask = (io.hand_type_array[b * 2] + io.hand_type_array[b * 2 + 1]) > 1.5f ? 1.f : 0.f;
    const float gs = w.collision * mask / (ws.loss_div * (float)(io.norm_batch > 0 ? io.norm_batch : B));
    // (an IHMR-MLP evaluation -- a keep / reject decision follows, no backward -- takes the values from the prep kernel's cell words:
    // the same bits, no gradient; round 6)
    if (sel.mode)
        sdf_sample_cells(vl, ws, io.loss_batch + 2 * B, io.coll_per_vert, io.coll_origin_scale, B, io.hand_type_array, red16, b, OPT_SAMPLE_WORKERS);
    else
        sdf_sample_block(vl, ws, 0.f, io.loss_batch + 2 * B, io.coll_per_vert, io.coll_origin_scale, nullptr, wk.g_verts, B, gs,
                         io.hand_type_array, red16, b, OPT_SAMPLE_WORKERS);
    // IHMR-MLP: keep / reject the stage's update of this sample and store it to the "prev" tables (mlp_infer.h)
    if (sel.mode) mlp_select_sample(sel, io.loss_batch, B, b);
}

// One optimizer step of parameter slot e (< 122) of sample b: adds the direct (non-MANO) gradient terms, takes the
// snapshot (before the step, as optimize_model.py:401-403) and applies torch.optim.Adam's single-tensor update
// (or torch.optim.SGD with momentum 0.9, optimize_model.py:345-347).  Slots follow the IHMR_PB_* block order:
// [cam 3 | trans 3 | R orient 3 | L orient 3 | R pose 45 | L pose 45 | R shape 10 | L shape 10]; only the slots of the
// blocks in `mask` (the stage's update_params) move, are snapshotted and own optimizer state.
struct ParamStep {
    int mask;             // IHMR_PB_* bits, 0: no update
    float w_shape_reg, step_size, bc2_sqrt;   // Adam: lr / (1 - beta1^t), sqrt(1 - beta2^t); SGD: step_size = lr
    int snap_idx;         // >= 0: snapshot slot of this iteration
    int reset_state;      // first iteration of a stage: zero the optimizer state instead of stepping
    int sgd;              // 0 = Adam, 1 = SGD(momentum 0.9)
};
__device__ __forceinline__ int opt_slot_block(int e) {
    return e < 3 ? 0 : (e < 6 ? 1 : (e < 9 ? 2 : (e < 12 ? 3 : (e < 57 ? 4 : (e < 102 ? 5 : (e < 112 ? 6 : 7))))));
}
// pointer to slot e of sample b in the caller's parameter buffers
__device__ __forceinline__ float* opt_slot_ptr(const ihmr_opt_io& io, int B, int b, int e, int blk) {
    switch (blk) {
        case 0: return io.cam + b * 3 + e;
        case 1: return io.trans + b * 3 + (e - 3);
        case 2: case 3: return io.orient + ((size_t)(blk - 2) * B + b) * 3 + (e - 6 - 3 * (blk - 2));
        case 4: case 5: return io.pose + ((size_t)(blk - 4) * B + b) * 45 + (e - 12 - 45 * (blk - 4));
        default: return io.shape + ((size_t)(blk - 6) * B + b) * 10 + (e - 102 - 10 * (blk - 6));
    }
}
__device__ __forceinline__ void opt_param_apply(const ihmr_opt_io& io, const OptWork& wk, int B, const ParamStep& st, int b, int e) {
    const int blk = opt_slot_block(e);
    if (!((st.mask >> blk) & 1)) return;
    float* p = opt_slot_ptr(io, B, b, e, blk);
    float g;
    if (blk == 0) {
        g = wk.g_cam[b * 3 + e];
    } else if (blk == 1) {
        g = wk.g_trans[b * 3 + (e - 3)] + wk.g_trans_direct[b * 3 + (e - 3)];
    } else if (blk < 4) {
        g = wk.g_orient[((size_t)(blk - 2) * B + b) * 3 + (e - 6 - 3 * (blk - 2))];
    } else if (blk < 6) {
        g = wk.g_pose[((size_t)(blk - 4) * B + b) * 45 + (e - 12 - 45 * (blk - 4))];
    } else {
        const int hnd = blk - 6, d = e - 102 - 10 * hnd;
        g = wk.g_shape[((size_t)hnd * B + b) * 10 + d];
        // shape regulariser mean((beta_r - beta_l)^2) (loss_utils.py:121-128)
        const float diff = io.shape[(size_t)b * 10 + d] - io.shape[((size_t)B + b) * 10 + d];
        const float gr = 2.0f * st.w_shape_reg / (float)((io.norm_batch > 0 ? io.norm_batch : B) * 10) * diff;
        g += hnd == 0 ? gr : -gr;
    }
    const float x = *p;
    if (st.snap_idx >= 0) io.snap_params[((size_t)st.snap_idx * B + b) * OPT_NPARAM + e] = x;
    float m = io.adam_m[b * OPT_NPARAM + e];
    if (st.sgd) {
        *p = opt_sgd_update(x, g, m, st.step_size);
        io.adam_m[b * OPT_NPARAM + e] = m;
        return;
    }
    float v = io.adam_v[b * OPT_NPARAM + e];
    const float xn = opt_adam_update(x, g, m, v, st.step_size, st.bc2_sqrt);       // (ihmr_pure.h: torch's operation order)
    io.adam_m[b * OPT_NPARAM + e] = m;
    io.adam_v[b * OPT_NPARAM + e] = v;
    *p = xn;
}
// the three per-sample losses a stage may filter / select on (IHMR_LOSS_* = rows 0..2 of loss_batch), kept per snapshot
__device__ __forceinline__ void opt_snapshot_losses(const ihmr_opt_io& io, int B, const ParamStep& st, int b, int e) {
    if (st.snap_idx >= 0 && e < 3) io.snap_loss[((size_t)st.snap_idx * 3 + e) * B + b] = io.loss_batch[e * B + b];
}

// stand-alone step (the last iteration of a stage): grid = B, block = 128, thread e < 122.  The shape slots read
// both hands' values of a sample while updating them: they sit in one wave (slots 102..121), reads before writes.
__global__ __launch_bounds__(128) void opt_adam_kernel(ihmr_opt_io io, OptWork wk, int B, ParamStep st) {
    opt_snapshot_losses(io, B, st, blockIdx.x, threadIdx.x);
    if ((int)threadIdx.x < OPT_NPARAM) opt_param_apply(io, wk, B, st, blockIdx.x, threadIdx.x);
}

// Head of a refinement iteration: the Adam step that closes the PREVIOUS iteration (its gradients and losses are
// still in place), then both skeletons of sample b from the updated parameters.  grid = B, block = 2 x 192
// (threads [0,192) right hand, [192,384) left hand).  The whole sample lives in one workgroup because the left
// hand's wrist shift reads the right hand's shape and the shared translation (optimize_model.py:196-206).
__global__ __launch_bounds__(384) void opt_adam_skel_kernel(ihmr_mano m, ihmr_opt_io io, OptWork wk, int B, ParamStep st,
                                                            int* inside_count) {
    TL_SCOPE(7);
    __shared__ float sk[2][SK_STRIDE];
    const int b = blockIdx.x, tid = threadIdx.x;
    // the collision kernels of this iteration append to the inside-voxel counter: start it at zero
    if (b == 0 && tid >= 192 && tid < 192 + SDF_NZERO) sdf_zero_counter(inside_count, tid - 192);
    if (st.reset_state && tid < OPT_NPARAM) { io.adam_m[b * OPT_NPARAM + tid] = 0.f; io.adam_v[b * OPT_NPARAM + tid] = 0.f; }
    if (st.mask && tid < OPT_NPARAM) { opt_snapshot_losses(io, B, st, b, tid); opt_param_apply(io, wk, B, st, b, tid); }
    __syncthreads();   // the updated parameters are read back below by other threads of this workgroup
    const int hl = tid / 192;
    lbs_skel_hand<true>(m, io.orient, io.pose, io.shape, io.trans, B, wk.lbs.skel, wk.joints_raw, sk[hl], hl * B + b, tid % 192);
}

// Tail of a refinement iteration in ONE launch per sample: collision sampling + joint losses (= opt_sample_loss_kernel), the LBS
// backward of both hands of the sample (= lbs_bwd1_kernel, threads [0,256) right hand, [256,512) left hand), and -- STEP: stages that
// do not move the finger pose, whose LBS backward ends here (the finger-pose stage goes on with a batch-wide GEMM) -- the optimizer
// step that closes the iteration plus both skeletons of the next one (= opt_adam_skel_kernel).  Each phase consumes what the previous one of the SAME workgroup
// wrote (g_verts / g_joints, then the parameter gradients): two launch boundaries and their tails fewer per iteration, nothing else
// changes -- the phases are the same device functions, the results the same bits.  grid = B, block = 512, 2 workgroups per CU
// (~56 KB static + 2 x nseg x 48 B dynamic LDS).
// SKIN (with STEP, stages that keep v_posed -- neither finger pose nor shape moves): a fourth phase skins the stored v_posed of both
// hands with the skeletons just computed (= lbs_skin_kernel<true, REUSE>, the same operations in the same order: the same bits), so
// the next iteration starts at the collision kernels: 3 launches per iteration.
// (Round 6 tried a third form for the shape stage -- the fourth phase rebuilding v_posed = (v_template + shapedirs . beta) + P from the stored
// pose offsets, no skinning launch: bit-identical, and slower in both regimes, 52.6 us against 40.0 + 10.3 us per 448 samples and 73.8
// against 71.1 us per iteration at one batch of 64 -- a thread's four vertices are four dependent gathers of eleven basis rows, where the
// skinning launch streams each row once for 4 - 8 hands.  docs/experiments.md)
// Phase stamps (experiment builds only, -DTAIL_STAMPS; scripts/tail_stamps.py): shader-clock time of each phase of a sample's workgroup,
// summed per sample and kernel form
#ifdef TAIL_STAMPS
__device__ long long g_tail_stamps[3][4096][8];
#define TAIL_TK(k) do { if (threadIdx.x == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); if (blockIdx.x < 4096) g_tail_stamps[STEP + SKIN][blockIdx.x][k] += now_ - tk_prev_; tk_prev_ = now_; } } while (0)
#else
#define TAIL_TK(k)
#endif
// dynamic LDS of opt_tail_kernel: [2][nseg][12] floats of the LBS backward
static inline int opt_tail_dynamic_lds(int nseg) { return 2 * nseg * 12 * (int)sizeof(float); }
template <bool STEP, bool SKIN = false>
__global__ __launch_bounds__(SDF_SAMPLE_THREADS, 4) void opt_tail_kernel(ihmr_mano m, ihmr_opt_io io, OptWork wk, int B, ihmr_opt_weights w,
                                                                         VertLayout vl, SdfWorkspace ws, int need_cam, int need_mask,
                                                                         ParamStep st, int* inside_count) {
    TL_SCOPE(3 + (STEP ? 1 : 0) + (SKIN ? 1 : 0));
    __shared__ LossShared sh;
    __shared__ float red16[SDF_SAMPLE_THREADS / WAVE];
    __shared__ LbsBwdShared bw[2];
    extern __shared__ __attribute__((aligned(16))) float tail_part[];   // [2][nseg][12]
    const int b = blockIdx.x, tid = threadIdx.x;
#ifdef TAIL_STAMPS
    long long tk_prev_ = (long long)__builtin_readcyclecounter();
    if (tid == 0 && blockIdx.x < 4096) g_tail_stamps[STEP + SKIN][blockIdx.x][7] += 1;
#endif
    // ---- phase 0: global -> LDS by DMA (no registers held): what the LBS backward (phase 2) needs from the forward of this iteration,
    //      v_posed and the skeleton record of both hands; it lands while phase 1 runs.  (Rounds 4-5 also staged the two inside-voxel
    //      bitmaps here and published them with a workgroup barrier in the middle of phase 1: since round 6 the prep kernel puts the
    //      corner mask of every query into its cell word, and phase 1 has no barrier before the block sum.)
    const int hl = tid / LBS_THREADS;
    auto dma_backward_inputs = [&]() {
        if ((need_mask & 7) != 0) {
            lds_dma_dwords(wk.lbs.v_posed + (size_t)(hl * B + b) * NV3, bw[hl].vp, NV3, tid % LBS_THREADS, LBS_THREADS);
            lds_dma_dwords(wk.lbs.skel + (size_t)(hl * B + b) * SK_STRIDE, bw[hl].sk, SK_STRIDE, tid % LBS_THREADS, LBS_THREADS);
        } else if (SKIN) {       // translation stage: the skeleton records stay valid for the next iteration's vertices (phase 3 / 4)
            lds_dma_dwords(wk.lbs.skel + (size_t)(hl * B + b) * SK_STRIDE, bw[hl].sk, SK_STRIDE, tid % LBS_THREADS, LBS_THREADS);
        }
    };
    dma_backward_inputs();
    // ---- phase 1: collision sampling (waves 0-6) + joint / translation / finger losses (wave 7).  Their gradients -- d L / d vertices,
    //      d L / d joints -- are handed to phase 2 through its LDS records, not through global memory (the same values).  One workgroup
    //      barrier inside, at the same place for both kinds of wave: the block sum
    const float mask = (io.hand_type_array[b * 2] + io.hand_type_array[b * 2 + 1]) > 1.5f ? 1.f : 0.f;
    const float gs = w.collision * mask / (ws.loss_div * (float)(io.norm_batch > 0 ? io.norm_batch : B));
    if (tid >= OPT_SAMPLE_WORKERS) {
        opt_loss_wave(io, wk, B, w, sh, b, tid - OPT_SAMPLE_WORKERS, need_cam, bw);
        if (tid == OPT_SAMPLE_WORKERS) red16[OPT_SAMPLE_WORKERS / WAVE] = 0.f;       // (its share of the block sum)
        __syncthreads();
    } else {
        // (inside the loop nobody reads the per-vertex depths: the 12 KB per sample and iteration are not written -- the forward that
        // closes optimize(), opt_sample_loss_kernel, writes the ones that are exported)
        sdf_sample_fused(vl, ws, io.loss_batch + 2 * B, B, gs, io.hand_type_array, red16, b, OPT_SAMPLE_WORKERS, bw[0].g, bw[1].g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wave's DMA writes to LDS have landed; the barrier publishes them)
    __syncthreads();         // the gradients of this sample and the DMA'd records: written above by this workgroup, read below by it
    TAIL_TK(0);
    // ---- phase 2: LBS backward of both hands
    lbs_bwd1_hand<true>(m, wk.lbs, B, hl * B + b, tid % LBS_THREADS, bw[hl], tail_part + (size_t)hl * m.nseg * 12, nullptr, nullptr,
                        wk.g_orient, wk.g_shape, wk.g_trans, need_mask, &bw[1]);
    if (!STEP) { TAIL_TK(1); return; }
    __syncthreads();         // the parameter gradients of this sample are in place
    TAIL_TK(1);
    // ---- phase 3: the optimizer step of this iteration, then both skeletons of the next one (threads [0,192) / [192,384))
    if (b == 0 && tid >= 384 && tid < 384 + SDF_NZERO) sdf_zero_counter(inside_count, tid - 384);   // the next iteration's collision kernels start from zero
    if (st.mask && tid < OPT_NPARAM) { opt_snapshot_losses(io, B, st, b, tid); opt_param_apply(io, wk, B, st, b, tid); }
    __syncthreads();         // the updated parameters are read back below by other threads of this workgroup
    TAIL_TK(2);
    // (SKIN: the vertex data of phase 4 is requested here, ahead of the skeleton chain: v_posed does not change in such a stage)
    constexpr int VR = (NV + LBS_THREADS - 1) / LBS_THREADS;
    const int lt = tid % LBS_THREADS, hv = hl * B + b;
    float vp[VR][3];
    float4 wr[VR];
    uint32_t jr[VR];
    if (SKIN) {
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = min(lt + r * LBS_THREADS, NV - 1);
            const float* s0 = wk.lbs.v_posed + ((size_t)hv * NV + v) * 3;
            vp[r][0] = s0[0]; vp[r][1] = s0[1]; vp[r][2] = s0[2];
            wr[r] = m.sparse4 ? m.w4_w[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            jr[r] = m.sparse4 ? m.w4_j[v] : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SKIN && (need_mask & 7) == 0) {
        // Translation stage: neither skeleton changes -- the records DMA'd into LDS in phase 0 ARE the next iteration's -- except the left hand's
        // shift (= hand_trans + right wrist - mirrored left wrist) and with it the left hand's posed joints.  The three shift components are
        // recomputed with lbs_skel_hand's own expression (the same bits) instead of both skeletons (Rodrigues, joint regression, chain by level:
        // 3.8 us of the 512-sample launch).
        if (tid < 3) {
            const float* br = io.shape + (size_t)b * 10;                 // right wrist: J_template[0] + J_shapedirs[0] . beta_right
            float jr = m.J_template[tid];
#pragma unroll
            for (int l = 0; l < 10; ++l) jr = __builtin_fmaf(m.J_shapedirs[tid * 10 + l], br[l], jr);
            const float* sJl = bw[1].sk + SK_J;
            const float jl = tid == 0 ? -sJl[0] : sJl[tid];              // mirrored left wrist
            const float sh = io.trans[b * 3 + tid] + (jr - jl);
            bw[1].sk[SK_SHIFT + tid] = sh;
            wk.lbs.skel[((size_t)B + b) * SK_STRIDE + SK_SHIFT + tid] = sh;
        }
        __syncthreads();
        if (tid < NJ * 3) {
            const int j = tid / 3, k = tid % 3;
            const float val = bw[1].sk[SK_G + 12 * j + 4 * k + 3];
            wk.joints_raw[((size_t)b * 42 + 21 + j) * 3 + k] = (k == 0 ? -val : val) + bw[1].sk[SK_SHIFT + k];
        }
    } else {
        const int hs = tid / 192;
        lbs_skel_hand<true>(m, io.orient, io.pose, io.shape, io.trans, B, wk.lbs.skel, wk.joints_raw, bw[hs < 2 ? hs : 0].sk, (hs < 2 ? hs : 0) * B + b, tid % 192,
                            hs < 2);
    }
    TAIL_TK(3);
    if (!SKIN) return;
    // ---- phase 4: the next iteration's vertices (threads [0,256) right hand, [256,512) left hand; the skinning matrices A and the left
    //      hand's shift are in this hand's LDS record -- visible since the barrier that closes lbs_skel_hand's last LDS phase)
    const float* sA = bw[hl].sk + SK_A;
    const float* sShift = bw[hl].sk + SK_SHIFT;
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int v = lt + r * LBS_THREADS;
        if (v >= NV) break;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        if (m.sparse4) {
            const float wv[4] = {wr[r].x, wr[r].y, wr[r].z, wr[r].w};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float4* A4 = reinterpret_cast<const float4*>(sA + 12 * (int)((jr[r] >> (8 * sI)) & 0xffu));
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 a = A4[q];
                    T[4 * q] = __builtin_fmaf(wv[sI], a.x, T[4 * q]);
                    T[4 * q + 1] = __builtin_fmaf(wv[sI], a.y, T[4 * q + 1]);
                    T[4 * q + 2] = __builtin_fmaf(wv[sI], a.z, T[4 * q + 2]);
                    T[4 * q + 3] = __builtin_fmaf(wv[sI], a.w, T[4 * q + 3]);
                }
            }
        } else {
            for (int j = 0; j < NJ; ++j) {
                const float wj = m.weights[v * NJ + j];
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = __builtin_fmaf(wj, sA[12 * j + e], T[e]);
            }
        }
        float out[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) out[q] = T[4 * q + 0] * vp[r][0] + T[4 * q + 1] * vp[r][1] + T[4 * q + 2] * vp[r][2] + T[4 * q + 3];
        if (hl == 1) {  // optimize_model.py:210-211, 222-228
            out[0] = -out[0] + sShift[0];
            out[1] = out[1] + sShift[1];
            out[2] = out[2] + sShift[2];
        }
        float* dst = io.verts + ((size_t)hv * NV + v) * 3;
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2];
#pragma unroll
        for (int t = 0; t < IHMR_NUM_TIPS; ++t)
            if (v == m.tip_ids[t]) {  // fingertip joints are vertices (:201-202)
                float* jd = wk.joints_raw + ((size_t)b * 42 + (hl ? 21 : 0) + NJ + t) * 3;
                jd[0] = out[0]; jd[1] = out[1]; jd[2] = out[2];
            }
    }
    TAIL_TK(4);
}

// IHMR-MLP, a stage that moves ONLY the camera (mlp_default's last: `pred_cam_params`, filter and select on joints_2d_loss_p;
// strategies/mlp_default.py) -- round 6.  The reference re-evaluates everything for it (mlp_model.py:504-511: MANO of both hands, the
// collision term), but no vertex, no 3-D joint and no penetration depth depends on the camera: the hands of the state the stage starts
// from -- every sample's ACCEPTED state -- are what they were when that state was evaluated.  So the evaluation is one launch of one wave
// per sample: the joint losses (opt_loss_wave, the same function, hence the same bits as a full evaluation) on the accepted state's raw
// joints (MlpSelect::acc_joints) with the new camera, the collision loss taken over from the "prev" table (what a re-evaluation of the
// same vertices returns bit for bit: the kernels are deterministic), then the keep / reject decision.  grid = B, block = 256.
__global__ __launch_bounds__(256) void mlp_camera_select_kernel(ihmr_opt_io io, OptWork wk, int B, ihmr_opt_weights w, MlpSelect sel) {
    __shared__ LossShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < WAVE) {
        opt_loss_wave(io, wk, B, w, sh, b, tid, 0);          // (wk.joints_raw = the accepted joints: set by the launcher)
        if (tid == 0) io.loss_batch[(size_t)2 * B + b] = sel.prev_loss[(size_t)sel.idx[b] * 3 + 2];
    }
    mlp_select_sample(sel, io.loss_batch, B, b);
}

// The reference's packed prediction vector final_params (B,122) = [cam 3 | R orient 3 | R pose 45 | L orient 3 |
// L pose 45 | R shape 10 | L shape 10 | trans 3] (baseline_model.py:262-270, mlp_model.py:426-439) scattered into
// the per-group state buffers of ihmr_opt_io.  grid = B, block = 128.
__global__ __launch_bounds__(128) void opt_unpack_params_kernel(ihmr_opt_io io, const float* __restrict__ packed, int B) {
    const int b = blockIdx.x, e = threadIdx.x;
    if (e >= 122) return;
    const float v = packed[(size_t)b * 122 + e];
    if (e < 3) io.cam[b * 3 + e] = v;
    else if (e < 6) io.orient[(size_t)b * 3 + (e - 3)] = v;
    else if (e < 51) io.pose[(size_t)b * 45 + (e - 6)] = v;
    else if (e < 54) io.orient[((size_t)B + b) * 3 + (e - 51)] = v;
    else if (e < 99) io.pose[((size_t)B + b) * 45 + (e - 54)] = v;
    else if (e < 109) io.shape[(size_t)b * 10 + (e - 99)] = v;
    else if (e < 119) io.shape[((size_t)B + b) * 10 + (e - 109)] = v;
    else io.trans[b * 3 + (e - 119)] = v;
}

// utils/opt_utils.py:104-153: validity filter (every criterion of the stage: loss <= origin * factor), 1e11 for
// invalid rows, row 0 restored, first argmin of the select loss, selected parameters written back.
// One thread per sample.  A loss without a criterion is NOT compared at all (use_filter = 0), as in the reference.
__global__ void opt_select_kernel(ihmr_opt_io io, int B, int S, ihmr_opt_stage sg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float bar[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) bar[l] = io.snap_loss[(size_t)l * B + b] * sg.filter_factor[l];
    int best = 0;
    float best_v = io.snap_loss[(size_t)sg.select_loss * B + b];
    for (int s = 1; s < S; ++s) {
        bool valid = true;
#pragma unroll
        for (int l = 0; l < 3; ++l)
            if (sg.use_filter[l]) valid = valid && (io.snap_loss[((size_t)s * 3 + l) * B + b] <= bar[l]);
        const float key = valid ? io.snap_loss[((size_t)s * 3 + sg.select_loss) * B + b] : 100000000000.0f;
        if (key < best_v) { best_v = key; best = s; }
    }
    io.selected[b] = best;
    for (int e = 0; e < OPT_NPARAM; ++e) {
        const int blk = opt_slot_block(e);
        if ((sg.param_mask >> blk) & 1) *opt_slot_ptr(io, B, b, e, blk) = io.snap_params[((size_t)best * B + b) * OPT_NPARAM + e];
    }
}
