// IHMR-MLP inference glue on the device (BASELINE.json configs[2]; the reference's models/mlp_model.py):
//   mlp_layer_kernel    four launches per stage = retrive_prev_prediction (:408-423: the batch's rows of the "prev" tables, by dataset
//                       index) -> the stage's sub-network (networks.py:83-105: Linear 1146-512-256-128-k with ReLU, input
//                       [img_feat | final_params]) -> __update_params_single (:459-472: the residual added to the stage's columns)
//                       -> the 122-vector scattered into the fused kernels' parameter buffers (= opt_unpack_params_kernel)
//   mlp_select_sample   select_better_params (:592-637) + save_pred_to_prev (:337-356), folded into the forward / loss launch
//                       that has just produced the new losses (opt_sample_loss_kernel)
// Before: ~25 torch element-wise launches per stage (gather, cat, clone, slice adds, compares, where, scatters) around four
// GEMM + split-K-reduce pairs.
//
// The sub-network at M = batch (128) rows is 192 MFLOP over 3 MB of weights: far too little for one workgroup per row block (the
// weights would stream through four CUs), so every layer runs as 16 x 16 output tiles, one workgroup each (256 / 128 / 64 / 8-48
// workgroups at batch 128), one launch per layer: 7 + 5 + 5 + 5 us.  (All four layers in ONE launch with grid-wide barriers between
// them -- 256 co-resident workgroups, atomic counter + polling -- took 59 us: a barrier over 256 workgroups on one address costs
// ~12 us here, three kernel boundaries ~1.5 us each.  Measured, dropped.)  A tile's four waves split the K loop
// (v_mfma_f32_16x16x4_f32: exact fp32 = a k-ordered fmaf chain per wave), the four partial tiles are added in fixed order through
// LDS, bias + ReLU, result to a small global buffer the next layer reads (L2).  Operands go from global memory straight to the
// MFMA registers: every operand element is used by exactly one MFMA of the workgroup.  The k order inside a 16-wide chunk is
// permuted (step j takes k = kb + 4 g + j of lane group g) so that a lane's A operands of four steps are ONE 16-byte load; any
// fixed order is as good as another, and the oracle comparison is by tolerance.
#pragma once
#include "ihmr_common.h"

#define MLPI_THREADS 256
#define MLPI_IN 1146                 // total_params_dim + 1024 (networks.py:87)
#define MLPI_KPAD 1152               // ... padded to the 16-row packing of the weight (zeros beyond 1146)
#define MLPI_MAX_WG 256

struct MlpHeadArgs {
    // retrive_prev_prediction (mlp_model.py:408-423) gathers the batch's rows of img_feat_all / prev_final by dataset index; the rows it
    // returns are the ones the previous launch (mlp_select_sample) has just scattered there from `img_feat` / `final_out`, which it also
    // keeps per batch row: the head reads those (no index load in front of every operand load; duplicate indices in a batch are padding
    // copies of one sample, opt_dataset.py:49-51, i.e. identical rows either way)
    const float* feat;               // (B, 1024)  the batch's image features
    const float* prev;               // (B, 122)   the batch's final_params after the previous stage
    const float* w[4];               // K-major packed weights [Kpad][ldw] (ihmr_amd/networks.py:_Packed)
    const float* b[4];
    int ldw[4];
    int kout;                        // outputs of the last layer = sum of the stage's update sizes
    unsigned char col[128];          // output j -> column of the 122-vector (mlp_model.py:426-439 order)
    float* h[3];                     // hidden activations (B,512), (B,256), (B,128)
    float* new_params;               // (B,122) out: prev + residual on the stage's columns
    int B;
};

// the 122-vector's column e of sample b -> the fused kernels' parameter buffers (= opt_unpack_params_kernel)
__device__ __forceinline__ void mlpi_unpack_store(const ihmr_opt_io& io, int B, int b, int e, float v) {
    if (e < 3) io.cam[b * 3 + e] = v;
    else if (e < 6) io.orient[(size_t)b * 3 + (e - 3)] = v;
    else if (e < 51) io.pose[(size_t)b * 45 + (e - 6)] = v;
    else if (e < 54) io.orient[((size_t)B + b) * 3 + (e - 51)] = v;
    else if (e < 99) io.pose[((size_t)B + b) * 45 + (e - 54)] = v;
    else if (e < 109) io.shape[(size_t)b * 10 + (e - 99)] = v;
    else if (e < 119) io.shape[((size_t)B + b) * 10 + (e - 109)] = v;
    else io.trans[b * 3 + (e - 119)] = v;
}

typedef float mlpi_f4 __attribute__((ext_vector_type(4)));

// One layer: out[r][c] = act(sum_k A[r][k] W[k][c] + bias[c]) over 16 x 16 tiles, tile t -> workgroup t mod grid.
// LAYER 0 reads A = [img_feat | final_params | 0]; LAYER 3 adds the residual and scatters instead of storing.
// CH = 16-wide k chunks per wave and batch of loads (K = 4 waves x NB x CH x 16).
template <int LAYER, int CH, int NB>
__device__ __forceinline__ void mlpi_layer(const MlpHeadArgs& a, const ihmr_opt_io& io, float (*red)[MLPI_THREADS]) {
    constexpr int KIN = LAYER == 0 ? MLPI_KPAD : (LAYER == 1 ? 512 : (LAYER == 2 ? 256 : 128));
    static_assert(4 * NB * CH * 16 == KIN, "the four waves cover K exactly");
    const int N = LAYER == 0 ? 512 : (LAYER == 1 ? 256 : (LAYER == 2 ? 128 : a.kout));
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE, B = a.B;
    const int ntr = (B + 15) / 16, ntc = (N + 15) / 16;
    const int m = lane & 15, g = lane >> 4;
    const float* __restrict__ W = a.w[LAYER];
    const int ldw = a.ldw[LAYER];
    for (int tile = (int)blockIdx.x; tile < ntr * ntc; tile += (int)gridDim.x) {
        const int r0 = (tile / ntc) * 16, c0 = (tile % ntc) * 16;
        const int row = r0 + m;
        const bool rok = row < B;
        mlpi_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int nb = 0; nb < NB; ++nb) {
            mlpi_f4 av[CH];
            float bv[CH][4];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int kb = (wave * NB + nb) * CH * 16 + c * 16;      // this wave's chunk
                const int k4 = kb + 4 * g;
                av[c] = mlpi_f4{0.f, 0.f, 0.f, 0.f};
                if (rok) {
                    if (LAYER == 0) {
                        if (k4 < 1024) av[c] = *reinterpret_cast<const mlpi_f4*>(a.feat + (size_t)row * 1024 + k4);
                        else {
                            const float* p = a.prev + (size_t)row * 122 + (k4 - 1024);
#pragma unroll
                            for (int j = 0; j < 4; ++j) av[c][j] = (k4 - 1024 + j) < 122 ? p[j] : 0.f;
                        }
                    } else {
                        av[c] = *reinterpret_cast<const mlpi_f4*>(a.h[LAYER > 0 ? LAYER - 1 : 0] + (size_t)row * KIN + k4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[c][j] = W[(size_t)(k4 + j) * ldw + c0 + m];     // (columns beyond N: the packing's zeros)
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][j], bv[c][j], acc, 0, 0, 0);
        }
        // the four waves' partial tiles, added in wave order: element v = lane * 4 + i <-> column lane & 15, row (lane >> 4) * 4 + i
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = acc[i];
        __syncthreads();
        {
            const float s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            const int ol = tid >> 2, oi = tid & 3;
            const int orow = r0 + (ol >> 4) * 4 + oi, ocol = c0 + (ol & 15);
            if (orow < B && ocol < N) {
                float v = s + a.b[LAYER][ocol];
                if (LAYER < 3) {
                    a.h[LAYER < 3 ? LAYER : 0][(size_t)orow * N + ocol] = fmaxf(v, 0.f);
                } else {
                    const int e = (int)a.col[ocol];
                    v = a.prev[(size_t)orow * 122 + e] + v;       // new[:, COLS[n]] += res[:, o:o+k]
                    a.new_params[(size_t)orow * 122 + e] = v;
                    mlpi_unpack_store(io, B, orow, e, v);
                }
            }
        }
        __syncthreads();             // `red` is free for the next tile
    }
}

// grid = the layer's tiles (<= 256), block = 256.  Layer 0 also copies prev -> new (every column; the last layer overwrites the stage's
// columns: ordered by the kernel boundaries in between) and scatters it to the parameter buffers.
template <int LAYER>
__global__ __launch_bounds__(MLPI_THREADS) void mlp_layer_kernel(MlpHeadArgs a, ihmr_opt_io io) {
    __shared__ float red[4][MLPI_THREADS];
    if (LAYER == 0) {
        for (int i = (int)(blockIdx.x * MLPI_THREADS + threadIdx.x); i < a.B * 122; i += (int)(gridDim.x * MLPI_THREADS)) {
            const int b = i / 122, e = i % 122;
            const float v = a.prev[i];
            a.new_params[i] = v;
            mlpi_unpack_store(io, a.B, b, e, v);
        }
        mlpi_layer<0, 9, 2>(a, io, red);
    } else if (LAYER == 1) mlpi_layer<1, 8, 1>(a, io, red);
    else if (LAYER == 2) mlpi_layer<2, 4, 1>(a, io, red);
    else mlpi_layer<3, 2, 1>(a, io, red);
}

// ------------------------------------------------------------------------------------------ select + save
// mlp_model.py:592-637 + :337-356 for sample b, by the workgroup that has just written the sample's new losses (loss_batch rows
// IHMR_LOSS_*): keep the stage's update only if every filter loss got better than prev * (1 + pct / 100) (strictly) and the select
// loss did not get worse; the kept / fallen-back row and its losses go to the "prev" tables (by dataset index) and to `final_out`.
struct MlpSelect {
    int mode;                        // 0: off (IHMR-OPT callers), 1: first evaluation of a batch (nothing to compare: save), 2: a stage
    int n_filter, filter_loss[4];
    float filter_factor[4];          // float32(1 + pct / 100), the factor torch multiplies the prev loss by
    int select_loss;
    const long long* idx;
    const float* new_params;         // (B,122) what the forward evaluated
    const float* img_feat;           // (B,1024) mode 1
    unsigned char* data_idxs_all;    // (num_data) bool
    float* img_feat_all;             // (num_data,1024)
    float* prev_final;               // (num_data,122)
    float* prev_loss;                // (num_data,3)
    float* final_out;                // (B,122) the batch's state after this stage
    unsigned char* kept;             // (B) this stage's decision
    // round 6: the raw joints (B,42,3) of every sample's ACCEPTED state, kept per batch so that a stage that moves only the camera can be
    // judged without re-evaluating the hands (mlp_camera_select_kernel): joints_now = what the forward of this evaluation produced,
    // copied to acc_joints for a sample that keeps the update (or is saved for the first time)
    const float* joints_now;
    float* acc_joints;
};

// called by all threads of the sample's workgroup after the losses of sample b are in loss_batch
__device__ __forceinline__ void mlp_select_sample(const MlpSelect& s, const float* __restrict__ loss_batch, int B, int b) {
    __syncthreads();                 // the sample's three losses were written by threads of this workgroup
    const int tid = threadIdx.x;
    const long long r = s.idx[b];
    float nl[3], pl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { nl[c] = loss_batch[(size_t)c * B + b]; pl[c] = s.mode == 2 ? s.prev_loss[(size_t)r * 3 + c] : 0.f; }
    bool ok = true;
    if (s.mode == 2) {
        for (int f = 0; f < s.n_filter; ++f) {
            const int c = s.filter_loss[f];
            ok = ok && (nl[c] < pl[c] * s.filter_factor[f]);
        }
        ok = ok && (nl[s.select_loss] <= pl[s.select_loss]);
    }
    float v = 0.f;
    if (tid < 122) {
        v = s.new_params[(size_t)b * 122 + tid];
        if (!ok) v = s.prev_final[(size_t)r * 122 + tid];      // rejected: the sample falls back to its previous parameters
    }
    __syncthreads();                 // every thread has read the old rows
    if (tid < 122) {
        s.prev_final[(size_t)r * 122 + tid] = v;
        s.final_out[(size_t)b * 122 + tid] = v;
    } else if (tid < 125) {
        s.prev_loss[(size_t)r * 3 + (tid - 122)] = ok ? nl[tid - 122] : pl[tid - 122];
    } else if (tid == 125) {
        s.kept[b] = ok ? 1 : 0;
        s.data_idxs_all[r] = 1;
    } else if (tid >= 128 && tid < 128 + 126 && ok && s.acc_joints && s.joints_now) {
        s.acc_joints[(size_t)b * 126 + (tid - 128)] = s.joints_now[(size_t)b * 126 + (tid - 128)];
    }
    if (s.mode == 1)
        for (int k = tid; k < 1024; k += (int)blockDim.x) s.img_feat_all[(size_t)r * 1024 + k] = s.img_feat[(size_t)b * 1024 + k];
}
