// MANO linear blend skinning -- forward and analytic backward.
//
// Replaces smplx 0.1.28 `MANO.forward` + `lbs()` as the reference calls it
// (models/optimize_model.py:194-198) and, in TWO_HAND mode, the whole of
// `OptimizeModel.get_mano_output` (:171-232): mirror of the left axis-angles (:180-188), right-hand
// model on 2B hands, 5 fingertip vertices appended (:201-202), x-negation of the left outputs
// (:210-211), left hand shifted by hand_trans + (right wrist - left wrist) (:222-228).
//
// MI355X decomposition (hands are independent, the 1.3 MB pose-blend basis is shared by all of them):
//   skel  : one small workgroup per hand -- Rodrigues, J = J_template + J_shapedirs.beta (the joint
//           regression folded into constants), kinematic chain by tree level, skinning transforms A_j,
//           pose feature; 3 KB record per hand in HBM, reused by the backward of the same iteration.
//   skin  : workgroup = 8 hands x 195 vertices, lane <-> vertex.  Each float4 of the pose/shape basis is
//           loaded ONCE (coalesced 16 B/lane) and applied to all 8 hands from LDS-broadcast coefficients,
//           so the basis is streamed from L2 N/8 times instead of N times.
//   bwd1  : one workgroup per hand -- d v_posed, the 16x12 transform gradients as fixed-order CSR-by-joint
//           sums, chain backward by tree level, orient / shape / translation gradients.
//   bwd2  : (finger-pose stage only) d pose_feature = posedirs . d v_posed as an LDS-tiled product,
//           workgroup = 8 hands x 192 basis columns, partial sums per column chunk (deterministic).
//   bwd3  : (finger-pose stage only) chunk reduction + Rodrigues backward.
#pragma once
#include "ihmr_common.h"

#define LBS_THREADS 256
#define LBS_HG 8          // hands per skin workgroup (large launches)
#define LBS_HG_SMALL 4    // ... of launches of up to LBS_SMALL_MAX_HANDS hands
#define LBS_SMALL_MAX_HANDS 256
#define LBS_TILE_V 195    // vertices per skin workgroup (4 x 195 = 780 >= 778)
#define LBS_KG 25         // bwd2: K groups (split-K partial sums, reduced in fixed order by bwd3)
#define LBS_KC 3          // bwd2: 32-column chunks per K group: 25 x 3 x 32 = 2400 >= 2334 basis columns
#define LBS_SEG 13         // CSR entries per dA segment
#define LBS_SEG_CAP 1024   // >= 778*16/13 + 16: every weight matrix fits
#define LBS_CSR_CAP 4096   // non-zero skinning weights staged in LDS by bwd1 (MANO-like: <= 4-5 per vertex); else read from L2

// skeleton record (floats), one per hand
#define SK_R 0        // [16][9]
#define SK_J 144      // [16][3]
#define SK_G 192      // [16][12] world transforms [R | t]
#define SK_A 384      // [16][12] skinning transforms [G.R | G.t - G.R J]
#define SK_PF 576     // [136] pose feature (135 used)
#define SK_POSE 712   // [48] full pose (+ mean), mirrored for left hands
#define SK_BETA 760   // [10]
#define SK_SHIFT 770  // [3] TWO_HAND left-hand shift
#define SK_STRIDE 784

struct LbsWork {       // carved from the caller's workspace
    float* skel;       // [N][SK_STRIDE]
    float* v_posed;    // [N][2334]
    float* dvp;        // [N][2334]
    float* chain;      // [N][192]: dR [16][9] from the chain, dJ [16][3]
    float* dpf_part;   // [LBS_KG][N][136]
    float* pose_off;   // [N][2334] the pose-blend offsets P of the hands' current finger poses (round 5; see lbs_skin_kernel MODE 2)
};

static inline size_t lbs_ws_bytes(int N) {
    size_t n = (size_t)N * (SK_STRIDE + 3 * NV3 + 192 + LBS_KG * 136) * sizeof(float);
    return ((n + 255) & ~(size_t)255) + 6 * 256;
}

static inline LbsWork lbs_carve(void* ws, int N) {
    LbsWork w;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return (float*)r; };
    w.skel = take((size_t)N * SK_STRIDE * 4);
    w.v_posed = take((size_t)N * NV3 * 4);
    w.dvp = take((size_t)N * NV3 * 4);
    w.chain = take((size_t)N * 192 * 4);
    w.dpf_part = take((size_t)LBS_KG * N * 136 * 4);
    w.pose_off = take((size_t)N * NV3 * 4);
    return w;
}

// ------------------------------------------------------------------------------------- skeleton
// 192 threads per hand h (tid = 0..191; every thread of the workgroup must call it: block-wide barriers inside; threads that own
// no hand call it with active = false and only take part in the barriers).
// TWO_HAND: hands [0,B) right, [B,2B) left of sample (h - B); joints out is (B,42,3) (posed joints only; the
// 5 tips are written by the skin kernel), else (N,16,3).  sk = SK_STRIDE floats of LDS owned by this hand.
template <bool TWO_HAND>
__device__ __forceinline__ void lbs_skel_hand(const ihmr_mano& m, const float* __restrict__ orient,
                                              const float* __restrict__ pose, const float* __restrict__ betas,
                                              const float* __restrict__ trans, int B, float* __restrict__ skel,
                                              float* __restrict__ joints, float* sk, int h, int tid_in, bool active = true) {
    const bool left = TWO_HAND && h >= B;
    const int tid = active ? tid_in : (1 << 20);       // an inactive thread fails every range test below
    float* sR = sk + SK_R; float* sJ = sk + SK_J; float* sG = sk + SK_G; float* sA = sk + SK_A;
    const int my_depth = active ? m.depth[tid_in / 12] : -1, my_parent = active ? m.parents[tid_in / 12] : 0;   // kinematic tree: read once, up front
    float* sPF = sk + SK_PF; float* sPose = sk + SK_POSE; float* sBeta = sk + SK_BETA; float* sShift = sk + SK_SHIFT;
    if (tid < 48) {
        float v = tid < 3 ? orient[h * 3 + tid] : pose[h * 45 + tid - 3];
        if (left && (tid % 3) != 0) v = -v;  // optimize_model.py:180-188
        sPose[tid] = v + m.pose_mean[tid];
    }
    if (tid >= 64 && tid < 74) sBeta[tid - 64] = betas[h * 10 + tid - 64];
    if (tid >= 128 && tid < 142) sk[SK_SHIFT + tid - 128] = 0.f;  // shift + padding
    if (tid == 150) sPF[135] = 0.f;
    __syncthreads();
    if (tid < NJ) rodrigues_fwd(&sPose[3 * tid], &sR[9 * tid]);
    if (tid >= 64 && tid < 64 + 48) {
        const int e = tid - 64;
        float acc = m.J_template[e];
#pragma unroll
        for (int l = 0; l < 10; ++l) acc = __builtin_fmaf(m.J_shapedirs[e * 10 + l], sBeta[l], acc);
        sJ[e] = acc;
    }
    __syncthreads();
    if (tid < NPF) {
        const int j = 1 + tid / 9, e = tid % 9;
        sPF[tid] = sR[9 * j + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    if (tid >= 160 && tid < 172) {
        const int e = tid - 160, r = e / 4, c = e % 4;
        sG[e] = c < 3 ? sR[3 * r + c] : sJ[r];
    }
    if (TWO_HAND && left && tid >= 176 && tid < 179) {
        // right wrist of the same sample: J_r[0] = J_template[0] + J_shapedirs[0] . beta_right
        const int k = tid - 176;
        const float* br = betas + (h - B) * 10;
        float jr = m.J_template[k];
#pragma unroll
        for (int l = 0; l < 10; ++l) jr = __builtin_fmaf(m.J_shapedirs[k * 10 + l], br[l], jr);
        const float jl = k == 0 ? -sJ[0] : sJ[k];  // mirrored left wrist
        sShift[k] = trans[(h - B) * 3 + k] + (jr - jl);
    }
    __syncthreads();
    // kinematic chain, level by level (MANO: depth <= 3); 12 lanes per joint
    for (int d = 1; d <= m.max_depth; ++d) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        if (my_depth == d) {
            const int p = my_parent;
            sG[12 * j + e] = lbs_chain_elem(sG + 12 * p, sR + 9 * j, sJ + 3 * j, sJ + 3 * p, r, c);
        }
        __syncthreads();
    }
    if (active) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        sA[12 * j + e] = lbs_rel_elem(sG + 12 * j, sJ + 3 * j, r, c);
    }
    __syncthreads();
    for (int i = tid; i < SK_STRIDE; i += 192) skel[(size_t)h * SK_STRIDE + i] = sk[i];
    if (tid < NJ * 3) {
        const int j = tid / 3, k = tid % 3;
        float val = sG[12 * j + 4 * k + 3];
        if (!TWO_HAND) {
            joints[((size_t)h * NJ + j) * 3 + k] = val;
        } else {
            if (left) val = (k == 0 ? -val : val) + sShift[k];
            const int b = left ? h - B : h;
            joints[((size_t)b * 42 + (left ? 21 : 0) + j) * 3 + k] = val;
        }
    }
}

// seam A / generic: grid = N hands, block = 192
template <bool TWO_HAND>
__global__ __launch_bounds__(192) void lbs_skel_kernel(ihmr_mano m, const float* __restrict__ orient,
                                                       const float* __restrict__ pose, const float* __restrict__ betas,
                                                       const float* __restrict__ trans, int B, float* __restrict__ skel,
                                                       float* __restrict__ joints) {
    __shared__ float sk[SK_STRIDE];
    lbs_skel_hand<TWO_HAND>(m, orient, pose, betas, trans, B, skel, joints, sk, blockIdx.x, threadIdx.x);
}

typedef float lbs_v2f __attribute__((ext_vector_type(2)));

// hand i of group (x, s) of HG hands: x + 8 * (HG s + i) -- all hands of a group share (hand % 8), i.e. the XCD that ran
// their skeleton workgroup and will run their collision / backward workgroups (speed only).
template <int HG>
__device__ __forceinline__ int lbs_group_hand(int x, int s, int i) { return x + 8 * (HG * s + i); }

// ------------------------------------------------------------------------------------- skin
// grid = (8, 4 vertex tiles x ceil(N / (8 HG)) groups), block = 256 (195 active lanes = vertices); HG = hands per workgroup: 8
// (LBS_HG) for large launches -- the basis rows a workgroup streams from L2 serve eight hands --, 4 for launches of up to
// LBS_SMALL_MAX_HANDS hands, where the kernel is as long as one thread's chain of FMAs (half as long with half the hands).
// REUSE: v_posed_ws already holds v_posed of exactly these pose and shape parameters (a refinement stage that updates
// neither -- translation, global orientation: optimize_model.py:393-407 -- after its first iteration): both blends are
// skipped and the stored values (the bits a recomputation would give) are skinned with the new joint transforms.  The two
// blends are 2/3 of the kernel's arithmetic and all of its L2 traffic (1.8 MB of basis rows per 8 hands and vertex tile).
// MODE (round 5): LBS_MODE_FULL both blends (pose_off_ws != nullptr: the pose offsets P are stored as well); LBS_MODE_REUSE as above;
// LBS_MODE_KEEP_P: pose_off_ws holds P of exactly these finger poses (a stage that moves the shape but not the pose, after its first
// iteration): v_shaped = v_template + S is recomputed, the 135 pose rows are NOT read -- v_posed = v_shaped + P with the stored P is
// the very operation the full kernel ends with, on the same bits (that is what summing the offsets from zero bought besides accuracy).
#define LBS_MODE_FULL 0
#define LBS_MODE_REUSE 1
#define LBS_MODE_KEEP_P 2
template <bool TWO_HAND, int MODE = LBS_MODE_FULL, int HG = LBS_HG>
__global__ __launch_bounds__(LBS_THREADS) void lbs_skin_kernel(ihmr_mano m, const float* __restrict__ skel, int N, int B,
                                                               float* __restrict__ verts, float* __restrict__ joints,
                                                               float* __restrict__ v_posed_ws, float* __restrict__ pose_off_ws) {
    constexpr bool REUSE = MODE == LBS_MODE_REUSE, KEEP_P = MODE == LBS_MODE_KEEP_P;
    TL_SCOPE(6);
    __shared__ float4 pfT[136][HG / 4];   // [e][hands 0-3 | 4-7]
    __shared__ float A_s[HG / 2][192][2];   // skinning matrices, the two hands of a pair interleaved
    __shared__ float beta_s[10][HG];  // [l][hand]
    __shared__ float shift_s[HG][4];
    __shared__ float vsh_s[MODE != LBS_MODE_FULL ? 1 : 3 * HG][MODE != LBS_MODE_FULL ? 1 : LBS_THREADS];   // v_shaped of the thread's HG hands, parked while the pose offsets are summed
    const int tid = threadIdx.x, gx = blockIdx.x, tile = blockIdx.y % 4, gs = blockIdx.y / 4;
    if (MODE == LBS_MODE_FULL)
        for (int idx = tid; idx < HG * 136; idx += LBS_THREADS) {
            const int hh = idx / 136, e = idx % 136, hid = lbs_group_hand<HG>(gx, gs, hh);
            const float v = (hid < N && e < NPF) ? skel[(size_t)hid * SK_STRIDE + SK_PF + e] : 0.f;
            reinterpret_cast<float*>(&pfT[e][0])[hh] = v;
        }
    for (int idx = tid; idx < HG * 192; idx += LBS_THREADS) {
        const int hh = idx / 192, e = idx % 192, hid = lbs_group_hand<HG>(gx, gs, hh);
        A_s[hh / 2][e][hh % 2] = hid < N ? skel[(size_t)hid * SK_STRIDE + SK_A + e] : 0.f;
    }
    if (!REUSE && tid < HG * 10) {
        const int hh = tid / 10, l = tid % 10, hid = lbs_group_hand<HG>(gx, gs, hh);
        beta_s[l][hh] = hid < N ? skel[(size_t)hid * SK_STRIDE + SK_BETA + l] : 0.f;
    }
    if (tid >= 128 && tid < 128 + HG * 4) {
        const int hh = (tid - 128) / 4, k = (tid - 128) % 4, hid = lbs_group_hand<HG>(gx, gs, hh);
        shift_s[hh][k] = (hid < N && k < 3) ? skel[(size_t)hid * SK_STRIDE + SK_SHIFT + k] : 0.f;
    }
    __syncthreads();
    const int v = tile * LBS_TILE_V + tid;
    if (tid >= LBS_TILE_V || v >= NV) return;

    // shape blend: v_shaped = v_template + shapedirs . beta   (8 hands at once, as 4 hand PAIRS: packed fp32 FMAs
    // do two hands per instruction and give the same IEEE results as scalar ones)
    lbs_v2f vq[HG / 2][3];
    if (REUSE) {
#pragma unroll
        for (int q = 0; q < HG / 2; ++q) {
            const int h0 = lbs_group_hand<HG>(gx, gs, 2 * q), h1 = lbs_group_hand<HG>(gx, gs, 2 * q + 1);
            const float* s0 = v_posed_ws + ((size_t)min(h0, N - 1) * NV + v) * 3;
            const float* s1 = v_posed_ws + ((size_t)min(h1, N - 1) * NV + v) * 3;
            vq[q][0] = lbs_v2f{s0[0], s1[0]}; vq[q][1] = lbs_v2f{s0[1], s1[1]}; vq[q][2] = lbs_v2f{s0[2], s1[2]};
        }
    }
    if (!REUSE) {
        const float4 t = m.vt4[v];
        float4 sd[10];
#pragma unroll
        for (int l = 0; l < 10; ++l) sd[l] = m.sd4[l * NVP + v];
        // The blend offsets are summed from ZERO and added in one operation each -- v_shaped = v_template + S, v_posed = v_shaped + P,
        // as the reference's smplx `lbs` does.  (Rounds 1-4 ran the 145 fused multiply-adds onto a running VERTEX: every one of them
        // rounds at the vertex's magnitude, ~1.3 ulp of the vertex in the result -- round 5's float64 arbiter found the forward 2.9 x as
        // far from the exact mesh as torch's; the same sums from zero round at the offsets' magnitude, a hundred times smaller.)
        // v_shaped waits in LDS (this thread's own slots: no barrier) while the registers sum the pose offsets.
#pragma unroll
        for (int q = 0; q < HG / 2; ++q) { vq[q][0] = lbs_v2f{0.f, 0.f}; vq[q][1] = lbs_v2f{0.f, 0.f}; vq[q][2] = lbs_v2f{0.f, 0.f}; }
#pragma unroll
        for (int l = 0; l < 10; ++l) {
#pragma unroll
            for (int q = 0; q < HG / 2; ++q) {
                const lbs_v2f bl = *reinterpret_cast<const lbs_v2f*>(&beta_s[l][2 * q]);
                vq[q][0] = __builtin_elementwise_fma(lbs_v2f{sd[l].x, sd[l].x}, bl, vq[q][0]);
                vq[q][1] = __builtin_elementwise_fma(lbs_v2f{sd[l].y, sd[l].y}, bl, vq[q][1]);
                vq[q][2] = __builtin_elementwise_fma(lbs_v2f{sd[l].z, sd[l].z}, bl, vq[q][2]);
            }
        }
#pragma unroll
        for (int q = 0; q < HG / 2; ++q) {
            const lbs_v2f tc[3] = {lbs_v2f{t.x, t.x}, lbs_v2f{t.y, t.y}, lbs_v2f{t.z, t.z}};
            const int h0 = lbs_group_hand<HG>(gx, gs, 2 * q), h1 = lbs_group_hand<HG>(gx, gs, 2 * q + 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const lbs_v2f vs = tc[c] + vq[q][c];
                if (KEEP_P) {       // v_posed = v_shaped + P, P as the full kernel stored it
                    const lbs_v2f pc = {pose_off_ws[((size_t)min(h0, N - 1) * NV + v) * 3 + c], pose_off_ws[((size_t)min(h1, N - 1) * NV + v) * 3 + c]};
                    vq[q][c] = vs + pc;
                } else {
                    vsh_s[6 * q + 2 * c][tid] = vs.x; vsh_s[6 * q + 2 * c + 1][tid] = vs.y;
                    vq[q][c] = lbs_v2f{0.f, 0.f};
                }
            }
        }
    }
    // pose blend: v_posed = v_shaped + pose_feature . posedirs.  The basis rows are fetched 9 at a time into two
    // register batches, the next batch in flight while the current one is consumed (explicit batches +
    // scheduling barriers: left alone, hipcc issues one load per use and waits vmcnt(0) on each).
    if (MODE == LBS_MODE_FULL) {
        // a basis row is fetched as two aligned float pairs (x,y) (z,pad): packed FMAs take their broadcast operand
        // straight from either half of such a pair, no register shuffling between the load and its use
        struct Row { lbs_v2f xy, zw; };
        Row pa[9], pb[9];
        auto fetch = [&](Row* p, int e0) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const lbs_v2f* src = reinterpret_cast<const lbs_v2f*>(&m.pd4[(e0 + u) * NVP + v]);
                p[u].xy = src[0];
                p[u].zw = src[1];
            }
        };
        auto consume = [&](const Row* p, int e0) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                lbs_v2f f[HG / 2];
#pragma unroll
                for (int q4 = 0; q4 < HG / 4; ++q4) {
                    const float4 fq = pfT[e0 + u][q4];
                    f[2 * q4] = lbs_v2f{fq.x, fq.y}; f[2 * q4 + 1] = lbs_v2f{fq.z, fq.w};
                }
#pragma unroll
                for (int q = 0; q < HG / 2; ++q) {
                    vq[q][0] = __builtin_elementwise_fma(f[q], __builtin_shufflevector(p[u].xy, p[u].xy, 0, 0), vq[q][0]);
                    vq[q][1] = __builtin_elementwise_fma(f[q], __builtin_shufflevector(p[u].xy, p[u].xy, 1, 1), vq[q][1]);
                    vq[q][2] = __builtin_elementwise_fma(f[q], __builtin_shufflevector(p[u].zw, p[u].zw, 0, 0), vq[q][2]);
                }
            }
        };
        static_assert(NPF % 9 == 0 && (NPF / 9) % 2 == 1, "15 batches of 9: 7 double steps + 1");
        fetch(pa, 0);
#pragma unroll 1
        for (int e0 = 0; e0 + 9 < NPF; e0 += 18) {
            fetch(pb, e0 + 9);
            __builtin_amdgcn_sched_barrier(0);
            consume(pa, e0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(pa, e0 + 18);
            __builtin_amdgcn_sched_barrier(0);
            consume(pb, e0 + 9);
            __builtin_amdgcn_sched_barrier(0);
        }
        consume(pa, NPF - 9);
        if (pose_off_ws) {          // (uniform) a stage that keeps the finger pose reuses P from its second iteration on
#pragma unroll
            for (int q = 0; q < HG / 2; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int h = lbs_group_hand<HG>(gx, gs, 2 * q + i);
                    if (h < N) {
                        float* dstp = pose_off_ws + ((size_t)h * NV + v) * 3;
                        dstp[0] = i ? vq[q][0].y : vq[q][0].x; dstp[1] = i ? vq[q][1].y : vq[q][1].x; dstp[2] = i ? vq[q][2].y : vq[q][2].x;
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < HG / 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) vq[q][c] = lbs_v2f{vsh_s[6 * q + 2 * c][tid], vsh_s[6 * q + 2 * c + 1][tid]} + vq[q][c];
    }
    // skinning weights: the vertex's (up to) four non-zero ones when the asset has no denser vertex (MANO's own weights), else all 16
    float w[NJ];
    float4 ws4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t js4 = 0u;
    if (m.sparse4) {
        ws4 = m.w4_w[v];
        js4 = m.w4_j[v];
    } else {
        const float4* w4 = reinterpret_cast<const float4*>(m.weights + v * NJ);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 x = w4[q]; w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w; }
    }
    int tip = -1;
    if (TWO_HAND) {
#pragma unroll
        for (int t = 0; t < IHMR_NUM_TIPS; ++t)
            if (v == m.tip_ids[t]) tip = t;
    }
    // skinning, one hand pair at a time: T = sum_j w_j A_j over all 16 joints without branches (a zero weight adds an
    // exact zero), the pair's matrices read from LDS as broadcast 16-byte rows, packed FMAs
#pragma unroll
    for (int q = 0; q < HG / 2; ++q) {
        lbs_v2f T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = lbs_v2f{0.f, 0.f};
        if (m.sparse4) {
            // the same sum without its zero terms (joint order kept; fma(0, a, T) == T): 4 instead of 16 joints, the rows
            // gathered from LDS per lane instead of broadcast
            const float wv[4] = {ws4.x, ws4.y, ws4.z, ws4.w};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const int j = (int)((js4 >> (8 * sI)) & 0xffu);
                const lbs_v2f wj = {wv[sI], wv[sI]};
                const float4* A4 = reinterpret_cast<const float4*>(&A_s[q][12 * j][0]);
#pragma unroll
                for (int e2 = 0; e2 < 6; ++e2) {
                    const float4 a = A4[e2];
                    T[2 * e2] = __builtin_elementwise_fma(wj, lbs_v2f{a.x, a.y}, T[2 * e2]);
                    T[2 * e2 + 1] = __builtin_elementwise_fma(wj, lbs_v2f{a.z, a.w}, T[2 * e2 + 1]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const lbs_v2f wj = {w[j], w[j]};
                const float4* A4 = reinterpret_cast<const float4*>(&A_s[q][12 * j][0]);   // {A[e] h0, A[e] h1, A[e+1] h0, A[e+1] h1}
#pragma unroll
                for (int e2 = 0; e2 < 6; ++e2) {
                    const float4 a = A4[e2];
                    T[2 * e2] = __builtin_elementwise_fma(wj, lbs_v2f{a.x, a.y}, T[2 * e2]);
                    T[2 * e2 + 1] = __builtin_elementwise_fma(wj, lbs_v2f{a.z, a.w}, T[2 * e2 + 1]);
                }
            }
        }
        lbs_v2f o2[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) o2[r] = T[4 * r + 0] * vq[q][0] + T[4 * r + 1] * vq[q][1] + T[4 * r + 2] * vq[q][2] + T[4 * r + 3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hh = 2 * q + i, h = lbs_group_hand<HG>(gx, gs, hh);
            if (h >= N) continue;
            float out[3] = {i ? o2[0].y : o2[0].x, i ? o2[1].y : o2[1].x, i ? o2[2].y : o2[2].x};
            if (!REUSE) {
                float* ws = v_posed_ws + ((size_t)h * NV + v) * 3;
                ws[0] = i ? vq[q][0].y : vq[q][0].x; ws[1] = i ? vq[q][1].y : vq[q][1].x; ws[2] = i ? vq[q][2].y : vq[q][2].x;
            }
            const bool left = TWO_HAND && h >= B;
            if (left) {  // optimize_model.py:210-211, 222-228
                out[0] = -out[0] + shift_s[hh][0];
                out[1] = out[1] + shift_s[hh][1];
                out[2] = out[2] + shift_s[hh][2];
            }
            float* dst = verts + ((size_t)h * NV + v) * 3;
            dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2];
            if (TWO_HAND && tip >= 0) {  // fingertip joints are vertices (:201-202)
                const int b = left ? h - B : h;
                float* jd = joints + ((size_t)b * 42 + (left ? 21 : 0) + NJ + tip) * 3;
                jd[0] = out[0]; jd[1] = out[1]; jd[2] = out[2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------- backward 1
// One workgroup per hand.  Inputs are gradients w.r.t. the forward OUTPUTS (final verts / joints).
// need_mask: bit0 orient, bit1 pose, bit2 betas, bit3 trans.
struct LbsBwdShared {
    __attribute__((aligned(16))) float sk[SK_STRIDE];
    float g[NV3];         // d L / d verts (raw hand frame)
    float vp[NV3];        // v_posed (saved by the forward)
    float dA[NJ][12];
    float dG[NJ][12];
    float dR[NJ][9];
    float drel[NJ][3];
    float dJ[NJ][3];
    float gsum[3];        // TWO_HAND: sum of the left-hand output gradients (= d L / d shift)
    float gj[21][3];      // joint gradients (raw hand frame)
    int par[NJ], dep[NJ];         // kinematic tree (parents, depth), staged once: the chain loops read them many times
    int nchild[NJ], child[NJ][NJ];  // children of every joint in index order (built in-kernel from par)
    float wsum[LBS_THREADS / WAVE][4];
};

// dynamic LDS: float part[nseg][12] -- per-segment partial sums of dA (nseg is a property of the weight matrix:
// 248 for 4 bones per vertex, up to LBS_SEG_CAP if dense); with it the workgroup needs ~38 KB: four fit a CU (1024 hands
// = one round of the 256 CUs); d v_posed goes through the workspace (L2) instead of LDS for that
// The backward of ONE hand by LBS_THREADS (256) threads tid = 0..255 (block-wide barriers inside: every thread of the workgroup
// calls it, with its own hand's LDS).  lbs_bwd1_kernel = one hand per workgroup; opt_tail_kernel (refine.h) = both hands of a sample.
template <bool TWO_HAND>
__device__ __forceinline__ void lbs_bwd1_hand(const ihmr_mano& m, const LbsWork& wk, int B, int h, int tid, LbsBwdShared& bw,
                                              float* bwd1_part /* [nseg][12], LDS */,
                                              const float* __restrict__ d_verts, const float* __restrict__ d_joints,
                                              float* __restrict__ d_orient, float* __restrict__ d_betas,
                                              float* __restrict__ d_trans, int need_mask, const LbsBwdShared* lds_left = nullptr) {
    // lds_left != nullptr (opt_tail_kernel): the caller has already put this hand's inputs into `bw` -- g and gj (raw hand frame) written
    // by the sampling / loss phase of the same workgroup, vp and sk by DMA -- and passes the LEFT hand's record for d L / d shift;
    // nothing is read back from global memory and the staging phase is skipped.  The same values in the same order: the same bits.
    const bool in_lds = TWO_HAND && lds_left != nullptr;
    const bool left = TWO_HAND && h >= B;
    const int b = TWO_HAND ? (left ? h - B : h) : 0;
    const bool need_orient = need_mask & 1, need_pose = need_mask & 2, need_betas = need_mask & 4, need_trans = need_mask & 8;

    // the kinematic tree, requested first (read several phases later)
    constexpr int VR = (NV + LBS_THREADS - 1) / LBS_THREADS;
    if (tid < NJ) { bw.par[tid] = m.parents[tid]; bw.dep[tid] = m.depth[tid]; }
    // every other global input of the workgroup in the same batch: the left hand's output gradients (for d L / d shift),
    // this hand's output gradients, v_posed and skeleton record, the joint gradients
    constexpr int NR = (NV3 + LBS_THREADS - 1) / LBS_THREADS, SR = (SK_STRIDE + LBS_THREADS - 1) / LBS_THREADS;
    float rg[NR], rv[NR], rs[SR], gl0[VR][3], gjl[3] = {0.f, 0.f, 0.f}, gjo = 0.f;
    if (in_lds) {
        // the left hand's raw output gradients from its LDS record (x stored negated: negated back, exactly)
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = tid + r * LBS_THREADS;
            gl0[r][0] = v < NV ? -lds_left->g[3 * v] : 0.f;
            gl0[r][1] = v < NV ? lds_left->g[3 * v + 1] : 0.f;
            gl0[r][2] = v < NV ? lds_left->g[3 * v + 2] : 0.f;
        }
        if (tid < 21) { gjl[0] = -lds_left->gj[tid][0]; gjl[1] = lds_left->gj[tid][1]; gjl[2] = lds_left->gj[tid][2]; }
    } else {
    if (TWO_HAND) {
        const float* gl = d_verts + ((size_t)(B + b) * NV) * 3;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = tid + r * LBS_THREADS;
#pragma unroll
            for (int k = 0; k < 3; ++k) gl0[r][k] = v < NV ? gl[3 * v + k] : 0.f;
        }
        if (tid < 21) {
            const float* gj = d_joints + ((size_t)b * 42 + 21 + tid) * 3;
            gjl[0] = gj[0]; gjl[1] = gj[1]; gjl[2] = gj[2];
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = min(tid + r * LBS_THREADS, NV3 - 1);
        rg[r] = d_verts[(size_t)h * NV3 + i];
        rv[r] = wk.v_posed[(size_t)h * NV3 + i];
    }
#pragma unroll
    for (int r = 0; r < SR; ++r) rs[r] = wk.skel[(size_t)h * SK_STRIDE + min(tid + r * LBS_THREADS, SK_STRIDE - 1)];
    if (tid < 21 * 3) {
        const int j = tid / 3, k = tid % 3;
        if (TWO_HAND) gjo = d_joints[((size_t)b * 42 + (left ? 21 : 0) + j) * 3 + k];
        else gjo = j < NJ ? d_joints[((size_t)h * NJ + j) * 3 + k] : 0.f;
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- TWO_HAND: d L / d shift = sum over the LEFT hand's vertex and joint gradients of this sample
    if (TWO_HAND) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < VR; ++r) { s0 += gl0[r][0]; s1 += gl0[r][1]; s2 += gl0[r][2]; }
        s0 += gjl[0]; s1 += gjl[1]; s2 += gjl[2];
        // fixed-order block sum: DPP inside each wave, then the 4 wave totals in index order
        s0 = wave_reduce_sum_dpp(s0); s1 = wave_reduce_sum_dpp(s1); s2 = wave_reduce_sum_dpp(s2);
        if (tid % WAVE == 0) { bw.wsum[tid / WAVE][0] = s0; bw.wsum[tid / WAVE][1] = s1; bw.wsum[tid / WAVE][2] = s2; }
        __syncthreads();
        if (tid < 3) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < LBS_THREADS / WAVE; ++wv) t += bw.wsum[wv][tid];
            bw.gsum[tid] = t;
            if (left && need_trans) d_trans[b * 3 + tid] = t;
        }
        if ((need_mask & 7) == 0) return;  // stage 0: only the translation moves
    }

    // ---- skeleton record of this iteration's forward, output gradients into the raw hand frame
    if (!in_lds) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = tid + r * LBS_THREADS;
        if (i < NV3) {
            bw.g[i] = (left && (i % 3) == 0) ? -rg[r] : rg[r];
            bw.vp[i] = rv[r];
        }
    }
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int i = tid + r * LBS_THREADS;
        if (i < SK_STRIDE) bw.sk[i] = rs[r];
    }
    if (tid < 21 * 3) {
        const int j = tid / 3, k = tid % 3;
        bw.gj[j][k] = (TWO_HAND && left && k == 0) ? -gjo : gjo;
    }
    }
    __syncthreads();
    if (TWO_HAND && tid < IHMR_NUM_TIPS * 3) {  // fingertip joints are vertices
        const int t = tid / 3, k = tid % 3;
        bw.g[3 * m.tip_ids[t] + k] += bw.gj[NJ + t][k];
    }
    __syncthreads();
    const float* sR = bw.sk + SK_R; const float* sJ = bw.sk + SK_J; const float* sG = bw.sk + SK_G; const float* sA = bw.sk + SK_A;
    if (tid >= LBS_THREADS - NJ) {   // last wave, off the critical path: child lists for the chain gathers (par[] is visible since the barriers above)
        const int p = tid - (LBS_THREADS - NJ);
        int n = 0;
        for (int j = p + 1; j < NJ; ++j)
            if (bw.par[j] == p) bw.child[p][n++] = j;
        bw.nchild[p] = n;
    }

    // ---- orientation stage (only the root rotation moves): every vertex and posed joint is R0 q + J0 with q independent of R0, so
    //      d L / d R0 = [sum_v g_v (x) (v - J0) + sum_j gj_j (x) (G_j.t - J0)] R0 -- one 3 x 3 reduction over the re-skinned vertices instead
    //      of the per-joint segmented reduction and the chain backward by tree level (round 4; ~14 -> ~4 us of opt_tail_kernel in that stage).
    //      Fixed summation order (lane partial sums, DPP wave sums, waves in index order, joints in index order).
    if ((need_mask & 7) == 1 && m.sparse4) {
        float4 wr[VR];
        uint32_t jr[VR];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = min(tid + r * LBS_THREADS, NV - 1);
            wr[r] = m.w4_w[v];
            jr[r] = m.w4_j[v];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float J0[3] = {sJ[0], sJ[1], sJ[2]};
        float acc[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = tid + r * LBS_THREADS;
            if (v >= NV) break;
            const float wv[4] = {wr[r].x, wr[r].y, wr[r].z, wr[r].w};
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
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
            const float p0 = bw.vp[3 * v], p1 = bw.vp[3 * v + 1], p2 = bw.vp[3 * v + 2];
            const float g[3] = {bw.g[3 * v], bw.g[3 * v + 1], bw.g[3 * v + 2]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = (T[4 * c] * p0 + T[4 * c + 1] * p1 + T[4 * c + 2] * p2 + T[4 * c + 3]) - J0[c];
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[3 * q + c] = __builtin_fmaf(g[q], d, acc[3 * q + c]);
            }
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[e] = wave_reduce_sum_dpp(acc[e]);
        if (tid % WAVE == 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e) bw.dA[tid / WAVE][e] = acc[e];       // (dA is free in this stage: scratch for the wave sums)
        }
        __syncthreads();
        if (tid < 9) {
            const int q = tid / 3, c = tid % 3;
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < LBS_THREADS / WAVE; ++wv) t += bw.dA[wv][tid];
            for (int j = 0; j < NJ; ++j) t = __builtin_fmaf(bw.gj[j][q], sG[12 * j + 4 * c + 3] - J0[c], t);
            bw.dG[0][tid] = t;                                               // M[q][c]
        }
        __syncthreads();
        if (tid == 0) {
            float dR0[9], dr[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    dR0[3 * q + c] = bw.dG[0][3 * q] * sR[c] + bw.dG[0][3 * q + 1] * sR[3 + c] + bw.dG[0][3 * q + 2] * sR[6 + c];
            rodrigues_bwd(bw.sk + SK_POSE, dR0, dr);
            if (left) { dr[1] = -dr[1]; dr[2] = -dr[2]; }
            d_orient[h * 3 + 0] = dr[0]; d_orient[h * 3 + 1] = dr[1]; d_orient[h * 3 + 2] = dr[2];
        }
        return;
    }

    // ---- per vertex: d v_posed = T.R^T g, T.R = sum_j w_j A_j.R over all 16 joints (no branches: a zero weight adds
    //      an exact zero), the matrices read from LDS as broadcast rows.  Only the finger-pose and shape gradients need it.
    if (need_pose || need_betas) {
    if (m.sparse4) {
        // the vertex's (up to) four non-zero weights (see lbs_skin_kernel): the same sums without their zero terms
        float4 wr[VR];
        uint32_t jr[VR];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = min(tid + r * LBS_THREADS, NV - 1);
            wr[r] = m.w4_w[v];
            jr[r] = m.w4_j[v];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const int v = tid + r * LBS_THREADS;
            if (v >= NV) break;
            const float wv[4] = {wr[r].x, wr[r].y, wr[r].z, wr[r].w};
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float4* A4 = reinterpret_cast<const float4*>(sA + 12 * (int)((jr[r] >> (8 * sI)) & 0xffu));
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 a = A4[q];
                    T[4 * q] = __builtin_fmaf(wv[sI], a.x, T[4 * q]);
                    T[4 * q + 1] = __builtin_fmaf(wv[sI], a.y, T[4 * q + 1]);
                    T[4 * q + 2] = __builtin_fmaf(wv[sI], a.z, T[4 * q + 2]);
                }
            }
            const float g0 = bw.g[3 * v], g1 = bw.g[3 * v + 1], g2 = bw.g[3 * v + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) wk.dvp[(size_t)h * NV3 + 3 * v + c] = T[c] * g0 + T[4 + c] * g1 + T[8 + c] * g2;
        }
    } else {
    // the skinning weights of this thread's (up to 4) vertices, all 16 loads in flight together (L2 hits; the other three
    // resident workgroups of the CU cover the round trip)
    float4 wreg[VR][4];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int v = min(tid + r * LBS_THREADS, NV - 1);
        const float4* w4 = reinterpret_cast<const float4*>(m.weights + v * NJ);
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[r][q] = w4[q];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < VR; ++r) {
        const int v = tid + r * LBS_THREADS;
        if (v >= NV) break;
        const float w[NJ] = {wreg[r][0].x, wreg[r][0].y, wreg[r][0].z, wreg[r][0].w, wreg[r][1].x, wreg[r][1].y, wreg[r][1].z, wreg[r][1].w,
                             wreg[r][2].x, wreg[r][2].y, wreg[r][2].z, wreg[r][2].w, wreg[r][3].x, wreg[r][3].y, wreg[r][3].z, wreg[r][3].w};
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll 4
        for (int j = 0; j < NJ; ++j) {   // 4 joints' rows in flight: unrolled further the LDS reads alone take ~190 VGPRs
            const float4* A4 = reinterpret_cast<const float4*>(sA + 12 * j);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 a = A4[q];
                T[4 * q] = __builtin_fmaf(w[j], a.x, T[4 * q]);
                T[4 * q + 1] = __builtin_fmaf(w[j], a.y, T[4 * q + 1]);
                T[4 * q + 2] = __builtin_fmaf(w[j], a.z, T[4 * q + 2]);
            }
        }
        const float g0 = bw.g[3 * v], g1 = bw.g[3 * v + 1], g2 = bw.g[3 * v + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wk.dvp[(size_t)h * NV3 + 3 * v + c] = T[c] * g0 + T[4 + c] * g1 + T[8 + c] * g2;
        }
    }
    }
    }
    // ---- dA[j][e] = sum_v W[v][j] * [g (x) v_posed | g][e].  The CSR-by-joint list is cut into single-joint
    //      segments of <= 13 entries, one lane each (balanced: the wrist alone owns ~600 entries), then the
    //      segment partials of a joint are summed in index order -- fixed order, bit-reproducible.
    for (int sg = tid; sg < m.nseg; sg += LBS_THREADS) {
        const int q0 = m.seg_q[sg];
        int q1 = m.seg_q[sg + 1];
        // a joint's last segment stops at the joint's end (the next segment belongs to the next joint)
        q1 = min(q1, q0 + LBS_SEG);
        int vv[LBS_SEG];
        float ww[LBS_SEG];
#pragma unroll
        for (int u = 0; u < LBS_SEG; ++u) {
            const int qq = min(q0 + u, q1 - 1);
            vv[u] = m.wj_vert[qq];
            ww[u] = q0 + u < q1 ? m.wj_w[qq] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        float acc[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = 0.f;
#pragma unroll
        for (int u = 0; u < LBS_SEG; ++u) {
            const int v = vv[u];
            const float g0 = bw.g[3 * v], g1 = bw.g[3 * v + 1], g2 = bw.g[3 * v + 2];
            const float p0 = bw.vp[3 * v], p1 = bw.vp[3 * v + 1], p2 = bw.vp[3 * v + 2];
            const float w = ww[u];
            const float wg[3] = {w * g0, w * g1, w * g2};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[4 * r + 0] = __builtin_fmaf(wg[r], p0, acc[4 * r + 0]);
                acc[4 * r + 1] = __builtin_fmaf(wg[r], p1, acc[4 * r + 1]);
                acc[4 * r + 2] = __builtin_fmaf(wg[r], p2, acc[4 * r + 2]);
                acc[4 * r + 3] += wg[r];
            }
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) bwd1_part[sg * 12 + e] = acc[e];
    }
    __syncthreads();
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid % 12;
        float acc = 0.f;
        const int s1 = m.jseg_start[j + 1];
        for (int sg = m.jseg_start[j]; sg < s1; ++sg) acc += bwd1_part[sg * 12 + e];
        bw.dA[j][e] = acc;
    }
    __syncthreads();

    // ---- chain backward by tree level
    // dG_j = [dA_j.R - dA_j.t (x) J_j | dA_j.t + d posed_joint_j];  dJ_j(direct) = -G_j.R^T dA_j.t
    if (tid < NJ * 12) {
        const int j = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
        const float dat = bw.dA[j][4 * r + 3];
        bw.dG[j][e] = c < 3 ? bw.dA[j][e] - dat * sJ[3 * j + c] : dat + bw.gj[j][r];
    }
    if (tid >= 192 && tid < 192 + NJ * 3) {
        const int j = (tid - 192) / 3, c = (tid - 192) % 3;
        const float* G = sG + 12 * j;
        bw.dJ[j][c] = -(G[c] * bw.dA[j][3] + G[4 + c] * bw.dA[j][7] + G[8 + c] * bw.dA[j][11]);
    }
    __syncthreads();
    for (int d = m.max_depth; d >= 1; --d) {
        // children at depth d: dR_j = Gp.R^T dG_j.R ; drel_j = Gp.R^T dG_j.t
        if (tid < NJ * 12) {
            const int j = tid / 12, e = tid % 12;
            if (bw.dep[j] == d) {
                const float* Gp = sG + 12 * bw.par[j];
                const float* dGj = bw.dG[j];
                if (e < 9) {
                    const int c = e / 3, c2 = e % 3;
                    bw.dR[j][e] = Gp[c] * dGj[c2] + Gp[4 + c] * dGj[4 + c2] + Gp[8 + c] * dGj[8 + c2];
                } else {
                    const int c = e - 9;
                    bw.drel[j][c] = Gp[c] * dGj[3] + Gp[4 + c] * dGj[7] + Gp[8 + c] * dGj[11];
                }
            }
        }
        __syncthreads();
        // parents gather from their children at depth d (index order): dGp.R += dG_j.R R_j^T + dG_j.t (x) rel_j ; dGp.t += dG_j.t
        if (tid < NJ * 12) {
            const int p = tid / 12, e = tid % 12, r = e / 4, c = e % 4;
            float acc = 0.f, accJ = 0.f;
            const int nc = bw.dep[p] + 1 == d ? bw.nchild[p] : 0;   // a joint's children all sit one level below it
            for (int q = 0; q < nc; ++q) {
                const int j = bw.child[p][q];
                const float* dGj = bw.dG[j];
                if (c < 3) {
                    const float* Rj = sR + 9 * j;
                    const float rel = sJ[3 * j + c] - sJ[3 * p + c];
                    acc += dGj[4 * r] * Rj[3 * c] + dGj[4 * r + 1] * Rj[3 * c + 1] + dGj[4 * r + 2] * Rj[3 * c + 2] + dGj[4 * r + 3] * rel;
                } else {
                    acc += dGj[4 * r + 3];
                    accJ += bw.drel[j][r];
                }
            }
            // only joints that gathered something touch their rows: the threads below update dJ of the depth-d joints
            // in this same phase, and an unconditional "-= 0" here would race with them (lost update)
            if (nc > 0) {
                bw.dG[p][e] += acc;
                if (c == 3) bw.dJ[p][r] -= accJ;
            }
        }
        if (tid >= 192 && tid < 192 + NJ * 3) {
            const int j = (tid - 192) / 3, c = (tid - 192) % 3;
            if (bw.dep[j] == d) bw.dJ[j][c] += bw.drel[j][c];
        }
        __syncthreads();
    }
    if (tid < 9) bw.dR[0][tid] = bw.dG[0][4 * (tid / 3) + (tid % 3)];
    if (tid >= 64 && tid < 67) bw.dJ[0][tid - 64] += bw.dG[0][4 * (tid - 64) + 3];
    __syncthreads();

    // ---- global orientation gradient (the root rotation is not part of the pose feature)
    if (need_orient && tid == 0) {
        float dr[3];
        rodrigues_bwd(bw.sk + SK_POSE, bw.dR[0], dr);
        if (left) { dr[1] = -dr[1]; dr[2] = -dr[2]; }
        d_orient[h * 3 + 0] = dr[0]; d_orient[h * 3 + 1] = dr[1]; d_orient[h * 3 + 2] = dr[2];
    }
    // ---- finger-pose stage: hand the chain part of dR to bwd3
    if (need_pose && tid < NJ * 9) wk.chain[(size_t)h * 192 + tid] = bw.dR[tid / 9][tid % 9];

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
        // d v_posed of this hand comes back from the workspace (written by this workgroup's per-vertex phase several
        // barriers ago: visible to the whole workgroup); rows and gradients are fetched in two batches of 7 / 6 vertices
        // per lane so that the loads of a batch are in flight together without exceeding the 128-register budget
        const float* dvp_h = wk.dvp + (size_t)h * NV3;
        for (int l = wave; l < 10; l += LBS_THREADS / WAVE) {
            float acc = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int T0[2] = {0, 7}, T1[2] = {7, NVP / WAVE};
                float4 srow[7];
                float dv[7][3];
#pragma unroll
                for (int t = T0[half]; t < T1[half]; ++t) {
                    const int v = min(lane + WAVE * t, NV - 1);
                    srow[t - T0[half]] = m.sd4[l * NVP + lane + WAVE * t];  // padding rows are zero
                    dv[t - T0[half]][0] = dvp_h[3 * v]; dv[t - T0[half]][1] = dvp_h[3 * v + 1]; dv[t - T0[half]][2] = dvp_h[3 * v + 2];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = T0[half]; t < T1[half]; ++t) {
                    if (lane + WAVE * t < NV) {
                        acc = __builtin_fmaf(srow[t - T0[half]].x, dv[t - T0[half]][0], acc);
                        acc = __builtin_fmaf(srow[t - T0[half]].y, dv[t - T0[half]][1], acc);
                        acc = __builtin_fmaf(srow[t - T0[half]].z, dv[t - T0[half]][2], acc);
                    }
                }
            }
            if (lane < 48) acc = __builtin_fmaf(m.J_shapedirs[lane * 10 + l], bw.dJ[lane / 3][lane % 3], acc);
            acc = wave_reduce_sum(acc);
            if (lane == 0) d_betas[h * 10 + l] = acc;
        }
    }
}

template <bool TWO_HAND>
__global__ __launch_bounds__(LBS_THREADS, 4) void lbs_bwd1_kernel(ihmr_mano m, LbsWork wk, int B,
                                                               const float* __restrict__ d_verts,
                                                               const float* __restrict__ d_joints,
                                                               float* __restrict__ d_orient, float* __restrict__ d_betas,
                                                               float* __restrict__ d_trans, int need_mask) {
    __shared__ LbsBwdShared bw;
    extern __shared__ __attribute__((aligned(16))) float bwd1_part[];   // [nseg][12]
    lbs_bwd1_hand<TWO_HAND>(m, wk, B, (int)blockIdx.x, (int)threadIdx.x, bw, bwd1_part, d_verts, d_joints, d_orient, d_betas, d_trans, need_mask);
}

// ------------------------------------------------------------------------------------- backward 2
// d pose_feature = posedirs (135 x 2334) . d v_posed^T (2334 x N): a GEMM, on the matrix cores.
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate = a k-ordered fmaf chain): M = pose features (5 tiles of 32),
// N = hands (tiles of 32), K = vertex coordinates, split into LBS_KG groups of LBS_KC 32-column chunks whose
// partial sums part[kg][hand][e] are added in index order by bwd3 (bit-reproducible, no atomics).
// grid = (5, ceil(N/32), LBS_KG), block = one wave (four waves per workgroup, one K group each -- 1000 workgroups instead of 4000 --
// changes nothing: 27.1 against 26.0 us per 1024 hands; the kernel is bound by address processing, not by workgroup dispatch).
// Lane l feeds row/column l % 32; the K index of a chunk is
// permuted so that lane half l / 32 owns 16 CONSECUTIVE columns (k = k0 + 16 (l/32) + s at MFMA step s): every
// lane then streams 64 contiguous bytes of its basis row / its hand's gradient row, no LDS staging.
typedef float lbs_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void lbs_bwd2_kernel(ihmr_mano m, LbsWork wk, int N) {
    TL_SCOPE(8);
    const int lane = threadIdx.x, l31 = lane & 31, kp = lane >> 5;
    const int e = blockIdx.x * 32 + l31, hand = blockIdx.y * 32 + l31, kg = blockIdx.z;
    const float* arow = m.posedirs + (size_t)min(e, NPF - 1) * NV3;
    const float* brow = wk.dvp + (size_t)min(hand, N - 1) * NV3;
    const bool a_ok = e < NPF, b_ok = hand < N;
    // 16-byte loads (rows are 8-byte aligned only: NV3 * 4 = 9336; the hardware takes dword-aligned global accesses of any width):
    // half as many address-processing slots as float pairs -- the kernel is bound by those, every lane reads its own cache line
    typedef float lbs_f4u __attribute__((ext_vector_type(4), aligned(8)));
    lbs_f4u av[LBS_KC][4], bv[LBS_KC][4];
#pragma unroll
    for (int c = 0; c < LBS_KC; ++c) {
        const int k0 = (kg * LBS_KC + c) * 32 + 16 * kp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = min(k0 + 4 * q, NV3 - 4);          // (a clamped load only feeds elements that are masked below)
            av[c][q] = *reinterpret_cast<const lbs_f4u*>(arow + k);
            bv[c][q] = *reinterpret_cast<const lbs_f4u*>(brow + k);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    lbs_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < LBS_KC; ++c) {
        const int k0 = (kg * LBS_KC + c) * 32 + 16 * kp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // NV3 % 4 == 2: the last group of four of a row is half valid; it is loaded from NV3 - 4 (shifted by two), so its
            // valid elements k0 + 4q, k0 + 4q + 1 sit in .z, .w
            const int kq = k0 + 4 * q;
            const bool sh = kq + 4 > NV3 && kq < NV3;
            const float ax[4] = {sh ? av[c][q].z : av[c][q].x, sh ? av[c][q].w : av[c][q].y, av[c][q].z, av[c][q].w};
            const float bx[4] = {sh ? bv[c][q].z : bv[c][q].x, sh ? bv[c][q].w : bv[c][q].y, bv[c][q].z, bv[c][q].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool k_ok = kq + t < NV3;
                const float a0 = (a_ok && k_ok) ? ax[t] : 0.f, b0 = (b_ok && k_ok) ? bx[t] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            }
        }
    }
    // C/D layout: column (hand) = lane & 31, row (feature) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (b_ok) {
        float* dst = wk.dpf_part + ((size_t)kg * N + hand) * 136 + blockIdx.x * 32 + 4 * kp;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int e4 = blockIdx.x * 32 + 4 * kp + 8 * r4;
            if (e4 + 3 < 136) *reinterpret_cast<float4*>(dst + 8 * r4) = make_float4(acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]);
        }
    }
}

// The same GEMM with its operands staged through LDS (round 3).  A workgroup = five waves, one per 32-feature tile, x 64 hands, for one K
// group: the 32 x (160 + 64) operand block of a 32-column chunk is loaded with the lanes ALONG k (8 lanes = 128 contiguous bytes of a
// row; the streaming form above has every lane on a row of its own, i.e. 64 cache lines per load instruction -- bound by address
// processing) and stored row-major with a row stride of 33 floats, so that the MFMA operand reads (lanes along the rows) hit 32
// different banks; chunks double-buffered.  Same partial layout, same k -> MFMA step assignment (lane half l / 32 owns 16 consecutive
// columns): the same bits as the streaming form.  grid = (ceil(N/64), LBS_KG), block = 320.
#ifndef LBS_B2_MIN_HANDS
#define LBS_B2_MIN_HANDS 256
#endif
#define LBS_B2_LDK 33
#define LBS_B2_ROWS (160 + 64)
__global__ __launch_bounds__(320) void lbs_bwd2_lds_kernel(ihmr_mano m, LbsWork wk, int N) {
    TL_SCOPE(8);
    __shared__ float tile[2][LBS_B2_ROWS][LBS_B2_LDK];
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE, l31 = lane & 31, kp = lane >> 5;
    const int h0 = blockIdx.x * 64, kg = blockIdx.y;
    // loader: unit u = tid + 320 i, i < 6 (1792 of 1920 slots used): row = u / 8 (0..159 features, 160..223 hands), 4 columns at 4 (u % 8)
    float v4[6][4];
    auto load_chunk = [&](int c) {
        const int k0 = (kg * LBS_KC + c) * 32;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int u = tid + 320 * i, row = u >> 3, k = k0 + 4 * (u & 7);
            v4[i][0] = v4[i][1] = v4[i][2] = v4[i][3] = 0.f;
            if (row >= LBS_B2_ROWS) continue;
            const bool is_a = row < 160;
            const int r = is_a ? row : h0 + (row - 160);
            if (is_a ? r >= NPF : r >= N) continue;
            const float* src = (is_a ? m.posedirs + (size_t)r * NV3 : wk.dvp + (size_t)r * NV3) + k;
            if (k + 3 < NV3) {
                typedef float lbs_f4u __attribute__((ext_vector_type(4), aligned(4)));
                const lbs_f4u x = *reinterpret_cast<const lbs_f4u*>(src);
                v4[i][0] = x.x; v4[i][1] = x.y; v4[i][2] = x.z; v4[i][3] = x.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v4[i][t] = k + t < NV3 ? src[t] : 0.f;
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int u = tid + 320 * i, row = u >> 3, kq = 4 * (u & 7);
            if (row >= LBS_B2_ROWS) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) tile[buf][row][kq + t] = v4[i][t];
        }
    };
    lbs_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < LBS_KC; ++c) {
        const int cur = c & 1;
        if (c + 1 < LBS_KC) load_chunk(c + 1);               // in flight during the MFMAs below
        const float* arow = &tile[cur][wave * 32 + l31][16 * kp];
        const float* b0row = &tile[cur][160 + l31][16 * kp];
        const float* b1row = &tile[cur][160 + 32 + l31][16 * kp];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const float a = arow[st], b0 = b0row[st], b1 = b1row[st];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        if (c + 1 < LBS_KC) store_chunk(cur ^ 1);            // the other buffer: its readers finished before the previous barrier
        __syncthreads();
    }
    // C/D layout: column (hand) = lane & 31, row (feature) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int hand = h0 + 32 * half + l31;
        if (hand >= N) continue;
        const lbs_f32x16& acc = half ? acc1 : acc0;
        float* dst = wk.dpf_part + ((size_t)kg * N + hand) * 136 + wave * 32 + 4 * kp;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int e4 = wave * 32 + 4 * kp + 8 * r4;
            if (e4 + 3 < 136) *reinterpret_cast<float4*>(dst + 8 * r4) = make_float4(acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------- backward 3
// finger-pose gradients: dR_j = chain part + pose-feature part (K-group sums in fixed order), through Rodrigues.
// grid = N, block = 64: the wave first reduces the LBS_KG partial rows (coalesced, all loads in flight at once),
// then lanes 1..15 = joints.
// The work of one hand h by 64 threads j = 0..63 (a workgroup barrier inside: every thread of the workgroup calls it, threads that own
// no hand with active = false); dpf = 192 floats of LDS of this hand.  (Rounds 5 and 6 each tried running it inside the next iteration's
// opt_adam_skel_kernel, one launch fewer in the finger-pose stage: -1.2 % throughput with three sequences in flight, and 87.8 -> 87.3 us
// per iteration at one batch of 64 -- the chain it adds in front of the step is as long as the launch it removes.  docs/experiments.md)
template <bool TWO_HAND>
__device__ __forceinline__ void lbs_bwd3_hand(const LbsWork& wk, int N, int B, float* __restrict__ d_pose, int h, int j, float* dpf, bool active = true) {
    float r[3] = {0.f, 0.f, 0.f}, chain[9];
    const int jj = min(max(j, 1), NJ - 1);
    if (active) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = j + 64 * t;
            if (e >= 136) break;
            float part[LBS_KG];
#pragma unroll
            for (int c = 0; c < LBS_KG; ++c) part[c] = wk.dpf_part[((size_t)c * N + h) * 136 + e];
            __builtin_amdgcn_sched_barrier(0);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < LBS_KG; ++c) acc += part[c];
            dpf[e] = acc;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] = wk.skel[(size_t)h * SK_STRIDE + SK_POSE + 3 * jj + k];
#pragma unroll
        for (int e = 0; e < 9; ++e) chain[e] = wk.chain[(size_t)h * 192 + 9 * jj + e];
    }
    __syncthreads();
    if (!active || j < 1 || j >= NJ) return;
    const bool left = TWO_HAND && h >= B;
    float dR[9], dr[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) dR[e] = chain[e] + dpf[(j - 1) * 9 + e];
    rodrigues_bwd(r, dR, dr);
    if (left) { dr[1] = -dr[1]; dr[2] = -dr[2]; }
    float* dst = d_pose + (size_t)h * 45 + 3 * (j - 1);
    dst[0] = dr[0]; dst[1] = dr[1]; dst[2] = dr[2];
}
template <bool TWO_HAND>
__global__ __launch_bounds__(64) void lbs_bwd3_kernel(LbsWork wk, int N, int B, float* __restrict__ d_pose) {
    TL_SCOPE(9);
    __shared__ float dpf[192];
    lbs_bwd3_hand<TWO_HAND>(wk, N, B, d_pose, (int)blockIdx.x, (int)threadIdx.x, dpf);
}
