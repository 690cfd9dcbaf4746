// Image encoder kernels: fp32 convolution as implicit GEMM on the CDNA4 matrix cores, NHWC activations.
//
// Stands behind the reference's `InterHandEncoder.forward` (models/networks.py:66-80) and `ResNet.forward`
// (models/resnet.py:138-156; Bottleneck :58-94): every Conv2d + BatchNorm2d (+ ReLU, + residual add) of the
// ResNet-50 trunk, the Linear layers (fc1, feat_encoder, the 3 IEF iterations of regressor_ih, hand_classifier)
// and the two pooling layers.  BatchNorm (eval mode, running statistics) is folded into the conv weights and a
// per-channel bias on the host; ReLU / residual / sigmoid are fused into the GEMM epilogue.
//
// GEMM view: Y[M = N*Ho*Wo][Cout] = A[M][K = kh*kw*Cin] . Wt[K][Cout], A gathered on the fly (never stored).
// MFMA: v_mfma_f32_32x32x2_f32 -- f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s
// peak on MI355X; the reference is fp32 end to end, so no reduced-precision path is needed for parity.
// Tile: BM x BN x 16 per workgroup (BM = 64 | 128, BN = 64 | 128), one wave per 64 x 32 sub-tile (2 to 8 waves);
// A is staged K-major in LDS so that the 32 lanes of an MFMA row read consecutive addresses; LDS is
// double-buffered: the global loads of tile t+1 are issued before the MFMAs of tile t and stored to the other
// buffer after them, ONE barrier per K step.  The host picks the largest tile that still gives >= 1.5 workgroups
// per CU (the 14x14 / 7x7 layers of a 64-image batch have only 12544 / 3136 rows).
#pragma once
#include <type_traits>
#include "ihmr_common.h"

#define CONV_BK 16

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;         // input, NHWC with pixel stride ldx
    const float* w;         // [Kpad][ldw] (K-major, BN folded in), Kpad = ceil16(kh*kw*Cin), zero padded
    const float* bias;      // [Cout] (folded BN shift or Linear bias)
    const float* residual;  // optional [M][ldr]
    float* y;               // [M][ldy]
    int N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad;
    int ldx, ldw, ldy, ldr;
    int act;                // 0 none, 1 relu, 2 sigmoid
    float* partial;         // split-K: [ksplit][M][Cout] raw partial sums (bias / residual / activation applied by conv_splitk_reduce_kernel)
    int ksplit;             // gridDim.z; 1 = no split
};

// Epilogue of one wave's 64 x 32 sub-tile (C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
// y = act(acc + bias (+ residual)).  Written for few vector-ALU instructions (they are not hidden behind other waves' MFMAs, and a layer with
// 4-16 K steps has only 64-256 MFMAs per wave to set them against): the activation and the presence of a residual are template parameters
// (rounds 1-3 branched on them per element), a row's address is the lane's base pointer + a wave-uniform (scalar) multiple of the row
// stride, and the per-row bounds test exists only in the EDGE instance (the layer's last M tile).
template <int ACT, bool RES, bool EDGE>
__device__ __forceinline__ void conv_store_subtile(const ConvArgs& a, const f32x16 (&acc)[2], int mw, int n, int M, float bv) {
    float* yp = a.y + (size_t)mw * a.ldy + n;              // mw = first row of this lane in the sub-tile (wave's row block + 4 (lane >> 5))
    const float* rp = RES ? a.residual + (size_t)mw * a.ldr + n : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        // residual rows first, the loads of one 32-row half in flight together; one half at a time keeps 16 instead of 32 registers live
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = mi * 32 + (r & 3) + 8 * (r >> 2);
            res[r] = (RES && (!EDGE || mw + dr < M)) ? rp[dr * a.ldr] : 0.f;    // (32-bit scalar product: one VALU instruction per address)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = mi * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[mi][r] + bv;
            if (RES) v += res[r];
            if (ACT == 1) v = fmaxf(v, 0.f);
            if (!EDGE || mw + dr < M) yp[dr * a.ldy] = v;
        }
    }
}

// bias / residual / activation dispatch of a sub-tile; tile_end = first row after the workgroup's tile (wave-uniform)
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[2], int mw, int n, int M, int tile_end) {
    const float bv = a.bias ? a.bias[n] : 0.f;
    const bool edge = tile_end > M;
    if (a.act == 2) {                                      // sigmoid: the hand classifier's single tile; generic form
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + mi * 32 + (r & 3) + 8 * (r >> 2);
                if (m >= M) continue;
                const float v = acc[mi][r] + bv + (a.residual ? a.residual[(size_t)m * a.ldr + n] : 0.f);
                a.y[(size_t)m * a.ldy + n] = 1.0f / (1.0f + expf(-v));
            }
        return;
    }
    const int sel = (a.act == 1 ? 4 : 0) | (a.residual ? 2 : 0) | (edge ? 1 : 0);
    switch (sel) {
        case 0: conv_store_subtile<0, false, false>(a, acc, mw, n, M, bv); break;
        case 1: conv_store_subtile<0, false, true>(a, acc, mw, n, M, bv); break;
        case 2: conv_store_subtile<0, true, false>(a, acc, mw, n, M, bv); break;
        case 3: conv_store_subtile<0, true, true>(a, acc, mw, n, M, bv); break;
        case 4: conv_store_subtile<1, false, false>(a, acc, mw, n, M, bv); break;
        case 5: conv_store_subtile<1, false, true>(a, acc, mw, n, M, bv); break;
        case 6: conv_store_subtile<1, true, false>(a, acc, mw, n, M, bv); break;
        default: conv_store_subtile<1, true, true>(a, acc, mw, n, M, bv); break;
    }
}

// Zero page: a padding pixel's loader reads from here instead of the image (one unconditional load per step, no select afterwards)
__device__ float g_conv_zero[2048 + 64];

// FAST: Cin % 16 == 0 (every layer but the stem), compiled without the generic gather so that its per-loader state does not
// cost registers or a branch per K step;
// and the 128 x 128 tile is held to 80 registers (three resident workgroups per CU instead of two; measured: 4 waves per SIMD
// 6.94 ms, 6 waves 6.86 ms, 8 waves -- with spills -- 7.07 ms per 64-image encoder pass; the runtime `fast` branch inside the
// K loop and the 32-register residual prefetch of the old epilogue cost 7.8 ms)
#ifndef CONV_WAVES_PER_EU
#define CONV_WAVES_PER_EU 6
#endif
#define CONV_GENERIC 0
#define CONV_FAST 1
#define CONV_C4 2
template <int BM, int BN, int MODE>                        // MODE: CONV_GENERIC, CONV_FAST (Cin % 16 == 0), CONV_C4 (Cin == 4: the padded stem)
__global__ __launch_bounds__((BM / 64) * (BN / 32) * 64)
__attribute__((amdgpu_waves_per_eu((MODE == CONV_FAST && BM == 128 && BN == 128) ? CONV_WAVES_PER_EU : 4)))
void conv_igemm_kernel(ConvArgs a) {
    constexpr bool FAST = MODE != CONV_GENERIC;             // float4 gathers, lean K loop
    constexpr bool C4 = MODE == CONV_C4;
    constexpr int WN_WAVES = BN / 32;                      // waves along n; each wave owns a 64 x 32 sub-tile
    constexpr int THREADS = (BM / 64) * WN_WAVES * 64;
    // LDS layouts.  FAST: A column c = 64 h + 2 j + b holds tile row 64 h + 32 b + j, so that a lane's two A operands (rows j and j + 32 of its
    // wave's 64-row block) are ONE 8-byte read; LDA = BM + 4 keeps the loaders' stores (16 consecutive columns x 4 k rows per wave) and
    // LDB = BN + 32 the B reads (32 consecutive columns x 2 k rows) free of bank conflicts.  Generic (stem): rows in order, scalar reads.
    constexpr int LDA = FAST ? BM + 4 : BM + 5, LDB = FAST ? BN + 32 : BN + 4;
    constexpr int A_F4 = BM * CONV_BK / 4 / THREADS;       // float4 loads of A per thread (1 or 2)
    constexpr int B_F4 = CONV_BK * BN / 4 / THREADS;       // float4 loads of B per thread (1 or 2)
    constexpr int B_F4_PER_ROW = BN / 4;
    __shared__ __attribute__((aligned(16))) float As[2][CONV_BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][CONV_BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int kl = lane >> 5, l31 = lane & 31;
    const int M = a.N * a.Ho * a.Wo, K = a.kh * a.kw * a.Cin, Kpad = (K + CONV_BK - 1) / CONV_BK * CONV_BK;
    // Tile of this workgroup.  Workgroups go to the XCDs round-robin in dispatch order (x fastest), each XCD with its own L2: XCD k = L % 8
    // takes the k-th CONTIGUOUS eighth of the tile sequence, walked with the column tiles of one row tile adjacent in time -- the row
    // tile's A operand (and the 3 x 3 halo it shares with its neighbours) is fetched into ONE L2 once, instead of once per column tile
    // and XCD.  (Placement is a performance assumption only: any map gives the same tiles, each computed exactly as before.)
#ifdef CONV_NO_XCD_MAP
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
#else
    int m0, n0;
    {
        const unsigned gy = gridDim.y, T = gridDim.x * gy, L = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned q = T >> 3, r = T & 7u, xcd = L & 7u, seq = xcd * q + min(xcd, r) + (L >> 3);
        m0 = (int)(seq / gy) * BM; n0 = (int)(seq % gy) * BN;
    }
#endif
    // split-K: this workgroup owns K tiles [kc0, kc1)
    const int nk_all = Kpad / CONV_BK, nk_per = (nk_all + a.ksplit - 1) / a.ksplit;
    const int kc0 = blockIdx.z * nk_per, kc1 = min(nk_all, kc0 + nk_per);

    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    if constexpr (FAST) {
        // CONV_FAST: Cin % 16 == 0 (every layer but the stem): a 16-wide K chunk never straddles a filter tap and the whole wave reads one tap.
        // CONV_C4: Cin == 4 (the image padded to 4 channels): a K chunk is 4 consecutive taps x 4 channels, a thread's float4 is ONE pixel of
        // tap 4 kc + (tid & 3) -- 16-byte loads instead of the generic path's scalar gather with a bounds test per element (stem 256 -> us).
        // The K loop follows conv_streamk_kernel
        // (below): as few vector-ALU instructions per MFMA as possible -- they are NOT hidden behind the MFMAs of the SIMD's other waves --
        // i.e. carried operand pointers (the gather's index arithmetic runs only when the tap changes, under a wave-uniform branch without
        // loads), padding pixels read from a zero page, two operand tiles in flight in registers, every load and LDS store unconditional.
        const int ak4 = (tid & 3) * 4;                     // THREADS % 4 == 0: the same k offset for every float4 of a thread
        int acol[A_F4], an[A_F4], hbase[A_F4], wbase[A_F4];
        bool am_ok[A_F4];
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            acol[i] = (tid + i * THREADS) >> 2;
            const int am = m0 + ((acol[i] & 64) | ((acol[i] & 1) << 5) | ((acol[i] >> 1) & 31));
            am_ok[i] = am < M;
            int aho = 0, awo = 0;
            an[i] = 0;
            if (am_ok[i]) { an[i] = am / (a.Ho * a.Wo); const int r = am % (a.Ho * a.Wo); aho = r / a.Wo; awo = r % a.Wo; }
            hbase[i] = aho * a.stride - a.pad; wbase[i] = awo * a.stride - a.pad;
        }
        // CONV_FAST: wave-uniform filter tap / channel offset of the NEXT tile to load.  CONV_C4: this THREAD's tap of the next tile
        // (tap_c unused), advanced by 4 taps per K step; taps past the last one (zero-padded K) read whatever pixel they land on or the
        // zero page -- their filter rows are zero.
        int tap_c, tap_h, tap_w;
        {
            const int k0 = kc0 * CONV_BK, tap = C4 ? k0 / 4 + (tid & 3) : k0 / a.Cin;
            tap_c = C4 ? 0 : k0 % a.Cin; tap_h = tap / a.kw; tap_w = tap % a.kw;
        }
        const float* pa[A_F4];
        auto retap = [&]() {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int hi = hbase[i] + tap_h, wi = wbase[i] + tap_w;
                const bool ok = am_ok[i] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                pa[i] = (ok ? a.x + ((size_t)(an[i] * a.H + hi) * a.W + wi) * a.ldx : g_conv_zero) + (C4 ? 0 : tap_c + ak4);
            }
        };
        retap();
        const float* pb[B_F4];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int g = tid + i * THREADS;
            pb[i] = a.w + (size_t)(kc0 * CONV_BK + g / B_F4_PER_ROW) * a.ldw + n0 + (g % B_F4_PER_ROW) * 4;   // ldw >= n0 + BN, zero padded
        }
        const size_t bstep = (size_t)CONV_BK * a.ldw;
        int kb = kc0;                                      // K tile pb points at, clamped to the last one (loads past the end are unused)
        float4 ar0[A_F4], ar1[A_F4], br0[B_F4], br1[B_F4];
        auto load2 = [&](auto set) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(pa[i]);
                if constexpr (S == 0) ar0[i] = v; else ar1[i] = v;
                if constexpr (!C4) pa[i] += CONV_BK;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(pb[i]);
                if constexpr (S == 0) br0[i] = v; else br1[i] = v;
                pb[i] += kb + 1 < nk_all ? bstep : 0;
            }
            kb += kb + 1 < nk_all ? 1 : 0;
            if constexpr (C4) {
                tap_w += 4;                                // kw >= 4 (7 x 7 stem): at most one wrap
                if (tap_w >= a.kw) { tap_w -= a.kw; ++tap_h; }
                retap();
            } else {
                tap_c += CONV_BK;
                if (tap_c >= a.Cin) { tap_c = 0; if (++tap_w == a.kw) { tap_w = 0; ++tap_h; } retap(); }
            }
        };
        auto store2 = [&](int buf, auto set) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const float4 v = S == 0 ? ar0[i] : ar1[i];
                As[buf][ak4 + 0][acol[i]] = v.x; As[buf][ak4 + 1][acol[i]] = v.y;
                As[buf][ak4 + 2][acol[i]] = v.z; As[buf][ak4 + 3][acol[i]] = v.w;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int g = tid + i * THREADS;
                *reinterpret_cast<float4*>(&Bs[buf][g / B_F4_PER_ROW][(g % B_F4_PER_ROW) * 4]) = S == 0 ? br0[i] : br1[i];
            }
        };
        auto mfma2 = [&](int cur) {
#pragma unroll
            for (int kk = 0; kk < CONV_BK; kk += 2) {
                const float2 av = *reinterpret_cast<const float2*>(&As[cur][kk + kl][wm * 64 + 2 * l31]);
                const float bf = Bs[cur][kk + kl][wn * 32 + l31];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf, acc[1], 0, 0, 0);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        if (kc0 < kc1) {
            load2(S0{});
            load2(S1{});
            store2(0, S0{});
        }
        __syncthreads();
        int kc = kc0;
        for (; kc + 1 < kc1; kc += 2) {
            load2(S0{}); mfma2(0); store2(1, S1{}); __syncthreads();
            load2(S1{}); mfma2(1); store2(0, S0{}); __syncthreads();
        }
        if (kc < kc1) { mfma2(0); __syncthreads(); }
    } else {
        // ---- generic gather (Cin not a multiple of 16, i.e. the 7x7 stem with Cin = 3): one tile in flight
        int arow[A_F4], ak4[A_F4], an[A_F4], aho[A_F4], awo[A_F4];
        bool am_ok[A_F4];
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * THREADS;               // float4 f -> (row f / 4, 4 consecutive k at (f % 4) * 4)
            arow[i] = f >> 2; ak4[i] = (f & 3) * 4;
            const int am = m0 + arow[i];
            am_ok[i] = am < M;
            an[i] = 0; aho[i] = 0; awo[i] = 0;
            if (am_ok[i]) { an[i] = am / (a.Ho * a.Wo); const int r = am % (a.Ho * a.Wo); aho[i] = r / a.Wo; awo[i] = r % a.Wo; }
        }
        float4 areg[A_F4], breg[B_F4];
        int gc[A_F4], gfw[A_F4], gfh[A_F4];                // (channel, tap column, tap row) of each loader's element 0
        const int gstep_c = CONV_BK % a.Cin, gstep_t = CONV_BK / a.Cin;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int k = kc0 * CONV_BK + ak4[i], tap = k / a.Cin;
            gc[i] = k % a.Cin; gfw[i] = tap % a.kw; gfh[i] = tap / a.kw;
        }
        auto load_tile = [&](int kc) {
            const int k0 = kc * CONV_BK;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                // element e of the float4 is k = k0 + ak4 + e; its (channel, tap column, tap row) is carried incrementally, no div / mod in the loop
                float v[4];
                int c = gc[i], fw = gfw[i], fh = gfh[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = 0.f;
                    if (am_ok[i] && fh < a.kh) {                 // fh >= kh: past the last tap (zero-padded K)
                        const int hi = aho[i] * a.stride + fh - a.pad, wi = awo[i] * a.stride + fw - a.pad;
                        if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) v[e] = a.x[((size_t)(an[i] * a.H + hi) * a.W + wi) * a.ldx + c];
                    }
                    if (++c == a.Cin) { c = 0; if (++fw == a.kw) { fw = 0; ++fh; } }
                }
                // advance this loader's element 0 by one K tile (16 elements)
                gc[i] += gstep_c;
                int tapinc = gstep_t;
                if (gc[i] >= a.Cin) { gc[i] -= a.Cin; ++tapinc; }
                gfw[i] += tapinc;
                while (gfw[i] >= a.kw) { gfw[i] -= a.kw; ++gfh[i]; }
                areg[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int g = tid + i * THREADS;
                const int k = k0 + g / B_F4_PER_ROW, n4 = (g % B_F4_PER_ROW) * 4;
                breg[i] = *reinterpret_cast<const float4*>(a.w + (size_t)k * a.ldw + n0 + n4);   // ldw >= n0 + BN, zero padded
            }
        };
        auto store_tile = [&](int buf) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                As[buf][ak4[i] + 0][arow[i]] = areg[i].x; As[buf][ak4[i] + 1][arow[i]] = areg[i].y;
                As[buf][ak4[i] + 2][arow[i]] = areg[i].z; As[buf][ak4[i] + 3][arow[i]] = areg[i].w;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int g = tid + i * THREADS;
                *reinterpret_cast<float4*>(&Bs[buf][g / B_F4_PER_ROW][(g % B_F4_PER_ROW) * 4]) = breg[i];
            }
        };
        if (kc0 < kc1) {
            load_tile(kc0);
            store_tile(0);
        }
        __syncthreads();
        for (int kc = kc0; kc < kc1; ++kc) {
            const int cur = (kc - kc0) & 1;
            if (kc + 1 < kc1) load_tile(kc + 1);   // in flight during the MFMAs below
#pragma unroll
            for (int kk = 0; kk < CONV_BK; kk += 2) {
                const float a0 = As[cur][kk + kl][wm * 64 + l31], a1 = As[cur][kk + kl][wm * 64 + 32 + l31], bf = Bs[cur][kk + kl][wn * 32 + l31];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf, acc[1], 0, 0, 0);
            }
            if (kc + 1 < kc1) store_tile(cur ^ 1); // the other buffer: its readers finished before the previous barrier
            __syncthreads();
        }
    }

    // ---- epilogue: bias (+ residual) (+ activation); C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int n = n0 + wn * 32 + l31, rbase = 4 * kl;
    if (n >= a.Cout) return;
    if (a.ksplit > 1) {   // raw partial sums; the reduce kernel finishes the layer
        const int mw = m0 + wm * 64 + rbase;
        float* part = a.partial + ((size_t)blockIdx.z * M + mw) * a.Cout + n;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = mi * 32 + (r & 3) + 8 * (r >> 2);
                if (mw + dr < M) part[dr * a.Cout] = acc[mi][r];
            }
        return;
    }
    conv_epilogue(a, acc, m0 + wm * 64 + rbase, n, M, m0 + BM);
}

// split-K epilogue: y = act(sum_z partial[z] (fixed order) + bias + residual); one thread per VEC output channels
template <int VEC>
__global__ void conv_splitk_reduce_kernel(ConvArgs a) {
    const int M = a.N * a.Ho * a.Wo, cv = a.Cout / VEC;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * cv) return;
    const int m = (int)(idx / cv), n = (int)(idx % cv) * VEC;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = a.partial[(size_t)m * a.Cout + n + e];
    for (int z = 1; z < a.ksplit; ++z)
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += a.partial[((size_t)z * M + m) * a.Cout + n + e];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        v[e] += a.bias ? a.bias[n + e] : 0.f;
        if (a.residual) v[e] += a.residual[(size_t)m * a.ldr + n + e];
        if (a.act == 1) v[e] = fmaxf(v[e], 0.f);
        else if (a.act == 2) v[e] = 1.0f / (1.0f + expf(-v[e]));
        a.y[(size_t)m * a.ldy + n + e] = v[e];
    }
}

// ---- Stream-K form of the 128 x 128 tile (fast path only) for the layers whose tile count does not fill the GPU evenly.
// ResNet-50 at batch 64 has M = 64 * 49 * 4^j output rows: 100, 196 or 392 tiles of 128 x 128 on 256 CUs, i.e. 0.8 or 1.5
// workgroups per CU with one tile per workgroup -- a quarter of the CUs' time is lost to the uneven last generation (and a 2-way
// K split only moves the problem: 392 or 784 workgroups).  Here a fixed number of workers (gridDim.x, two per CU) share the
// layer's tiles x K-steps evenly: worker w owns the steps [w * total / W, (w + 1) * total / W) of the sequence (tile 0: steps
// 0..nk-1, tile 1: ...; tiles ordered M-fastest), at most two of its tile segments are partial.  A tile a worker covers
// completely gets the ordinary epilogue; partial sums go to the worker's two 64 KB slots of the workspace (slot 1: the segment
// that begins a tile, slot 0: any other) and conv_streamk_fixup_kernel adds a tile's segments in ascending K order -- a fixed
// order, so the result does not depend on timing -- and applies bias / residual / activation.
// Phase stamps (experiment builds only, -DCONV_STAMPS; scripts/experiments/conv_stamps.py): shader-clock time thread 0 of every worker spends
// in each phase of conv_streamk_kernel: 0 segment prologue, 1 load issue, 2 LDS reads + MFMA issue, 3 wait for the older tile + LDS stores,
// 4 barrier, 5 epilogue / partial store; 6 = K steps, 7 = segments
#ifdef CONV_STAMPS
__device__ long long g_conv_stamps[1024][10];   // 8 = the worker's life in shader clocks, 9 = in 100 MHz wall-clock ticks
// (sums are kept in registers and written once at the end: a read-modify-write of global memory per stamp would drain the operand loads)
#define CONV_TK(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); tk_acc_[k] += now_ - tk_prev_; tk_prev_ = now_; } while (0)
#define CONV_CNT(k) do { tk_acc_[k] += 1; } while (0)
#else
#define CONV_TK(k)
#define CONV_CNT(k)
#endif

// The K loop is written for the fewest VECTOR-ALU instructions per MFMA.  Measured (scripts/experiments/dummy_valu.sh): 32 extra v_add per wave
// and K step cost the 3 x 3 layers 137 -> 149 us, 64 cost 158 us -- plain VALU work does NOT run in the shadow of the MFMAs of the other
// waves of the SIMD, it is added to them (4 cycles per instruction against 64 per MFMA; the one-tile-per-workgroup kernel of rounds 1-3
// issued 4.0 VALU instructions per MFMA = a quarter of its MFMA time).  So: operand pointers are carried and incremented (the gather's
// index arithmetic runs only when the filter tap changes, every Cin / 16 steps, under a wave-uniform branch that contains no load), padding
// pixels are read from a zero page (no select), A is stored in LDS with rows r and r + 32 interleaved so that a lane's two A operands are
// ONE 8-byte read (the compiler pairs two of them into a ds_read2_b64), B rows are 160 floats apart (the two k rows of a read hit
// disjoint banks): ~10 VALU instructions per K step and wave instead of ~50.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void conv_streamk_kernel(ConvArgs a, int tiles_m, int nk, int total) {
    constexpr int BM = 128, BN = 128, LDA = BM + 4, LDB = BN + 32;
    __shared__ __attribute__((aligned(16))) float As[2][CONV_BK][LDA];   // column c = 64 h + 2 j + b holds tile row 64 h + 32 b + j
    __shared__ __attribute__((aligned(16))) float Bs[2][CONV_BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, kl = lane >> 5, l31 = lane & 31;
    const int M = a.N * a.Ho * a.Wo;
    const int acol = tid >> 2, ak4 = (tid & 3) * 4;        // A loader: one float4 per thread: LDS column acol, 4 consecutive k
    const int arow = (acol & 64) | ((acol & 1) << 5) | ((acol >> 1) & 31);
    const int bk = tid >> 5, bn4 = (tid & 31) * 4;         // B loader: one float4 per thread (k row, 4 consecutive n)
    // consecutive workgroup ids go to the 8 XCDs round-robin; worker numbers are handed out so that an XCD's workers own ONE contiguous
    // eighth of the step sequence -- neighbouring tiles, whose activation rows and filter columns then meet in that XCD's L2
    const int worker = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // gridDim.x is a multiple of 8
    int s = (int)((long)worker * total / gridDim.x);
    const int s_end = (int)((long)(worker + 1) * total / gridDim.x);
    const size_t bstep = (size_t)CONV_BK * a.ldw;
#ifdef CONV_STAMPS
    long long tk_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tk_prev_ = (long long)__builtin_readcyclecounter();
    const long long tk_c0_ = tk_prev_, tk_w0_ = (long long)wall_clock64();
#endif
    while (s < s_end) {
        const int tile = s / nk, kc0 = s - tile * nk, kc1 = min(nk, kc0 + (s_end - s));
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
        const int am = m0 + arow;
        const bool am_ok = am < M;
        int an = 0, aho = 0, awo = 0;
        if (am_ok) { an = am / (a.Ho * a.Wo); const int r = am % (a.Ho * a.Wo); aho = r / a.Wo; awo = r % a.Wo; }
        const int hbase = aho * a.stride - a.pad, wbase = awo * a.stride - a.pad;
        int tap_c, tap_h, tap_w;                           // wave-uniform: filter tap / channel offset of the NEXT tile to load
        { const int k0 = kc0 * CONV_BK, tap = k0 / a.Cin; tap_c = k0 % a.Cin; tap_h = tap / a.kw; tap_w = tap % a.kw; }
        const float* pa;                                   // this thread's 4 channels of the next tile's pixel (or the zero page)
        auto retap = [&]() {
            const int hi = hbase + tap_h, wi = wbase + tap_w;
            const bool ok = am_ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            pa = (ok ? a.x + ((size_t)(an * a.H + hi) * a.W + wi) * a.ldx : g_conv_zero) + tap_c + ak4;
        };
        retap();
        const float* pb = a.w + (size_t)(kc0 * CONV_BK + bk) * a.ldw + n0 + bn4;
        int kb = kc0;                                      // K tile pb points at (clamped to the last one: loads past the segment are unused)
        // Two operand tiles are in flight in registers (sets 0 / 1): a tile's loads have two K steps to arrive.  Every load and LDS store
        // of the loop is unconditional -- behind a branch hipcc waits for ALL outstanding loads at the next use (`s_waitcnt vmcnt(0)`),
        // which would put the newest tile's round trip back on the critical path.
        float4 areg0, areg1, breg0, breg1;
        auto load_tile = [&](auto set) {
            constexpr int S = decltype(set)::value;
            const float4 av = *reinterpret_cast<const float4*>(pa);
            const float4 bv = *reinterpret_cast<const float4*>(pb);
            if constexpr (S == 0) { areg0 = av; breg0 = bv; } else { areg1 = av; breg1 = bv; }
            pb += kb + 1 < nk ? bstep : 0;
            kb += kb + 1 < nk ? 1 : 0;
            tap_c += CONV_BK; pa += CONV_BK;
            if (tap_c >= a.Cin) { tap_c = 0; if (++tap_w == a.kw) { tap_w = 0; ++tap_h; } retap(); }
        };
        auto store_tile = [&](int buf, auto set) {
            constexpr int S = decltype(set)::value;
            const float4 av = S == 0 ? areg0 : areg1, bv = S == 0 ? breg0 : breg1;
            As[buf][ak4 + 0][acol] = av.x; As[buf][ak4 + 1][acol] = av.y;
            As[buf][ak4 + 2][acol] = av.z; As[buf][ak4 + 3][acol] = av.w;
            *reinterpret_cast<float4*>(&Bs[buf][bk][bn4]) = bv;
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        // K step with LDS buffer CUR holding tile kc: register set CUR is free (its tile went to LDS one step ago), set CUR ^ 1 holds
        // tile kc + 1 (in flight since the previous step)
        auto k_step = [&](auto cur_c) {
            constexpr int CUR = decltype(cur_c)::value;
            load_tile(std::integral_constant<int, CUR>{});
#ifdef CONV_DUMMY_VALU
            {   // experiment (scripts/experiments/dummy_valu.sh): does plain VALU work run in the shadow of the MFMAs?
                int dummy_ = kb;
#pragma unroll
                for (int q = 0; q < CONV_DUMMY_VALU; ++q) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy_));
                if (dummy_ == -12345) acc[0][0] = 1.f;
            }
#endif
            CONV_TK(1);
#pragma unroll
            for (int kk = 0; kk < CONV_BK; kk += 2) {
                const float2 av = *reinterpret_cast<const float2*>(&As[CUR][kk + kl][wm * 64 + 2 * l31]);
                const float bf = Bs[CUR][kk + kl][wn * 32 + l31];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf, acc[1], 0, 0, 0);
            }
            CONV_TK(2);
            store_tile(CUR ^ 1, std::integral_constant<int, CUR ^ 1>{});
            CONV_TK(3);
            __syncthreads();                                   // also fences the LDS buffers against the next segment's first store
            CONV_TK(4); CONV_CNT(6);
        };
        load_tile(S0{});
        load_tile(S1{});
        store_tile(0, S0{});
        __syncthreads();
        CONV_TK(0); CONV_CNT(7);
        int kc = kc0;
        for (; kc + 1 < kc1; kc += 2) {
            k_step(S0{});
            k_step(S1{});
        }
        if (kc < kc1) k_step(S0{});
        const int nl = wn * 32 + l31, n = n0 + nl, rbase = 4 * kl;
        if (kc0 == 0 && kc1 == nk) {                           // the whole tile: ordinary epilogue
            if (n < a.Cout) conv_epilogue(a, acc, m0 + wm * 64 + rbase, n, M, m0 + BM);
        } else {
            float* part = a.partial + ((size_t)worker * 2 + (kc0 == 0 ? 1 : 0)) * (BM * BN);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    part[ml * BN + nl] = acc[mi][r];
                }
        }
        CONV_TK(5);
        s += kc1 - kc0;
    }
#ifdef CONV_STAMPS
    if (threadIdx.x == 0) {
        for (int k = 0; k < 8; ++k) g_conv_stamps[blockIdx.x][k] += tk_acc_[k];
        g_conv_stamps[blockIdx.x][8] += (long long)__builtin_readcyclecounter() - tk_c0_;
        g_conv_stamps[blockIdx.x][9] += (long long)wall_clock64() - tk_w0_;
    }
#endif
}

// One workgroup per (tile, band of 16 rows): adds the tile's partial segments in ascending K order and finishes the layer.
__global__ __launch_bounds__(256)
void conv_streamk_fixup_kernel(ConvArgs a, int tiles_m, int nk, int total, int W) {
    constexpr int BM = 128, BN = 128;
    const int tile = blockIdx.x, s_lo = tile * nk, s_hi = s_lo + nk;
    auto start = [&](int w) { return (int)((long)w * total / W); };
    int w = (int)((long)s_lo * W / total);
    while (start(w + 1) <= s_lo) ++w;
    while (start(w) > s_lo) --w;                               // worker w owns step s_lo
    if (start(w + 1) >= s_hi) return;                          // it owns the whole tile and has written y itself
    const int M = a.N * a.Ho * a.Wo;
    const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + i * 256;                   // 16 rows x 32 float4
        const int ml = blockIdx.y * 16 + (f >> 5), nl = (f & 31) * 4;
        const int m = m0 + ml, n = n0 + nl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ww = w; ww < W && start(ww) < s_hi; ++ww) {
            const int slot = start(ww) <= s_lo ? 1 : 0;       // the segment that begins the tile sits in slot 1
            const float4 p = *reinterpret_cast<const float4*>(a.partial + ((size_t)ww * 2 + slot) * (BM * BN) + ml * BN + nl);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (m >= M || n >= a.Cout) continue;
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] += a.bias ? a.bias[n + e] : 0.f;
            if (a.residual) o[e] += a.residual[(size_t)m * a.ldr + n + e];
            if (a.act == 1) o[e] = fmaxf(o[e], 0.f);
            else if (a.act == 2) o[e] = 1.0f / (1.0f + expf(-o[e]));
        }
        *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (resnet.py:107,141); one thread per (pixel, 4 channels)
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = C / 4;
    const long total = (long)N * Ho * Wo * c4;
    if (idx >= total) return;
    const int c = (int)(idx % c4) * 4;
    long p = idx / c4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dh = 0; dh < 3; ++dh) {
        const int hi = ho * 2 + dh - 1;
        if (hi < 0 || hi >= H) continue;
        for (int dw = 0; dw < 3; ++dw) {
            const int wi = wo * 2 + dw - 1;
            if (wi < 0 || wi >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + hi) * W + wi) * C + c);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(y + ((size_t)(n * Ho + ho) * Wo + wo) * C + c) = m;
}

// AvgPool2d(7) over the whole 7x7 map followed by ReLU (resnet.py:111,149-151); y row stride ldy
__global__ void avgpool_relu_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C, int ldy) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx % C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((size_t)n * HW + p) * C + c];
    y[(size_t)n * ldy + c] = fmaxf(s / (float)HW, 0.f);
}
