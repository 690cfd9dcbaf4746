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
// Tile: 128 x BN (BN = 64 | 128) x 16 per workgroup of 4 waves; A is staged K-major in LDS so that the 32
// lanes of an MFMA row read consecutive addresses; global loads for tile t+1 are issued before the MFMAs of
// tile t and written to LDS after them.
#pragma once
#include "ihmr_common.h"

#define CONV_BM 128
#define CONV_BK 16
#define CONV_THREADS 256

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
};

template <int BN>
__global__ __launch_bounds__(CONV_THREADS) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WN_WAVES = BN == 128 ? 2 : 1;            // waves along n
    constexpr int WM_WAVES = 4 / WN_WAVES;                 // waves along m
    constexpr int WM = CONV_BM / WM_WAVES;                 // 64 (BN=128) or 32 (BN=64)
    constexpr int WN = BN / WN_WAVES;                      // 64
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int LDA = CONV_BM + 5, LDB = BN + 4;   // 8*LDA % 32 != 0: the two k-halves of a row hit different banks
    __shared__ float As[CONV_BK][LDA];
    __shared__ float Bs[CONV_BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int M = a.N * a.Ho * a.Wo, K = a.kh * a.kw * a.Cin, Kpad = (K + CONV_BK - 1) / CONV_BK * CONV_BK;
    const int m0 = blockIdx.x * CONV_BM, n0 = blockIdx.y * BN;
    const bool fast = (a.Cin % CONV_BK) == 0;              // a 16-wide K chunk never straddles a filter tap

    // ---- A loader: thread -> (row, 8 consecutive k)
    const int arow = tid >> 1, akoff = (tid & 1) * 8;
    const int am = m0 + arow;
    const bool am_ok = am < M;
    int an = 0, aho = 0, awo = 0;
    if (am_ok) { an = am / (a.Ho * a.Wo); const int r = am % (a.Ho * a.Wo); aho = r / a.Wo; awo = r % a.Wo; }
    // ---- B loader: thread -> (k row(s), 4 consecutive n)
    constexpr int B_F4_PER_ROW = BN / 4;                   // 32 or 16
    constexpr int B_ROWS_PER_PASS = CONV_THREADS / B_F4_PER_ROW;  // 8 or 16
    constexpr int B_PASSES = CONV_BK / B_ROWS_PER_PASS;    // 2 or 1
    const int bk = tid / B_F4_PER_ROW, bn4 = (tid % B_F4_PER_ROW) * 4;

    float areg[8];
    float4 breg[B_PASSES];
    auto load_tile = [&](int kc) {
        const int k0 = kc * CONV_BK;
        if (fast) {
            const int tap = k0 / a.Cin, c0 = k0 % a.Cin;
            const int fh = tap / a.kw, fw = tap % a.kw;
            const int hi = aho * a.stride + fh - a.pad, wi = awo * a.stride + fw - a.pad;
            const bool ok = am_ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            if (ok) {
                const float4* p = reinterpret_cast<const float4*>(a.x + ((size_t)(an * a.H + hi) * a.W + wi) * a.ldx + c0 + akoff);
                const float4 v0 = p[0], v1 = p[1];
                areg[0] = v0.x; areg[1] = v0.y; areg[2] = v0.z; areg[3] = v0.w;
                areg[4] = v1.x; areg[5] = v1.y; areg[6] = v1.z; areg[7] = v1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) areg[e] = 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + akoff + e;
                float v = 0.f;
                if (am_ok && k < K) {
                    const int c = k % a.Cin, tap = k / a.Cin, fh = tap / a.kw, fw = tap % a.kw;
                    const int hi = aho * a.stride + fh - a.pad, wi = awo * a.stride + fw - a.pad;
                    if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) v = a.x[((size_t)(an * a.H + hi) * a.W + wi) * a.ldx + c];
                }
                areg[e] = v;
            }
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int k = k0 + bk + p * B_ROWS_PER_PASS;
            breg[p] = *reinterpret_cast<const float4*>(a.w + (size_t)k * a.ldw + n0 + bn4);   // ldw >= n0 + BN, zero padded
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) As[akoff + e][arow] = areg[e];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) *reinterpret_cast<float4*>(&Bs[bk + p * B_ROWS_PER_PASS][bn4]) = breg[p];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = Kpad / CONV_BK;
    load_tile(0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();            // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (kc + 1 < nk) load_tile(kc + 1);   // in flight during the MFMAs below
        const int kl = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int kk = 0; kk < CONV_BK; kk += 2) {
            float af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = As[kk + kl][wm * WM + mi * 32 + l31];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = Bs[kk + kl][wn * WN + ni * 32 + l31];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
    }

    // ---- epilogue: bias (+ residual) (+ activation); C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * WN + ni * 32 + col;
            if (n >= a.Cout) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                if (m >= M) continue;
                float v = acc[mi][ni][r] + bv;
                if (a.residual) v += a.residual[(size_t)m * a.ldr + n];
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = 1.0f / (1.0f + expf(-v));
                a.y[(size_t)m * a.ldy + n] = v;
            }
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
