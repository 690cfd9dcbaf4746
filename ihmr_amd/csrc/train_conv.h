// Training-mode kernels of the image encoder (groundwork for `train_baseline.py`): what
// `loss.backward()` runs through for `InterHandEncoder` (models/networks.py:30-80) / `ResNet` (models/resnet.py:97-156;
// Bottleneck :58-94) in train mode -- BatchNorm2d with batch statistics (forward and backward), the weight gradient of a
// convolution as an implicit GEMM on the fp32 matrix cores, zero insertion for the input gradient of the stride-2
// convolutions (the input gradient itself is a stride-1 convolution with the flipped, transposed filter and runs through
// conv_igemm_kernel), and the backward passes of MaxPool2d(3, 2, 1) and AvgPool2d(7) + ReLU.  NHWC activations as in
// encoder.h; every reduction has a fixed order (deterministic results).
#pragma once
#include "encoder.h"

// ------------------------------------------------------------------------------------- column reductions (BatchNorm)
// z [M][C] (row stride ld; C, ld % 4 == 0).  grid = (ceil(C / 4 / 256), S), block = 256: workgroup (cb, s) owns the columns of its 256
// threads' float4s and rows [s * rows_per, (s + 1) * rows_per).  part [S][nq][C].
//   MODE 0: sum z                      (mean)
//   MODE 1: sum (z - mean)^2           (biased variance, second pass as torch's batch_norm_cpu_update_stats does)
//   MODE 2: sum g, sum g * xhat        (backward: g = gradient w.r.t. the BN output, xhat = (z - mean) * invstd)
//   MODE 3: sum z, sum (z - z0)^2      (mean and variance in ONE pass: z0 = row 0 of the matrix, a sample of the channel,
//                                        as the pivot: var = E(z - z0)^2 - (mean - z0)^2 loses no more than the factor
//                                        1 + (mean - z0)^2 / var of fp32 precision, a few units for a pivot within 2 sigma)
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ z, const float* __restrict__ g, int M, int C, int ld,
                                                         int ldg, int rows_per, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ part,
                                                         const float* __restrict__ relu_y = nullptr) {
    // Round 4: 16-byte loads, four rows in flight per thread, up to 1024 row chunks (round 3: 4-byte loads, one row in flight, <= 256 chunks:
    // 1.4-2.7 TB/s on tensors of 50-200 MB, a quarter of the IHMR-Baseline training step).  Thread = 4 consecutive channels; the workgroup
    // covers CW = min(C / 4, 256) such columns and 256 / CW rows at a time (C = 64: 16 rows = 4 KB contiguous per step); the row phases'
    // sums meet in LDS and are added in phase order.
    constexpr int NQ = MODE >= 2 ? 2 : 1;
    __shared__ float4 red[NQ][256];
    const int c4 = C / 4, CW = min(c4, 256), RP = 256 / CW;
    const int tid = threadIdx.x, col = blockIdx.x * CW + tid % CW, phase = tid / CW, c = col * 4, s = blockIdx.y;
    const int r0 = s * rows_per, r1 = min(M, r0 + rows_per);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (col < c4 && phase < RP) {                          // (C / 4 not a power of two: the last 256 - RP * CW threads have no row phase)
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 mu = MODE == 3 ? *reinterpret_cast<const float4*>(z + c) : (MODE >= 1 ? *reinterpret_cast<const float4*>(mean + c) : zero);
        const float4 is = MODE == 2 ? *reinterpret_cast<const float4*>(invstd + c) : zero;
        auto add = [&](const float4 v, float4 gv, const float4 y) {
            if (MODE == 0) { a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
            else if (MODE == 1) {
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                a0.x += dx * dx; a0.y += dy * dy; a0.z += dz * dz; a0.w += dw * dw;
            } else if (MODE == 3) {
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
                a1.x += dx * dx; a1.y += dy * dy; a1.z += dz * dz; a1.w += dw * dw;
            } else {    // relu_y: the unit's output; its ReLU mask is applied to g on the fly (no separate masking pass)
                if (relu_y) {
                    if (!(y.x > 0.f)) gv.x = 0.f;
                    if (!(y.y > 0.f)) gv.y = 0.f;
                    if (!(y.z > 0.f)) gv.z = 0.f;
                    if (!(y.w > 0.f)) gv.w = 0.f;
                }
                a0.x += gv.x; a0.y += gv.y; a0.z += gv.z; a0.w += gv.w;
                a1.x += gv.x * ((v.x - mu.x) * is.x); a1.y += gv.y * ((v.y - mu.y) * is.y);
                a1.z += gv.z * ((v.z - mu.z) * is.z); a1.w += gv.w * ((v.w - mu.w) * is.w);
            }
        };
        int r = r0 + phase;
        for (; r + 3 * RP < r1; r += 4 * RP) {             // four rows of this thread in flight
            float4 v[4], gv[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const float4*>(z + (size_t)(r + u * RP) * ld + c);
                gv[u] = MODE == 2 ? *reinterpret_cast<const float4*>(g + (size_t)(r + u * RP) * ldg + c) : zero;
                y[u] = (MODE == 2 && relu_y) ? *reinterpret_cast<const float4*>(relu_y + (size_t)(r + u * RP) * ld + c) : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u], gv[u], y[u]);
        }
        for (; r < r1; r += RP) {
            const float4 v = *reinterpret_cast<const float4*>(z + (size_t)r * ld + c);
            const float4 gv = MODE == 2 ? *reinterpret_cast<const float4*>(g + (size_t)r * ldg + c) : zero;
            const float4 y = (MODE == 2 && relu_y) ? *reinterpret_cast<const float4*>(relu_y + (size_t)r * ld + c) : zero;
            add(v, gv, y);
        }
    }
    red[0][tid] = a0;
    if (NQ == 2) red[NQ - 1][tid] = a1;
    __syncthreads();
    if (phase == 0 && col < c4) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 t = red[q][tid];
            for (int ph = 1; ph < RP; ++ph) {
                const float4 o = red[q][tid + ph * CW];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            *reinterpret_cast<float4*>(part + ((size_t)s * NQ + q) * C + c) = t;
        }
    }
}

// out[q][c] = sum_s part[s][q][c] * scale, accumulated in float64: a workgroup owns 16 (q, c) entries; thread (g, e) = (tid / 16,
// tid % 16) adds the chunks g, g + 16, ... of entry e (16 serial additions for 256 chunks instead of 64), the sixteen group sums
// are added in group order.  With `invstd_out` the result is a variance: invstd = 1 / sqrt(var + eps).
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ part, int S, int NQ, int C, double scale,
                                                        float* __restrict__ out, float* __restrict__ invstd_out, float eps,
                                                        float* __restrict__ out1 = nullptr) {
    __shared__ double red[16][17];
    const int e = threadIdx.x % 16, g = threadIdx.x / 16, i = blockIdx.x * 16 + e;
    const bool ok = i < NQ * C;
    const int q = ok ? i / C : 0, c = ok ? i % C : 0;
    double s = 0.0;
    if (ok) {
        int k = g;
        for (; k + 7 * 16 < S; k += 8 * 16) {            // eight chunks' loads in flight, added in chunk order (the same sum as one by one)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(k + 16 * u) * NQ + q) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; k < S; k += 16) s += (double)part[((size_t)k * NQ + q) * C + c];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && ok) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e];
        const float v = (float)(t * scale);
        if (q == 1 && out1) out1[c] = v;                 // second quantity to its own buffer (dgamma beside dbeta)
        else out[(size_t)q * C + c] = v;
        if (invstd_out) invstd_out[c] = 1.0f / sqrtf(v + eps);
    }
}

// mean / biased variance / invstd from the MODE 3 partial sums (same thread layout as bn_finish_kernel): part [S][2][C]
__global__ __launch_bounds__(256) void bn_finish_stats_kernel(const float* __restrict__ part, const float* __restrict__ z, int S, int C,
                                                              long M, float* __restrict__ mean, float* __restrict__ var,
                                                              float* __restrict__ invstd, float eps, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, float momentum) {
    __shared__ double red[2][16][17];
    const int e = threadIdx.x % 16, g = threadIdx.x / 16, c = blockIdx.x * 16 + e;
    const bool ok = c < C;
    double s0 = 0.0, s1 = 0.0;
    if (ok) {
        int k = g;
        for (; k + 7 * 16 < S; k += 8 * 16) {            // eight chunks' loads in flight, added in chunk order
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v0[u] = part[((size_t)(k + 16 * u) * 2) * C + c]; v1[u] = part[((size_t)(k + 16 * u) * 2 + 1) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (double)v0[u]; s1 += (double)v1[u]; }
        }
        for (; k < S; k += 16) { s0 += (double)part[((size_t)k * 2) * C + c]; s1 += (double)part[((size_t)k * 2 + 1) * C + c]; }
    }
    red[0][g][e] = s0; red[1][g][e] = s1;
    __syncthreads();
    if (g == 0 && ok) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t0 += red[0][k][e]; t1 += red[1][k][e]; }
        const double mu = t0 / (double)M, d = mu - (double)z[c];
        const float v = (float)fmax(t1 / (double)M - d * d, 0.0);
        mean[c] = (float)mu;
        var[c] = v;
        invstd[c] = 1.0f / sqrtf(v + eps);
        if (run_mean) {     // nn.BatchNorm2d's running statistics: (1 - momentum) * running + momentum * batch (unbiased variance)
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mu;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (v * (float)M / (float)(M > 1 ? M - 1 : 1));
        }
    }
}

// y = [relu]( gamma * (z - mean) * invstd + beta [+ residual] ), 4 channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ residual,
                                                       float* __restrict__ y, long M, int C, int relu) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= M * c4) return;
    const int c = (int)(idx % c4) * 4;
    const size_t off = (size_t)(idx / c4) * C + c;
    const float4 v = *reinterpret_cast<const float4*>(z + off);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v.x - mu.x) * is.x * ga.x + be.x; o.y = (v.y - mu.y) * is.y * ga.y + be.y;
    o.z = (v.z - mu.z) * is.z * ga.z + be.z; o.w = (v.w - mu.w) * is.w * ga.w + be.w;
    if (residual) { const float4 r = *reinterpret_cast<const float4*>(residual + off); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + off) = o;
}

// dz = gamma * invstd * (g - sum_g / M - xhat * sum_gx / M)   (torch's batch_norm_backward in training mode)
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                                const float* __restrict__ sum_gx, float* __restrict__ dz, long M, int C,
                                                                const float* __restrict__ relu_y) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= M * c4) return;
    const int c = (int)(idx % c4) * 4;
    const size_t off = (size_t)(idx / c4) * C + c;
    const float inv_m = 1.0f / (float)M;
    const float4 v = *reinterpret_cast<const float4*>(z + off);
    float4 gv = *reinterpret_cast<const float4*>(g + off);
    if (relu_y) {
        const float4 yy = *reinterpret_cast<const float4*>(relu_y + off);
        if (!(yy.x > 0.f)) gv.x = 0.f;
        if (!(yy.y > 0.f)) gv.y = 0.f;
        if (!(yy.z > 0.f)) gv.z = 0.f;
        if (!(yy.w > 0.f)) gv.w = 0.f;
    }
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    const float4 s1 = *reinterpret_cast<const float4*>(sum_g + c), s2 = *reinterpret_cast<const float4*>(sum_gx + c);
    float4 o;
    o.x = ga.x * is.x * (gv.x - s1.x * inv_m - (v.x - mu.x) * is.x * (s2.x * inv_m));
    o.y = ga.y * is.y * (gv.y - s1.y * inv_m - (v.y - mu.y) * is.y * (s2.y * inv_m));
    o.z = ga.z * is.z * (gv.z - s1.z * inv_m - (v.z - mu.z) * is.z * (s2.z * inv_m));
    o.w = ga.w * is.w * (gv.w - s1.w * inv_m - (v.w - mu.w) * is.w * (s2.w * inv_m));
    *reinterpret_cast<float4*>(dz + off) = o;
}

// ------------------------------------------------------------------------------------- pooling backward
// MaxPool2d(3, stride 2, padding 1) backward: the gradient of an output pixel goes to the FIRST maximum of its window in
// (row, column) order, as torch records it.  One thread per (input pixel, 4 channels) gathers from the <= 2 x 2 windows
// that contain it (no atomics: deterministic).
__global__ __launch_bounds__(256) void maxpool3x3s2_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= (long)N * H * W * c4) return;
    const int c = (int)(idx % c4) * 4;
    long p = idx / c4;
    const int wi = (int)(p % W); p /= W;
    const int hi = (int)(p % H);
    const int n = (int)(p / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ho = (hi + 1) / 2 - ((hi + 1) % 2 == 0 ? 1 : 0); ho <= (hi + 1) / 2; ++ho) {      // windows with 2 ho - 1 <= hi <= 2 ho + 1
        if (ho < 0 || ho >= Ho) continue;
        for (int wo = (wi + 1) / 2 - ((wi + 1) % 2 == 0 ? 1 : 0); wo <= (wi + 1) / 2; ++wo) {
            if (wo < 0 || wo >= Wo) continue;
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int arg[4] = {-1, -1, -1, -1};
            for (int dh = 0; dh < 3; ++dh) {
                const int h2 = ho * 2 + dh - 1;
                if (h2 < 0 || h2 >= H) continue;
                for (int dw = 0; dw < 3; ++dw) {
                    const int w2 = wo * 2 + dw - 1;
                    if (w2 < 0 || w2 >= W) continue;
                    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + h2) * W + w2) * C + c);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (vv[e] > best[e] || vv[e] != vv[e]) { best[e] = vv[e]; arg[e] = h2 * W + w2; }
                }
            }
            const float4 g = *reinterpret_cast<const float4*>(dy + ((size_t)(n * Ho + ho) * Wo + wo) * C + c);
            const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (arg[e] == hi * W + wi) acc[e] += gg[e];
        }
    }
    *reinterpret_cast<float4*>(dx + ((size_t)(n * H + hi) * W + wi) * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// AvgPool2d over the whole map followed by ReLU, backward: dx[n][p][c] = (y[n][c] > 0 ? dy[n][c] : 0) / HW
__global__ __launch_bounds__(256) void avgpool_relu_backward_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int N, int HW, int C, int ldy) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * HW * C) return;
    const int c = (int)(idx % C);
    const int n = (int)(idx / ((long)HW * C));
    dx[idx] = y[(size_t)n * ldy + c] > 0.f ? dy[(size_t)n * ldy + c] / (float)HW : 0.f;
}

// zero insertion for the input gradient of a stride-2 convolution: out [N][2 Ho][2 Wo][C], out[n][2 ho][2 wo] = dy[n][ho][wo]
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ dy, float* __restrict__ out, int N, int Ho, int Wo, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= (long)N * 2 * Ho * 2 * Wo * c4) return;
    const int c = (int)(idx % c4) * 4;
    long p = idx / c4;
    const int w = (int)(p % (2 * Wo)); p /= 2 * Wo;
    const int h = (int)(p % (2 * Ho));
    const int n = (int)(p / (2 * Ho));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(h & 1) && !(w & 1)) v = *reinterpret_cast<const float4*>(dy + ((size_t)(n * Ho + h / 2) * Wo + w / 2) * C + c);
    *reinterpret_cast<float4*>(out + idx * 4) = v;
}

// the four parity phases of a stride-2 input gradient back into place: dx[n][2 io + pi][2 jo + pj] = phase[2 pi + pj][n][io][jo]
// (each phase [N][Ho][Wo][C]; dx [N][2Ho][2Wo][C])
__global__ __launch_bounds__(256) void interleave2_kernel(const float* __restrict__ p00, const float* __restrict__ p01,
                                                          const float* __restrict__ p10, const float* __restrict__ p11,
                                                          float* __restrict__ dx, int N, int Ho, int Wo, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (idx >= (long)N * 2 * Ho * 2 * Wo * c4) return;
    const int c = (int)(idx % c4) * 4;
    long p = idx / c4;
    const int w = (int)(p % (2 * Wo)); p /= 2 * Wo;
    const int h = (int)(p % (2 * Ho));
    const int n = (int)(p / (2 * Ho));
    const float* src = (h & 1) ? ((w & 1) ? p11 : p10) : ((w & 1) ? p01 : p00);
    *reinterpret_cast<float4*>(dx + idx * 4) = *reinterpret_cast<const float4*>(src + ((size_t)(n * Ho + h / 2) * Wo + w / 2) * C + c);
}

// the filter of the input gradient from the forward filter: w [kh*kw*Cin][ldw] (k = (fh, fw, ci), column co) ->
// out [kh*kw*Cout][ldo] with out[((kh-1-fh) * kw + (kw-1-fw)) * Cout + co][ci] = w[(fh * kw + fw) * Cin + ci][co]; one thread per
// output element, coalesced over ci (the reads walk a column of w: layout plumbing, once per optimizer step)
__global__ __launch_bounds__(256) void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int kh, int kw,
                                                                int Cin, int Cout, int ldw, int ldo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)kh * kw * Cout * Cin) return;
    const int ci = (int)(idx % Cin);
    long r = idx / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout), fh = kh - 1 - tap / kw, fw = kw - 1 - tap % kw;
    out[(size_t)r * ldo + ci] = w[((size_t)(fh * kw + fw) * Cin + ci) * ldw + co];
}

// ------------------------------------------------------------------------------------- weight gradient
// dW[k][n] = sum_m A(x)[m][k] . dY[m][n]:  k = (filter tap, input channel) as in the forward's K index, m = output pixel.
// The same tiling as conv_igemm_kernel with the roles turned: the REDUCTION runs over the pixels (16 per step), a workgroup
// owns BM values of k x BN output channels; the x patch of 16 pixels x BM k-values is gathered as float4 along the channel
// (Cin % 4 == 0: a float4 never straddles a filter tap; the tap of a loader is fixed for the whole kernel, only the pixel
// moves) and stored pixel-major in LDS -- exactly what the MFMA's A operand wants, no transposition; dY rows are the B operand.
// gridDim.z workgroups split the pixel range; partial sums go to `partial` [z][K][Cout] and conv_splitk_reduce_kernel adds
// them in order.
struct WgradArgs {
    const float* x;      // NHWC input of the convolution, pixel stride ldx
    const float* dy;     // [M][lddy] gradient w.r.t. the convolution output
    float* partial;      // [msplit][K][Cout]
    int N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, ldx, lddy, chunks_per;
};

template <int BM, int BN>
__global__ __launch_bounds__((BM / 64) * (BN / 32) * 64) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int WN_WAVES = BN / 32;
    constexpr int THREADS = (BM / 64) * WN_WAVES * 64;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_F4 = CONV_BK * BM / 4 / THREADS, B_F4 = CONV_BK * BN / 4 / THREADS;
    constexpr int A_PER_ROW = BM / 4, B_PER_ROW = BN / 4;
    __shared__ __attribute__((aligned(16))) float As[2][CONV_BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][CONV_BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int M = a.N * a.Ho * a.Wo, K = a.kh * a.kw * a.Cin, HoWo = a.Ho * a.Wo;
    const int k0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // A loaders: float4 f -> (pixel of the chunk f / A_PER_ROW, 4 consecutive k at k0 + (f % A_PER_ROW) * 4)
    int arow[A_F4], acol[A_F4], afh[A_F4], afw[A_F4], ac[A_F4];
    bool aok[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * THREADS;
        arow[i] = f / A_PER_ROW; acol[i] = (f % A_PER_ROW) * 4;
        const int k = k0 + acol[i];
        aok[i] = k < K;
        const int tap = aok[i] ? k / a.Cin : 0;
        ac[i] = aok[i] ? k % a.Cin : 0; afh[i] = tap / a.kw; afw[i] = tap % a.kw;
    }
    const int nchunks = (M + CONV_BK - 1) / CONV_BK;
    const int mc0 = blockIdx.z * a.chunks_per, mc1 = min(nchunks, mc0 + a.chunks_per);
    float4 areg[A_F4], breg[B_F4];
    // Vector-ALU instructions are not hidden behind the MFMAs (csrc/encoder.h), and this loop walks PIXELS: rounds 1-3 decomposed every
    // loader's pixel index with four integer divisions per K step (~200 VALU instructions per 16 MFMAs).  Now: two exact small-integer
    // divisions by multiplication with a float reciprocal (+- 1 correction; m < 2^23) and the filter-tap offsets folded into per-loader
    // constants; the dY loader carries its pointer.
    const float r_howo = 1.0f / (float)HoWo, r_wo = 1.0f / (float)a.Wo;
    auto sdiv = [](int v, int d, float rd) {               // v / d for 0 <= v < 2^23, d > 0
        int q = (int)((float)v * rd);
        const int r = v - q * d;
        q += r >= d ? 1 : (r < 0 ? -1 : 0);
        return q;
    };
    int am[A_F4], hoff[A_F4], woff[A_F4];                  // next pixel of each loader; its tap's offset in the input image
#pragma unroll
    for (int i = 0; i < A_F4; ++i) { am[i] = mc0 * CONV_BK + arow[i]; hoff[i] = afh[i] - a.pad; woff[i] = afw[i] - a.pad; }
    const float* pb[B_F4];
    int bm[B_F4];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int g = tid + i * THREADS;
        bm[i] = mc0 * CONV_BK + g / B_PER_ROW;
        pb[i] = a.dy + (size_t)bm[i] * a.lddy + n0 + (g % B_PER_ROW) * 4;
    }
    const bool bn_ok0 = n0 + (tid % B_PER_ROW) * 4 < a.lddy;   // THREADS % B_PER_ROW == 0: the same column for every float4 of a thread
    const size_t bstep = (size_t)CONV_BK * a.lddy;
    auto load_tile = [&](int) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int m = am[i];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (aok[i] && m < M) {
                const int n = sdiv(m, HoWo, r_howo), r = m - n * HoWo, ho = sdiv(r, a.Wo, r_wo), wo = r - ho * a.Wo;
                const int hi = ho * a.stride + hoff[i], wi = wo * a.stride + woff[i];
                if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                    v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + hi) * a.W + wi) * a.ldx + ac[i]);
            }
            areg[i] = v;
            am[i] += CONV_BK;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            breg[i] = (bm[i] < M && bn_ok0) ? *reinterpret_cast<const float4*>(pb[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            bm[i] += CONV_BK; pb[i] += bstep;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(&As[buf][arow[i]][acol[i]]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int g = tid + i * THREADS;
            *reinterpret_cast<float4*>(&Bs[buf][g / B_PER_ROW][(g % B_PER_ROW) * 4]) = breg[i];
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    const int kl = lane >> 5, l31 = lane & 31;
    if (mc0 < mc1) { load_tile(mc0); store_tile(0); }
    __syncthreads();
    for (int mc = mc0; mc < mc1; ++mc) {
        const int cur = (mc - mc0) & 1;
        if (mc + 1 < mc1) load_tile(mc + 1);
#pragma unroll
        for (int mm = 0; mm < CONV_BK; mm += 2) {
            const float a0 = As[cur][mm + kl][wm * 64 + l31], a1 = As[cur][mm + kl][wm * 64 + 32 + l31];
            const float bf = Bs[cur][mm + kl][wn * 32 + l31];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf, acc[1], 0, 0, 0);
        }
        if (mc + 1 < mc1) store_tile(cur ^ 1);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n >= a.Cout) return;
    const int kw0 = k0 + wm * 64 + 4 * kl;
    float* part = a.partial + ((size_t)blockIdx.z * K + kw0) * a.Cout + n;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = mi * 32 + (r & 3) + 8 * (r >> 2);
            if (kw0 + dr < K) part[dr * a.Cout] = acc[mi][r];
        }
}

// dw[k][n] = sum_z partial[z][k][n] for many pixel-range partials: thread (g, e) = (tid / 16, tid % 16) adds the partials g, g + 16, ...
// of float4 e of its workgroup's 16 (16 serial loads for 256 partials instead of 256), the sixteen group sums are added in group order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int K, int Cout,
                                                           int ldw, int nsplit) {
    __shared__ float4 red[16][17];
    const int e = threadIdx.x % 16, g = threadIdx.x / 16, cv = Cout / 4;
    const long i = (long)blockIdx.x * 16 + e;
    const bool ok = i < (long)K * cv;
    const int k = ok ? (int)(i / cv) : 0, n = ok ? (int)(i % cv) * 4 : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok)
        for (int z = g; z < nsplit; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(partial + ((size_t)z * K + k) * Cout + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && ok) {
        float4 t = red[0][e];
#pragma unroll
        for (int q = 1; q < 16; ++q) { t.x += red[q][e].x; t.y += red[q][e].y; t.z += red[q][e].z; t.w += red[q][e].w; }
        *reinterpret_cast<float4*>(dw + (size_t)k * ldw + n) = t;
    }
}
