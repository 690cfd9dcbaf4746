// Image preprocessing in front of the encoder: the reference does this per image on the CPU inside
// its DataLoader workers --
//   DataProcessor.padding_and_resize    data/data_preprocess.py:45-60   (cv2.resize, default INTER_LINEAR, uint8 BGR)
//   DataProcessor.random_flip(do_flip)  data/data_preprocess.py:63-72   (image and 2-D joints; left-only samples at test time,
//                                                                        baseline_dataset.py:71-74)
//   DataProcessor.normalize_joints_2d   data/data_preprocess.py:162-169
//   ToTensor + Normalize(0.5, 0.5)      data/baseline_dataset.py:41-44,202
// Arithmetic = oracle/preprocess_ref.py (cv2 4.2.0 8-bit linear resize in 11-bit fixed point, restated there), bit for bit:
// the coefficients are derived per pixel with the same double / float operations, the two passes with the same
// integer shifts.
//
// Byte-bound work: one thread per output pixel, all three channels; the output is written as three planes so that
// consecutive lanes store consecutive floats (64 x 4 B per wave and plane), the (up to 4 x 3) source bytes of a pixel
// are adjacent, and neighbouring lanes read neighbouring source pixels.  Nothing is staged: a source pixel is read by at
// most 2 x 2 output pixels, which sit in the same wave or the next row's (L2 hit).
#pragma once
#include "ihmr_common.h"

#define PRE_THREADS 256
#define PRE_COEF_SCALE 2048.0f   // INTER_RESIZE_COEF_SCALE (11 bits)

// saturate_cast<short>(float): round half to even, clamp
__device__ __forceinline__ int pre_sat_short(float x) { return max(-32768, min(32767, __float2int_rn(x))); }

struct PreAxis { int s; int w0, w1; };

// coefficient of destination index d on an axis resized src -> dst (resizeGeneric's tables, one entry)
__device__ __forceinline__ PreAxis pre_axis(int d, double scale, int src, bool clamp_fraction) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_fraction) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    PreAxis a;
    a.s = s;
    a.w0 = pre_sat_short((1.0f - f) * PRE_COEF_SCALE);
    a.w1 = pre_sat_short(f * PRE_COEF_SCALE);
    return a;
}

// grid = (ceil(S*S / 256), B), block = 256
__global__ __launch_bounds__(PRE_THREADS) void preprocess_kernel(const uint8_t* __restrict__ pixels, const int64_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ sizes, const uint8_t* __restrict__ do_flip,
                                                                 int S, float* __restrict__ img_out, uint8_t* __restrict__ img_u8,
                                                                 const float* __restrict__ joints_in, float* __restrict__ joints_out) {
    const int b = blockIdx.y;
    const int H = sizes[2 * b], W = sizes[2 * b + 1];
    const bool flip = do_flip && do_flip[b];
    // padding_and_resize: the longer side becomes S
    double ratio;
    int nh, nw;
    if (H > W) { ratio = (double)S / (double)H; nh = S; nw = (int)(ratio * (double)W); }
    else { ratio = (double)S / (double)W; nw = S; nh = (int)(ratio * (double)H); }

    if (joints_in && blockIdx.x == 0 && threadIdx.x < 42) {
        const int j = threadIdx.x;
        const int src = flip ? (j + 21) % 42 : j;                     // the two hands swap on a flip
        const float* q = joints_in + ((size_t)b * 42 + src) * 3;
        const float r = (float)ratio;                                  // joints_2d[:, :2] *= ratio on a float32 array
        float x = q[0] * r, y = q[1] * r;
        if (flip) x = (float)S - x;
        float* o = joints_out + ((size_t)b * 42 + j) * 3;
        o[0] = (x / (float)S) * 2.0f - 1.0f;
        o[1] = (y / (float)S) * 2.0f - 1.0f;
        o[2] = q[2];
    }

    const int p = blockIdx.x * PRE_THREADS + threadIdx.x;
    if (p >= S * S) return;
    const int oy = p / S, ox = p % S;
    const int x = flip ? S - 1 - ox : ox;                             // np.fliplr of the padded image
    int v[3] = {0, 0, 0};
    if (oy < nh && x < nw) {
        const uint8_t* src = pixels + offsets[b];
        const size_t row = (size_t)W * 3;
        if (nw == W && nh == H) {                                     // same size: copy
            const uint8_t* q = src + (size_t)oy * row + (size_t)x * 3;
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
        } else {
            const double sx = 1.0 / ((double)nw / (double)W), sy = 1.0 / ((double)nh / (double)H);
            const double eps = 2.220446049250313e-16;
            if (fabs(sx - 2.0) < eps && fabs(sy - 2.0) < eps) {       // exact 2x decimation: 2x2 box mean
                const uint8_t* q0 = src + (size_t)(2 * oy) * row + (size_t)(2 * x) * 3;
                const uint8_t* q1 = q0 + row;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = ((int)q0[c] + (int)q0[3 + c] + (int)q1[c] + (int)q1[3 + c] + 2) >> 2;
            } else {
                const PreAxis ax = pre_axis(x, sx, W, true), ay = pre_axis(oy, sy, H, false);
                const int x1 = min(ax.s + 1, W - 1);
                const int y0 = min(max(ay.s, 0), H - 1), y1 = min(max(ay.s + 1, 0), H - 1);
                const uint8_t* r0 = src + (size_t)y0 * row;
                const uint8_t* r1 = src + (size_t)y1 * row;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = (int)r0[(size_t)ax.s * 3 + c] * ax.w0 + (int)r0[(size_t)x1 * 3 + c] * ax.w1;
                    const int h1 = (int)r1[(size_t)ax.s * 3 + c] * ax.w0 + (int)r1[(size_t)x1 * 3 + c] * ax.w1;
                    v[c] = ((((ay.w0 * (h0 >> 4)) >> 16) + ((ay.w1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
                }
            }
        }
    }
    const size_t plane = (size_t)S * S;
    float* o = img_out + (size_t)b * 3 * plane + (size_t)oy * S + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = ((float)v[c] / 255.0f - 0.5f) / 0.5f;   // ToTensor, Normalize(0.5, 0.5)
    if (img_u8) {
        uint8_t* u = img_u8 + ((size_t)b * plane + (size_t)oy * S + ox) * 3;
        u[0] = (uint8_t)v[0]; u[1] = (uint8_t)v[1]; u[2] = (uint8_t)v[2];
    }
}
