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
// Byte-bound work: one thread per FOUR horizontally adjacent output pixels, all three channels; the output is written as
// three planes with one 16-byte store per thread and plane (64 x 16 B contiguous per wave), the padded uint8 image with three
// 4-byte stores; the per-image scale factors (double divisions) are derived once per workgroup and shared through LDS.  Nothing is
// staged: a source pixel is read by at most 2 x 2 output pixels, which sit in the same thread, wave or the next row's (L2 hit).
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

// per-image geometry, derived once per workgroup by its first thread (the double divisions of the scale factors are the
// expensive part of the coefficient arithmetic; every pixel of an image shares them)
struct PreImage {
    double sx, sy;        // source step per destination pixel: 1.0 / ((double)dst / src)
    float ratio;          // joints_2d[:, :2] *= ratio on a float32 array
    int H, W, nh, nw;
    int mode;             // 0 copy (equal sizes), 1 exact 2x decimation (2x2 box mean), 2 generic fixed-point linear
};

// geometry of image b from its (H, W): padding_and_resize's size arithmetic and cv::resize's scale factors
__device__ __forceinline__ PreImage pre_geometry(int H, int W, int S) {
    double ratio;
    int nh, nw;
    if (H > W) { ratio = (double)S / (double)H; nh = S; nw = (int)(ratio * (double)W); }      // the longer side becomes S
    else { ratio = (double)S / (double)W; nw = S; nh = (int)(ratio * (double)H); }
    PreImage p;
    p.H = H; p.W = W; p.nh = nh; p.nw = nw; p.ratio = (float)ratio;
    p.sx = 1.0 / ((double)nw / (double)W);
    p.sy = 1.0 / ((double)nh / (double)H);
    const double eps = 2.220446049250313e-16;
    p.mode = (nw == W && nh == H) ? 0 : ((fabs(p.sx - 2.0) < eps && fabs(p.sy - 2.0) < eps) ? 1 : 2);
    return p;
}

// grid = (ceil(S*S / 4 / 256), B), block = 256: a thread owns FOUR horizontally adjacent output pixels (S % 4 == 0): the row
// coefficients are shared, every plane gets one 16-byte store per thread (64 lanes x 16 B contiguous per wave and plane).
#define PRE_PPT 4
__global__ __launch_bounds__(PRE_THREADS) void preprocess_kernel(const uint8_t* __restrict__ pixels, const int64_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ sizes, const uint8_t* __restrict__ do_flip,
                                                                 int S, float* __restrict__ img_out, uint8_t* __restrict__ img_u8,
                                                                 const float* __restrict__ joints_in, float* __restrict__ joints_out) {
    __shared__ PreImage gsh;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) gsh = pre_geometry(sizes[2 * b], sizes[2 * b + 1], S);
    __syncthreads();
    const PreImage g = gsh;
    const int H = g.H, W = g.W, nh = g.nh, nw = g.nw;
    const bool flip = do_flip && do_flip[b];

    if (joints_in && blockIdx.x == 0 && threadIdx.x < 42) {
        const int j = threadIdx.x;
        const int src = flip ? (j + 21) % 42 : j;                     // the two hands swap on a flip
        const float* q = joints_in + ((size_t)b * 42 + src) * 3;
        float x = q[0] * g.ratio, y = q[1] * g.ratio;
        if (flip) x = (float)S - x;
        float* o = joints_out + ((size_t)b * 42 + j) * 3;
        o[0] = (x / (float)S) * 2.0f - 1.0f;
        o[1] = (y / (float)S) * 2.0f - 1.0f;
        o[2] = q[2];
    }

    const int p4 = blockIdx.x * PRE_THREADS + threadIdx.x;
    if (p4 * PRE_PPT >= S * S) return;
    const int oy = (p4 * PRE_PPT) / S, ox0 = (p4 * PRE_PPT) % S;
    const uint8_t* src = pixels + offsets[b];
    const size_t row = (size_t)W * 3;
    int v[PRE_PPT][3];
    // the two source rows and their weights are shared by the four pixels
    PreAxis ay = {0, 0, 0};
    const uint8_t *r0 = src, *r1 = src;
    if (oy < nh) {
        if (g.mode == 2) {
            ay = pre_axis(oy, g.sy, H, false);
            r0 = src + (size_t)min(max(ay.s, 0), H - 1) * row;
            r1 = src + (size_t)min(max(ay.s + 1, 0), H - 1) * row;
        } else if (g.mode == 1) {
            r0 = src + (size_t)(2 * oy) * row; r1 = r0 + row;
        } else {
            r0 = src + (size_t)oy * row;
        }
    }
#pragma unroll
    for (int i = 0; i < PRE_PPT; ++i) {
        const int ox = ox0 + i;
        const int x = flip ? S - 1 - ox : ox;                         // np.fliplr of the padded image
        v[i][0] = v[i][1] = v[i][2] = 0;
        if (oy < nh && x < nw) {
            if (g.mode == 0) {                                        // same size: copy
                const uint8_t* q = r0 + (size_t)x * 3;
                v[i][0] = q[0]; v[i][1] = q[1]; v[i][2] = q[2];
            } else if (g.mode == 1) {                                 // exact 2x decimation: 2x2 box mean
                const uint8_t *q0 = r0 + (size_t)(2 * x) * 3, *q1 = r1 + (size_t)(2 * x) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[i][c] = ((int)q0[c] + (int)q0[3 + c] + (int)q1[c] + (int)q1[3 + c] + 2) >> 2;
            } else {
                const PreAxis ax = pre_axis(x, g.sx, W, true);
                const int x1 = min(ax.s + 1, W - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = (int)r0[(size_t)ax.s * 3 + c] * ax.w0 + (int)r0[(size_t)x1 * 3 + c] * ax.w1;
                    const int h1 = (int)r1[(size_t)ax.s * 3 + c] * ax.w0 + (int)r1[(size_t)x1 * 3 + c] * ax.w1;
                    v[i][c] = ((((ay.w0 * (h0 >> 4)) >> 16) + ((ay.w1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
                }
            }
        }
    }
    const size_t plane = (size_t)S * S;
    float* o = img_out + (size_t)b * 3 * plane + (size_t)oy * S + ox0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                     // ToTensor, Normalize(0.5, 0.5)
        float4 f;
        f.x = ((float)v[0][c] / 255.0f - 0.5f) / 0.5f; f.y = ((float)v[1][c] / 255.0f - 0.5f) / 0.5f;
        f.z = ((float)v[2][c] / 255.0f - 0.5f) / 0.5f; f.w = ((float)v[3][c] / 255.0f - 0.5f) / 0.5f;
        *reinterpret_cast<float4*>(o + c * plane) = f;
    }
    if (img_u8) {
        uint8_t* u = img_u8 + ((size_t)b * plane + (size_t)oy * S + ox0) * 3;
        uint32_t w0 = (uint32_t)v[0][0] | ((uint32_t)v[0][1] << 8) | ((uint32_t)v[0][2] << 16) | ((uint32_t)v[1][0] << 24);
        uint32_t w1 = (uint32_t)v[1][1] | ((uint32_t)v[1][2] << 8) | ((uint32_t)v[2][0] << 16) | ((uint32_t)v[2][1] << 24);
        uint32_t w2 = (uint32_t)v[2][2] | ((uint32_t)v[3][0] << 8) | ((uint32_t)v[3][1] << 16) | ((uint32_t)v[3][2] << 24);
        uint32_t* uw = reinterpret_cast<uint32_t*>(u);                // 12 bytes, 4-byte aligned (ox0 % 4 == 0)
        uw[0] = w0; uw[1] = w1; uw[2] = w2;
    }
}
