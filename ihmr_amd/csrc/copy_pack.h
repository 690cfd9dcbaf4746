// One launch instead of many small copies: batches of 2-D strided dword copies described by a segment table passed in the kernel
// arguments, and the root alignment of the exported annotation joints.
//
// Reference: MLPModel.set_input (models/mlp_model.py:120-170) moves 17 tensors to the device one `.cuda()` at a time and
// get_pred_result (:702-719) 13 tensors back one `.cpu()` at a time; rounds 1-4 of this build did the same with 17 + 13 copy
// launches (+ ~10 torch element-wise kernels for the joints' root alignment of the export, mlp_model.py:530-531): 21.9 % of the GPU
// time of an IHMR-MLP test() batch (profiles/r4_v6_mlp_kernel_stats.csv).  Now: one packed host-to-device copy (host inputs) and one
// segment launch in, one root-alignment launch + one segment launch + one device-to-host copy out.
#pragma once
#include "ihmr_common.h"

struct CopySegTable { ihmr_copy_seg s[IHMR_COPY_MAX_SEGS]; };

// grid = (blocks per segment, segments), block = 256: dst[r * dst_ld + c] = src[r * src_ld + c] for r < rows, c < width (dwords)
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegTable t) {
    const ihmr_copy_seg g = t.s[blockIdx.y];
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(g.src);
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(g.dst);
    const long total = (long)g.rows * g.width;
    const bool flat = g.src_ld == g.width && g.dst_ld == g.width;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (flat) { dst[i] = src[i]; continue; }
        const long r = i / g.width, c = i - r * g.width;
        dst[r * g.dst_ld + c] = src[r * g.src_ld + c];
    }
}

// mlp_model.py:530-531 + loss_utils.py:90-98: the export's GT joints are the annotation root-aligned IN PLACE by the 3-D loss: every
// joint minus the root joint of its sample (root = joint 0 if its weight > 0.5, joint 21 if < 1e-7, none otherwise); the weights
// column is copied.  grid = B, block = 64 (lane j < 42 owns joint j).
__global__ __launch_bounds__(64) void root_align_joints_kernel(const float* __restrict__ joints4, float* __restrict__ out4, int B) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= 42) return;
    const float* src = joints4 + (size_t)b * 42 * 4;
    const float w0 = src[3];
    const int root = w0 > 0.5f ? 0 : (w0 < 1e-7f ? 21 : -1);
    const float has = root >= 0 ? 1.0f : 0.0f;
    const int rr = root >= 0 ? root : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) out4[((size_t)b * 42 + j) * 4 + k] = src[j * 4 + k] - src[rr * 4 + k] * has;   // (no root: minus an exact zero)
    out4[((size_t)b * 42 + j) * 4 + 3] = src[j * 4 + 3];
}
