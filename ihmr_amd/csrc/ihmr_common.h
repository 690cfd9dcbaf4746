// Shared device helpers for libihmr_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ihmr_hip.h"

#define NV IHMR_NUM_VERTS      // 778
#define NF IHMR_NUM_FACES      // 1538
#define NJ IHMR_NUM_JOINTS     // 16
#define NV3 (NV * 3)           // 2334
#define NPF 135                // pose-feature length (15 joints x 9)
#define NFP 1600               // faces padded to a multiple of 64 (SoA row length)
#define NVP 832                // vertices padded to a multiple of 64 (float4 basis rows)
#define SDF_G IHMR_SDF_GRID    // 32
#define SDF_NVOX (SDF_G * SDF_G * SDF_G)
#define WAVE 64

#define HIP_TRY(expr)                         \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

// Workgroup timeline (experiment builds only, -DIHMR_TIMELINE; scripts/timeline_wg.py): every workgroup of the instrumented kernels
// appends (kind, waves, start, end, hardware id) to a ring -- constant 100 MHz wall clock, the same on every CU -- from which the
// host reconstructs how many waves of which kernel were resident when: what the kernels of several streams really overlap
// (a rocprofv3 kernel trace serialises the dispatches it times and cannot show it).
#ifdef IHMR_TIMELINE
__device__ unsigned long long* g_tl_ring = nullptr;      // [TL_SEGS][seg_cap][3]
__device__ unsigned g_tl_cap = 0, g_tl_cursor[256 * 32];   // records per segment; one cursor per segment, 128 bytes apart (1.8 M atomics
                                                           // on ONE address took as long as the run: 12 M/s across the XCDs)
struct TlScope {
    unsigned long long t0; unsigned kind;
    __device__ __forceinline__ TlScope(unsigned k) : t0(wall_clock64()), kind(k) {}
    __device__ __forceinline__ ~TlScope() {
        if (threadIdx.x == 0 && g_tl_ring) {
            const unsigned seg = (blockIdx.x * 7u + kind * 41u) & 255u;
            const unsigned i = atomicAdd(&g_tl_cursor[seg * 32], 1u);
            if (i < g_tl_cap) {
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                unsigned long long* r = g_tl_ring + 3 * ((size_t)seg * g_tl_cap + i);
                r[0] = t0 | ((unsigned long long)kind << 58) | ((unsigned long long)(blockDim.x / 64) << 52);
                r[1] = wall_clock64();
                r[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            }
        }
    }
};
#define TL_SCOPE(k) TlScope tl_scope_(k)
#else
#define TL_SCOPE(k)
#endif

#include "ihmr_pure.h"        // the pure arithmetic helpers (host-compilable): DOT3, SDF rays / distances, Rodrigues, chain, optimizer step


struct ihmr_mano {
    // device pointers (fp32 unless stated)
    float* v_template;    // [2334]
    float* shapedirs_t;   // [10][2334]
    float* posedirs;      // [135][2334]
    float4* pd4;          // [135][NVP]  posedirs as (x,y,z,0) per vertex: one coalesced 16 B load per lane
    float4* sd4;          // [10][NVP]   shapedirs likewise
    float4* vt4;          // [NVP]       v_template likewise
    float* J_template;    // [48]   = J_regressor . v_template
    float* J_shapedirs;   // [48][10] = J_regressor . shapedirs
    float* weights;       // [778][16]
    float4* w4_w;         // [778] the (up to) four non-zero skinning weights of a vertex in joint order, zero padded ...
    uint32_t* w4_j;       // [778] ... and their joints, one byte each (padding: joint 0 with weight 0)
    int tail_fits;        // opt_tail_kernel's static + dynamic LDS fits this device (set by ihmr_mano_create; 0: three separate launches)
    int sparse4;          // every vertex has at most four non-zero weights (MANO's own weights do): the skinning loops run over w4_*
    float* pose_mean;     // [48]
    int32_t* parents;     // [16]
    int32_t* depth;       // [16] depth in the kinematic tree (root = 0)
    int32_t* tip_ids;     // [5]
    int32_t* wj_start;    // [17]  CSR by joint of the non-zero skinning weights
    int32_t* wj_vert;     // [nnz]
    float* wj_w;          // [nnz]
    int32_t* seg_q;       // [nseg+1] CSR split into single-joint segments of <= LBS_SEG entries (balanced dA reduction)
    int32_t* jseg_start;  // [17] first segment of each joint
    int nseg;
    int32_t* faces;       // [3][NFP] SoA, padded with face 0
    uint32_t* faces_pk;   // [NFP] the same as a | b << 10 | c << 20 (vertex ids < 1024): one load per triangle in the distance kernel
    float* J_regressor;   // [16][778] (host-side precompute source, kept for update_shapedirs)
    int max_depth;
    int nnz;
};

// wave64 min / max: 4 DPP row rotations (min over each 16-lane row, no LDS traffic), then the 4 row results
// through scalar registers.  Exact (min/max are order-independent); every lane gets the result.
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ float wave_reduce_min(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(4), 0xf, 0xf, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(2), 0xf, 0xf, false)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(1), 0xf, 0xf, false)));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_reduce_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(4), 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(2), 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(1), 0xf, 0xf, false)));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// fixed-order butterfly sum: every lane ends with the same value, bit-reproducible run to run
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// the same sum without LDS traffic: DPP row rotations, then lanes 0 / 16 / 32 / 48 through scalar registers (fixed order)
__device__ __forceinline__ float wave_reduce_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(4), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(2), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(1), 0xf, 0xf, false));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ unsigned wave_reduce_xor(unsigned v) {
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR(8), 0xf, 0xf, false);
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR(4), 0xf, 0xf, false);
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR(2), 0xf, 0xf, false);
    v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR(1), 0xf, 0xf, false);
    return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) ^ __builtin_amdgcn_readlane((int)v, 16) ^
                      __builtin_amdgcn_readlane((int)v, 32) ^ __builtin_amdgcn_readlane((int)v, 48));
}
// wave64 inclusive prefix sum of small ints: DPP row shifts inside the 16-lane rows, row totals via SGPRs
#define DPP_ROW_SHR(n) (0x110 + (n))
__device__ __forceinline__ int wave_incl_scan(int v, int& total) {
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, false);
    // across the four rows of 16: lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3 (two DPP adds; as four
    // readlanes + a select on the row index the compiler built a branch ladder)
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    total = __builtin_amdgcn_readlane(v, 63);
    return v;
}

// block-wide fixed-order sum through LDS (buf holds >= blockDim.x floats); result broadcast
__device__ __forceinline__ float block_reduce_sum(float v, float* buf) {
    const int tid = threadIdx.x;
    __syncthreads();
    buf[tid] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) buf[tid] += buf[tid + s];
        __syncthreads();
    }
    const float r = buf[0];
    __syncthreads();
    return r;
}

// n dwords global -> LDS by DMA (global_load_lds: no registers, asynchronous; the issuing wave's `s_waitcnt vmcnt(0)` + a workgroup barrier
// make them visible), by a group of `threads` consecutive threads of which this is thread `t`.  The LDS layout stays linear (the LDS
// address of a DMA load is a wave-uniform base + lane * 4).
__device__ __forceinline__ void lds_dma_dwords(const float* __restrict__ src, float* dst_lds, int n, int t, int threads) {
    const int lane = t % WAVE;
    for (int base = (t / WAVE) * WAVE; base < n; base += threads)
        if (base + lane < n)
            __builtin_amdgcn_global_load_lds(src + base + lane, (__attribute__((address_space(3))) void*)(dst_lds + base), 4, 0, 0);
}
