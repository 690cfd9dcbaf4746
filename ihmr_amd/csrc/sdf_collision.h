// Two-hand collision (penetration) term: sparse voxel signed-distance evaluation + trilinear sampling.
//
// Stands behind the reference's third-party `sdf.SDFLoss` module (models/loss_utils.py:13,38,181-182).
// Semantics = DESIGN.md "SDF arithmetic spec" (identical to oracle/sdf_grid.c + oracle/sdf_ref.py):
// per hand a 32^3 grid phi over the box-normalised mesh, phi = distance to the surface for voxel
// centres inside the mesh (odd +x ray crossings), 0 outside; the OTHER hand's vertices sample it
// trilinearly (zeros padding, align_corners = False).
//
// MI355X design (results identical to the dense grid, bit for bit):
//   * only voxels that a sample actually reads are evaluated (<= 8 corners per query vertex,
//     collected in a 1024 x 32-bit mask per hand);
//   * all voxels of a (k,j) column share the +x ray, so the (u,v) triangle test is done once per
//     column -- one wave per column, lanes across the 1538 triangles, coalesced SoA reads of a
//     per-iteration triangle table -- and only the surviving candidates are tested per voxel (t > 0);
//   * the distance of an inside voxel is a wave-level min-reduction: lanes across triangles,
//     culled by a bounding-sphere lower bound against a wave-wide upper bound (exact: a culled
//     triangle can never be the minimum).
#pragma once
#include "ihmr_common.h"

#define SDF_THREADS 256
#define SDF_NCOL (SDF_G * SDF_G)   // 1024 columns (k,j)
#define SDF_TRI_ROWS 20            // per-hand triangle table rows (SoA, NFP floats each)
// rows: 0-8 a,b,c (xyz each) | 9 ay 10 az 11 e1y 12 e1z 13 e2y 14 e2z 15 inv_det | 16-18 centroid 19 radius
#define SDF_BIN_CAP 32768          // (triangle, column) pairs per hand before falling back to a full scan
#define SDF_EVAL_CHUNKS 8          // workgroups per hand in the eval kernel

struct SdfWorkspace {          // carved from the caller's workspace, per hand (H = 2B hands)
    float* box;                // [H][4]  centre xyz, scale
    float* tri;                // [H][SDF_TRI_ROWS][NFP]
    float* phi;                // [H][32768]  (only `needed` entries are defined)
    unsigned* needed;          // [H][1024] bitmask over i per column (k*32+j)
    int* col_off;              // [H][1025] start of each column's triangle list (exclusive prefix)
    unsigned short* col_tris;  // [H][SDF_BIN_CAP] triangle ids binned by column
    unsigned short* vox_list;  // [H][32768] needed voxels, id = col*32 + i
    int* counts;               // [H][4]: 0 = #needed voxels, 1 = #binned pairs (> SDF_BIN_CAP => overflow)
    unsigned* inside_list;     // [H*32768] inside voxels of the whole batch: (hand << 16) | voxel id, grouped by workgroup
    int* inside_count;         // [1] (zeroed by the prep kernel of hand 0 ... see sdf_launch)
    unsigned long long* stats; // [8] optional work counters
};

__host__ __device__ inline size_t sdf_ws_bytes(int H) {
    size_t n = 0;
    n += (size_t)H * 4 * sizeof(float);
    n += (size_t)H * SDF_TRI_ROWS * NFP * sizeof(float);
    n += (size_t)H * SDF_NVOX * sizeof(float);
    n += (size_t)H * SDF_NCOL * sizeof(unsigned);
    n += (size_t)H * 1028 * sizeof(int);
    n += (size_t)H * SDF_BIN_CAP * sizeof(unsigned short);
    n += (size_t)H * SDF_NVOX * sizeof(unsigned short);
    n += (size_t)H * 4 * sizeof(int);
    n += (size_t)H * SDF_NVOX * sizeof(unsigned);
    n += 64 + 64 + 256;
    return (n + 255) & ~(size_t)255;
}

static inline SdfWorkspace sdf_carve(void* ws, int H) {
    SdfWorkspace w;
    char* p = (char*)ws;
    w.box = (float*)p; p += (size_t)H * 4 * sizeof(float);
    w.tri = (float*)p; p += (size_t)H * SDF_TRI_ROWS * NFP * sizeof(float);
    w.phi = (float*)p; p += (size_t)H * SDF_NVOX * sizeof(float);
    w.needed = (unsigned*)p; p += (size_t)H * SDF_NCOL * sizeof(unsigned);
    w.col_off = (int*)p; p += (size_t)H * 1028 * sizeof(int);
    w.counts = (int*)p; p += (size_t)H * 4 * sizeof(int);
    w.stats = (unsigned long long*)p; p += 64;
    w.inside_count = (int*)p; p += 64;
    w.inside_list = (unsigned*)p; p += (size_t)H * SDF_NVOX * sizeof(unsigned);
    w.col_tris = (unsigned short*)p; p += (size_t)H * SDF_BIN_CAP * sizeof(unsigned short);
    w.vox_list = (unsigned short*)p;
    return w;
}

// vertex addressing: hand (b, hnd) of a batch stored with arbitrary strides (floats)
struct VertLayout {
    const float* base;
    long stride_b, stride_h;
    __device__ __forceinline__ const float* hand(int b, int hnd) const { return base + b * stride_b + hnd * stride_h; }
};

// grid_sample un-normalisation, align_corners = False: ((x + 1) * G - 1) / 2
__device__ __forceinline__ float sdf_unnorm(float x) { return ((x + 1.0f) * (float)SDF_G - 1.0f) / 2.0f; }

// exclusive prefix sum of data[0..1023] (LDS) with 256 threads; returns the total.  scratch: >= 8 ints (LDS)
__device__ __forceinline__ int block_excl_scan_1024(int* data, int* scratch) {
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    __syncthreads();
    const int a0 = data[4 * tid], a1 = data[4 * tid + 1], a2 = data[4 * tid + 2], a3 = data[4 * tid + 3];
    const int mine = a0 + a1 + a2 + a3;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == WAVE - 1) scratch[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < SDF_THREADS / WAVE; ++w) {
        if (w < wave) base += scratch[w];
        total += scratch[w];
    }
    const int ex = base + inc - mine;
    data[4 * tid] = ex;
    data[4 * tid + 1] = ex + a0;
    data[4 * tid + 2] = ex + a0 + a1;
    data[4 * tid + 3] = ex + a0 + a1 + a2;
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------------------------- prep
// grid = 2B (hand id H = 2*b + hnd), block = 256: box, triangle table, needed-voxel mask + list,
// triangles binned by (k,j) column (conservative yz bounding box, only for needed columns).
template <bool DENSE>
__global__ __launch_bounds__(SDF_THREADS) void sdf_prep_kernel(VertLayout vl, const int32_t* __restrict__ faces_r,
                                                               const int32_t* __restrict__ faces_l, SdfWorkspace ws) {
    __shared__ float vn[NV3];
    __shared__ float red[6][SDF_THREADS];
    __shared__ unsigned needed[SDF_NCOL];
    __shared__ int cnt[SDF_NCOL];
    __shared__ int cur[SDF_NCOL];
    __shared__ float box[4];
    __shared__ int scratch[8];
    const int H = blockIdx.x, b = H >> 1, hnd = H & 1, tid = threadIdx.x;
    const float* own = vl.hand(b, hnd);
    const float* other = vl.hand(b, 1 - hnd);
    const int32_t* faces = hnd == 0 ? faces_r : faces_l;  // SoA [3][NFP]

    // ---- bounding box (min / max are exact, any order)
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = tid; v < NV; v += SDF_THREADS) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = own[3 * v + k];
            vn[3 * v + k] = x;
            mn[k] = fminf(mn[k], x);
            mx[k] = fmaxf(mx[k], x);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { red[k][tid] = mn[k]; red[3 + k][tid] = mx[k]; }
    __syncthreads();
    for (int s = SDF_THREADS >> 1; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                red[k][tid] = fminf(red[k][tid], red[k][tid + s]);
                red[3 + k][tid] = fmaxf(red[3 + k][tid], red[3 + k][tid + s]);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float ex = red[3][0] - red[0][0], ey = red[4][0] - red[1][0], ez = red[5][0] - red[2][0];
        box[0] = (red[0][0] + red[3][0]) * 0.5f;
        box[1] = (red[1][0] + red[4][0]) * 0.5f;
        box[2] = (red[2][0] + red[5][0]) * 0.5f;
        box[3] = 0.6f * fmaxf(ex, fmaxf(ey, ez));  // (1 + 0.2) * 0.5 * max extent
    }
    for (int i = tid; i < SDF_NCOL; i += SDF_THREADS) { needed[i] = DENSE ? 0xffffffffu : 0u; cnt[i] = 0; }
    __syncthreads();
    const float cx = box[0], cy = box[1], cz = box[2], sc = box[3];
    if (tid < 4) ws.box[H * 4 + tid] = box[tid];

    // ---- normalise own vertices into [-1,1]^3
    for (int i = tid; i < NV3; i += SDF_THREADS) {
        const int k = i % 3;
        vn[i] = (vn[i] - (k == 0 ? cx : (k == 1 ? cy : cz))) / sc;
    }
    // ---- which voxels will the other hand's vertices read?
    if (!DENSE) {
        for (int v = tid; v < NV; v += SDF_THREADS) {
            const float qx = (other[3 * v] - cx) / sc, qy = (other[3 * v + 1] - cy) / sc, qz = (other[3 * v + 2] - cz) / sc;
            const float ix = sdf_unnorm(qx), iy = sdf_unnorm(qy), iz = sdf_unnorm(qz);
            const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
            // completely outside the grid (or non-finite): contributes nothing
            if (!(fx >= -1.0f && fx <= (float)(SDF_G - 1) && fy >= -1.0f && fy <= (float)(SDF_G - 1) && fz >= -1.0f &&
                  fz <= (float)(SDF_G - 1)))
                continue;
            const int i0 = (int)fx, j0 = (int)fy, k0 = (int)fz;
            unsigned mi = 0;
            if (i0 >= 0) mi |= 1u << i0;
            if (i0 + 1 < SDF_G) mi |= 1u << (i0 + 1);
#pragma unroll
            for (int dk = 0; dk < 2; ++dk)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const int k = k0 + dk, j = j0 + dj;
                    if (k >= 0 && k < SDF_G && j >= 0 && j < SDF_G) atomicOr(&needed[k * SDF_G + j], mi);
                }
        }
    }
    __syncthreads();

    // ---- per-iteration triangle table (SoA rows of NFP floats) + column binning, pass 1 (count)
    float* T = ws.tri + (size_t)H * SDF_TRI_ROWS * NFP;
    for (int f = tid; f < NFP; f += SDF_THREADS) {
        const int fa = faces[f], fb = faces[NFP + f], fc = faces[2 * NFP + f];
        const float a[3] = {vn[3 * fa], vn[3 * fa + 1], vn[3 * fa + 2]};
        const float bb[3] = {vn[3 * fb], vn[3 * fb + 1], vn[3 * fb + 2]};
        const float c[3] = {vn[3 * fc], vn[3 * fc + 1], vn[3 * fc + 2]};
#pragma unroll
        for (int k = 0; k < 3; ++k) { T[k * NFP + f] = a[k]; T[(3 + k) * NFP + f] = bb[k]; T[(6 + k) * NFP + f] = c[k]; }
        const float e1y = bb[1] - a[1], e1z = bb[2] - a[2], e2y = c[1] - a[1], e2z = c[2] - a[2];
        const float det = __builtin_fmaf(e1z, e2y, -(e1y * e2z));
        const bool ok = f < NF && fabsf(det) >= 1e-12f;
        T[9 * NFP + f] = a[1]; T[10 * NFP + f] = a[2];
        T[11 * NFP + f] = e1y; T[12 * NFP + f] = e1z; T[13 * NFP + f] = e2y; T[14 * NFP + f] = e2z;
        T[15 * NFP + f] = ok ? 1.0f / det : __builtin_nanf("");  // NaN => never a hit
        // bounding sphere about the centroid (conservative radius)
        const float gx = (a[0] + bb[0] + c[0]) * (1.0f / 3.0f), gy = (a[1] + bb[1] + c[1]) * (1.0f / 3.0f),
                    gz = (a[2] + bb[2] + c[2]) * (1.0f / 3.0f);
        float r2 = 0.f;
        {
            float dx = a[0] - gx, dy = a[1] - gy, dz = a[2] - gz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
            dx = bb[0] - gx; dy = bb[1] - gy; dz = bb[2] - gz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
            dx = c[0] - gx; dy = c[1] - gy; dz = c[2] - gz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
        }
        T[16 * NFP + f] = gx; T[17 * NFP + f] = gy; T[18 * NFP + f] = gz;
        T[19 * NFP + f] = f < NF ? sqrtf(r2) * 1.0001f + 1e-6f : -1.0f;  // radius < 0 marks padding
        if (ok) {
            // columns whose ray (py, pz) can cross the triangle: centres inside the yz bounding box (+ margin)
            const float ymin = fminf(a[1], fminf(bb[1], c[1])) - 1e-4f, ymax = fmaxf(a[1], fmaxf(bb[1], c[1])) + 1e-4f;
            const float zmin = fminf(a[2], fminf(bb[2], c[2])) - 1e-4f, zmax = fmaxf(a[2], fmaxf(bb[2], c[2])) + 1e-4f;
            const int j0 = max(0, (int)ceilf((ymin + 1.0f) * 16.0f - 0.5f)), j1 = min(SDF_G - 1, (int)floorf((ymax + 1.0f) * 16.0f - 0.5f));
            const int k0 = max(0, (int)ceilf((zmin + 1.0f) * 16.0f - 0.5f)), k1 = min(SDF_G - 1, (int)floorf((zmax + 1.0f) * 16.0f - 0.5f));
            for (int k = k0; k <= k1; ++k)
                for (int j = j0; j <= j1; ++j)
                    if (needed[k * SDF_G + j]) atomicAdd(&cnt[k * SDF_G + j], 1);
        }
    }
    const int total_pairs = block_excl_scan_1024(cnt, scratch);
    int* goff = ws.col_off + (size_t)H * 1028;
    for (int i = tid; i < SDF_NCOL; i += SDF_THREADS) { goff[i] = cnt[i]; cur[i] = 0; }
    if (tid == 0) { goff[SDF_NCOL] = total_pairs; ws.counts[H * 4 + 1] = total_pairs; }
    __syncthreads();
    // ---- binning pass 2 (fill); list order inside a column is irrelevant (parity is an XOR)
    if (total_pairs <= SDF_BIN_CAP) {
        unsigned short* lst = ws.col_tris + (size_t)H * SDF_BIN_CAP;
        for (int f = tid; f < NF; f += SDF_THREADS) {
            const float inv = T[15 * NFP + f];
            if (inv != inv) continue;
            const int fa = faces[f], fb = faces[NFP + f], fc = faces[2 * NFP + f];
            const float ay = vn[3 * fa + 1], az = vn[3 * fa + 2], by = vn[3 * fb + 1], bz = vn[3 * fb + 2], cy2 = vn[3 * fc + 1],
                        cz2 = vn[3 * fc + 2];
            const float ymin = fminf(ay, fminf(by, cy2)) - 1e-4f, ymax = fmaxf(ay, fmaxf(by, cy2)) + 1e-4f;
            const float zmin = fminf(az, fminf(bz, cz2)) - 1e-4f, zmax = fmaxf(az, fmaxf(bz, cz2)) + 1e-4f;
            const int j0 = max(0, (int)ceilf((ymin + 1.0f) * 16.0f - 0.5f)), j1 = min(SDF_G - 1, (int)floorf((ymax + 1.0f) * 16.0f - 0.5f));
            const int k0 = max(0, (int)ceilf((zmin + 1.0f) * 16.0f - 0.5f)), k1 = min(SDF_G - 1, (int)floorf((zmax + 1.0f) * 16.0f - 0.5f));
            for (int k = k0; k <= k1; ++k)
                for (int j = j0; j <= j1; ++j) {
                    const int col = k * SDF_G + j;
                    if (needed[col]) lst[cnt[col] + atomicAdd(&cur[col], 1)] = (unsigned short)f;
                }
        }
    }
    // ---- needed-voxel list (order = voxel id, deterministic)
    unsigned* gneeded = ws.needed + (size_t)H * SDF_NCOL;
    __syncthreads();
    for (int i = tid; i < SDF_NCOL; i += SDF_THREADS) { gneeded[i] = needed[i]; cur[i] = __popc(needed[i]); }
    const int nvox = block_excl_scan_1024(cur, scratch);
    unsigned short* vlist = ws.vox_list + (size_t)H * SDF_NVOX;
    for (int col = tid; col < SDF_NCOL; col += SDF_THREADS) {
        unsigned m = needed[col];
        int o = cur[col];
        while (m) {
            const int i = __ffs((int)m) - 1;
            m &= m - 1;
            vlist[o++] = (unsigned short)(col * SDF_G + i);
        }
    }
    if (tid == 0) ws.counts[H * 4] = nvox;
}

// squared distance point -> triangle, closest point by Voronoi region (same operation order as
// oracle/sdf_grid.c point_tri_dist2, branch-free selects)
__device__ __forceinline__ float sdf_point_tri_dist2(const float* a, const float* b, const float* c, float px, float py,
                                                     float pz) {
    const float abx = b[0] - a[0], aby = b[1] - a[1], abz = b[2] - a[2];
    const float acx = c[0] - a[0], acy = c[1] - a[1], acz = c[2] - a[2];
    const float apx = px - a[0], apy = py - a[1], apz = pz - a[2];
    const float d1 = DOT3(abx, aby, abz, apx, apy, apz);
    const float d2 = DOT3(acx, acy, acz, apx, apy, apz);
    const float bpx = px - b[0], bpy = py - b[1], bpz = pz - b[2];
    const float d3 = DOT3(abx, aby, abz, bpx, bpy, bpz);
    const float d4 = DOT3(acx, acy, acz, bpx, bpy, bpz);
    const float vc = __builtin_fmaf(d1, d4, -(d3 * d2));
    const float cpx = px - c[0], cpy = py - c[1], cpz = pz - c[2];
    const float d5 = DOT3(abx, aby, abz, cpx, cpy, cpz);
    const float d6 = DOT3(acx, acy, acz, cpx, cpy, cpz);
    const float vb = __builtin_fmaf(d5, d2, -(d1 * d6));
    const float va = __builtin_fmaf(d3, d6, -(d5 * d4));
    float qx, qy, qz;
    if (d1 <= 0.0f && d2 <= 0.0f) {
        qx = a[0]; qy = a[1]; qz = a[2];
    } else if (d3 >= 0.0f && d4 <= d3) {
        qx = b[0]; qy = b[1]; qz = b[2];
    } else if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        qx = __builtin_fmaf(v, abx, a[0]); qy = __builtin_fmaf(v, aby, a[1]); qz = __builtin_fmaf(v, abz, a[2]);
    } else if (d6 >= 0.0f && d5 <= d6) {
        qx = c[0]; qy = c[1]; qz = c[2];
    } else if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float w = d2 / (d2 - d6);
        qx = __builtin_fmaf(w, acx, a[0]); qy = __builtin_fmaf(w, acy, a[1]); qz = __builtin_fmaf(w, acz, a[2]);
    } else if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        qx = __builtin_fmaf(w, c[0] - b[0], b[0]); qy = __builtin_fmaf(w, c[1] - b[1], b[1]);
        qz = __builtin_fmaf(w, c[2] - b[2], b[2]);
    } else {
        const float denom = 1.0f / (va + vb + vc);
        const float v = vb * denom, w = vc * denom;
        qx = __builtin_fmaf(acx, w, __builtin_fmaf(abx, v, a[0]));
        qy = __builtin_fmaf(acy, w, __builtin_fmaf(aby, v, a[1]));
        qz = __builtin_fmaf(acz, w, __builtin_fmaf(abz, v, a[2]));
    }
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    return DOT3(dx, dy, dz, dx, dy, dz);
}

// ------------------------------------------------------------------------------------- parity
// grid = (SDF_EVAL_CHUNKS, 2B), block = 256: one lane per needed voxel -- +x ray parity against the
// triangles binned to its column.  Outside voxels get phi = 0 here; inside voxels are appended to the
// batch-wide list (one atomic per workgroup, so a workgroup's entries -- all of one hand -- stay adjacent).
__global__ __launch_bounds__(SDF_THREADS) void sdf_parity_kernel(SdfWorkspace ws, int collect_stats) {
    __shared__ unsigned short inside_loc[SDF_NVOX / SDF_EVAL_CHUNKS + SDF_THREADS];
    __shared__ int n_inside, g_base;
    const int H = blockIdx.y, tid = threadIdx.x, lane = tid % WAVE;
    const float* T = ws.tri + (size_t)H * SDF_TRI_ROWS * NFP;
    const int nvox = ws.counts[H * 4];
    const bool overflow = ws.counts[H * 4 + 1] > SDF_BIN_CAP;
    const int* coff = ws.col_off + (size_t)H * 1028;
    const unsigned short* ctris = ws.col_tris + (size_t)H * SDF_BIN_CAP;
    const unsigned short* vlist = ws.vox_list + (size_t)H * SDF_NVOX;
    float* phi = ws.phi + (size_t)H * SDF_NVOX;
    if (tid == 0) n_inside = 0;
    __syncthreads();
    unsigned long long st_tests = 0;
    // interleaved assignment so every workgroup sees a uniform sample of the hand's voxels
    for (int q = tid * gridDim.x + blockIdx.x; q < nvox; q += SDF_THREADS * gridDim.x) {
        const int id = vlist[q], col = id >> 5, i = id & 31, k = col >> 5, j = col & 31;
        const float px = (float)(2 * i + 1) / (float)SDF_G - 1.0f;
        const float py = (float)(2 * j + 1) / (float)SDF_G - 1.0f;
        const float pz = (float)(2 * k + 1) / (float)SDF_G - 1.0f;
        const int t0 = overflow ? 0 : coff[col], t1 = overflow ? NF : coff[col + 1];
        int hits = 0;
        for (int t = t0; t < t1; ++t) {
            const int f = overflow ? t : (int)ctris[t];
            const float inv = T[15 * NFP + f];
            const float ay = T[9 * NFP + f], az = T[10 * NFP + f];
            const float e1y = T[11 * NFP + f], e1z = T[12 * NFP + f], e2y = T[13 * NFP + f], e2z = T[14 * NFP + f];
            const float sy = py - ay, sz = pz - az;
            const float u = __builtin_fmaf(sz, e2y, -(sy * e2z)) * inv;
            const float qx = __builtin_fmaf(sy, e1z, -(sz * e1y));
            const float v = qx * inv;
            if ((u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f)) {
                const float ax = T[0 * NFP + f];
                const float e1x = T[3 * NFP + f] - ax, e2x = T[6 * NFP + f] - ax;
                const float sx = px - ax;
                const float qy = __builtin_fmaf(sz, e1x, -(sx * e1z));
                const float qz = __builtin_fmaf(sx, e1y, -(sy * e1x));
                const float tt = DOT3(e2x, e2y, e2z, qx, qy, qz) * inv;
                hits += tt > 0.0f ? 1 : 0;
            }
            st_tests += 1;
        }
        if (hits & 1) inside_loc[atomicAdd(&n_inside, 1)] = (unsigned short)id;
        else phi[id] = 0.0f;
    }
    __syncthreads();
    const int nin = n_inside;
    if (tid == 0) g_base = nin > 0 ? atomicAdd(ws.inside_count, nin) : 0;
    __syncthreads();
    for (int q = tid; q < nin; q += SDF_THREADS) ws.inside_list[g_base + q] = ((unsigned)H << 16) | (unsigned)inside_loc[q];
    if (collect_stats) {
        unsigned long long c = st_tests;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) atomicAdd(&ws.stats[0], c);                                              // ray tests
        if (tid == 0) atomicAdd(&ws.stats[2], (unsigned long long)nin);                         // inside voxels
        if (tid == 0 && blockIdx.x == 0) atomicAdd(&ws.stats[3], (unsigned long long)nvox);     // needed voxels
    }
}

// ------------------------------------------------------------------------------------- distance
// grid = SDF_DIST_BLOCKS, block = 256 (4 waves).  Every wave takes a contiguous slice of the batch-wide
// inside-voxel list; per voxel: exact min distance over the mesh as a wave-level min-reduction (lanes
// across triangles).  The 1538 bounding spheres of the current hand live in registers (25 per lane) and are
// reloaded only when the slice crosses into another hand; survivors of the sphere cull are compacted
// through a per-wave LDS list so the expensive closest-point evaluation runs on dense lanes.
#define SDF_DIST_BLOCKS 512
__global__ __launch_bounds__(SDF_THREADS) void sdf_dist_kernel(SdfWorkspace ws, int collect_stats) {
    __shared__ unsigned short surv[SDF_THREADS / WAVE][NFP];
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    const int total = *ws.inside_count;
    const int nwaves = gridDim.x * (SDF_THREADS / WAVE), gw = blockIdx.x * (SDF_THREADS / WAVE) + wave;
    const int per = (total + nwaves - 1) / nwaves;
    const int q0 = gw * per, q1 = min(total, q0 + per);
    unsigned short* mylist = surv[wave];
    float sx[NFP / WAVE], sy[NFP / WAVE], sz[NFP / WAVE], sr[NFP / WAVE];
    int curH = -1;
    const float* T = nullptr;
    unsigned long long st_dist = 0;
    for (int q = q0; q < q1; ++q) {
        const unsigned ent = ws.inside_list[q];
        const int H = (int)(ent >> 16), id = (int)(ent & 0xffffu);
        if (H != curH) {
            curH = H;
            T = ws.tri + (size_t)H * SDF_TRI_ROWS * NFP;
#pragma unroll
            for (int t = 0; t < NFP / WAVE; ++t) {
                const int f = lane + WAVE * t;
                sx[t] = T[16 * NFP + f]; sy[t] = T[17 * NFP + f]; sz[t] = T[18 * NFP + f]; sr[t] = T[19 * NFP + f];
            }
        }
        const int col = id >> 5, i = id & 31, k = col >> 5, j = col & 31;
        const float px = (float)(2 * i + 1) / (float)SDF_G - 1.0f;
        const float py = (float)(2 * j + 1) / (float)SDF_G - 1.0f;
        const float pz = (float)(2 * k + 1) / (float)SDF_G - 1.0f;
        float d2[NFP / WAVE];
        float ub2 = INFINITY;
#pragma unroll
        for (int t = 0; t < NFP / WAVE; ++t) {
            const float dx = px - sx[t], dy = py - sy[t], dz = pz - sz[t];
            d2[t] = dx * dx + dy * dy + dz * dz;
            if (sr[t] >= 0.0f) ub2 = fminf(ub2, d2[t]);
        }
        ub2 = wave_reduce_min(ub2);  // the centroid is a point of the triangle: dist <= |p - centroid|
        const float ub_lim = sqrtf(ub2) * 1.0001f + 1e-6f;
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < NFP / WAVE; ++t) {
            const float lim = ub_lim + sr[t];
            // cull iff |p - centroid| - radius > upper bound (exact: such a triangle cannot be the minimum)
            const bool keep = (sr[t] >= 0.0f) && !(d2[t] > lim * lim * 1.00001f);
            const unsigned long long bal = __ballot(keep);
            if (keep) mylist[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(lane + WAVE * t);
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float best = INFINITY;
        for (int sidx = lane; sidx < cnt; sidx += WAVE) {
            const int f = mylist[sidx];
            const float a[3] = {T[0 * NFP + f], T[1 * NFP + f], T[2 * NFP + f]};
            const float b[3] = {T[3 * NFP + f], T[4 * NFP + f], T[5 * NFP + f]};
            const float c[3] = {T[6 * NFP + f], T[7 * NFP + f], T[8 * NFP + f]};
            best = fminf(best, sdf_point_tri_dist2(a, b, c, px, py, pz));
            st_dist += 1;
        }
        best = wave_reduce_min(best);
        if (lane == 0) ws.phi[(size_t)H * SDF_NVOX + id] = sqrtf(best);
        __builtin_amdgcn_wave_barrier();
    }
    if (collect_stats) {
        unsigned long long d = st_dist;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) atomicAdd(&ws.stats[1], d);  // exact point-triangle distances
    }
}

// ------------------------------------------------------------------------------------- sample
// grid = B, block = 256.  Entry e = hnd*778 + v samples phi of hand `hnd` at vertex v of hand 1-hnd.
// Writes per_vert / origin_scale (B,1556), dval (B,1556,3) = d per_vert / d vertex, loss (B).
// If gverts != nullptr: fused-path gradient gverts[(1-hnd), b, v, :] = gscale[b] * dval  (layout (2,B,778,3)).
__global__ __launch_bounds__(SDF_THREADS) void sdf_sample_kernel(VertLayout vl, SdfWorkspace ws, float robustifier,
                                                                 float* __restrict__ loss, float* __restrict__ per_vert,
                                                                 float* __restrict__ origin, float* __restrict__ dval,
                                                                 float* __restrict__ gverts, int B,
                                                                 const float* __restrict__ gscale) {
    __shared__ float red[SDF_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    const float gs = gscale ? gscale[b] : 0.f;
    for (int e = tid; e < 2 * NV; e += SDF_THREADS) {
        const int hnd = e / NV, v = e % NV;
        const int H = 2 * b + hnd;
        const float cx = ws.box[H * 4], cy = ws.box[H * 4 + 1], cz = ws.box[H * 4 + 2], sc = ws.box[H * 4 + 3];
        const float* q = vl.hand(b, 1 - hnd) + 3 * v;
        const float ix = sdf_unnorm((q[0] - cx) / sc), iy = sdf_unnorm((q[1] - cy) / sc), iz = sdf_unnorm((q[2] - cz) / sc);
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
        float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        if (x0 >= -1.0f && x0 <= (float)(SDF_G - 1) && y0 >= -1.0f && y0 <= (float)(SDF_G - 1) && z0 >= -1.0f &&
            z0 <= (float)(SDF_G - 1)) {
            const int i0 = (int)x0, j0 = (int)y0, k0 = (int)z0;
            const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
            const float wx1 = fx, wx0 = (x0 + 1.0f) - ix, wy1 = fy, wy0 = (y0 + 1.0f) - iy, wz1 = fz, wz0 = (z0 + 1.0f) - iz;
            const float* phi = ws.phi + (size_t)H * SDF_NVOX;
#pragma unroll
            for (int dk = 0; dk < 2; ++dk)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int di = 0; di < 2; ++di) {
                        const int i = i0 + di, j = j0 + dj, k = k0 + dk;
                        if (i >= 0 && i < SDF_G && j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                            const float p = phi[(k * SDF_G + j) * SDF_G + i];
                            const float wx = di ? wx1 : wx0, wy = dj ? wy1 : wy0, wz = dk ? wz1 : wz0;
                            val += p * (wx * wy * wz);
                            gx += (di ? p : -p) * (wy * wz);
                            gy += (dj ? p : -p) * (wx * wz);
                            gz += (dk ? p : -p) * (wx * wy);
                        }
                    }
        }
        // chain: ix = ((x+1)*G - 1)/2, x = (q - c)/s  =>  d ix / d q = G / (2 s)
        const float chain = (0.5f * (float)SDF_G) / sc;
        gx *= chain; gy *= chain; gz *= chain;
        if (robustifier > 0.f) {
            const float r = val / robustifier, fr = r * r;
            const float dfr = 2.0f * r / robustifier;       // d fr / d val
            const float dout = dfr / ((fr + 1.0f) * (fr + 1.0f));
            val = fr / (fr + 1.0f);
            gx *= dout; gy *= dout; gz *= dout;
        }
        per_vert[(size_t)b * 2 * NV + e] = val;
        origin[(size_t)b * 2 * NV + e] = val * sc;
        if (dval) {
            float* d = dval + ((size_t)b * 2 * NV + e) * 3;
            d[0] = gx; d[1] = gy; d[2] = gz;
        }
        if (gverts) {
            float* g = gverts + (((size_t)(1 - hnd) * B + b) * NV + v) * 3;
            g[0] = gs * gx; g[1] = gs * gy; g[2] = gs * gz;
        }
        acc += val;
    }
    const float tot = block_reduce_sum(acc, red);
    if (tid == 0) loss[b] = tot / 4.0f;  // parent project: sum / num_hands^2
}
