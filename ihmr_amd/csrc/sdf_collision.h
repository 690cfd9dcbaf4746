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
// rows: 0-8 a,b,c (xyz each) | 9 ay 10 az 11 e1y 12 e1z 13 e2y 14 e2z 15 inv_det | 16-18 sphere centre 19 radius

struct SdfWorkspace {          // carved from the caller's workspace, per hand (H = 2B hands)
    float* box;                // [H][4]  centre xyz, scale
    float* tri;                // [H][SDF_TRI_ROWS][NFP]
    unsigned* needed;          // [H][1024] bitmask over i per column (k*32+j)
    unsigned short* col_list;  // [H][1024]
    int* col_count;            // [H]
    float* phi;                // [H][32768]  (only `needed` entries are defined)
    unsigned long long* stats; // [4] optional work counters (columns, candidate tests, inside voxels, dist evals)
};

__host__ __device__ inline size_t sdf_ws_bytes(int H) {
    size_t n = 0;
    n += (size_t)H * 4 * sizeof(float);
    n += (size_t)H * SDF_TRI_ROWS * NFP * sizeof(float);
    n += (size_t)H * SDF_NCOL * sizeof(unsigned);
    n += (size_t)H * SDF_NCOL * sizeof(unsigned short);
    n += (size_t)H * sizeof(int);
    n += (size_t)H * SDF_NVOX * sizeof(float);
    n += 64;
    return (n + 255) & ~(size_t)255;
}

static inline SdfWorkspace sdf_carve(void* ws, int H) {
    SdfWorkspace w;
    char* p = (char*)ws;
    w.box = (float*)p; p += (size_t)H * 4 * sizeof(float);
    w.tri = (float*)p; p += (size_t)H * SDF_TRI_ROWS * NFP * sizeof(float);
    w.phi = (float*)p; p += (size_t)H * SDF_NVOX * sizeof(float);
    w.needed = (unsigned*)p; p += (size_t)H * SDF_NCOL * sizeof(unsigned);
    w.col_count = (int*)p; p += (size_t)H * sizeof(int);
    w.stats = (unsigned long long*)p; p += 32;
    w.col_list = (unsigned short*)p;
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

// ------------------------------------------------------------------------------------- prep
// grid = 2B (hand id H = 2*b + hnd), block = 256: box, triangle table, needed-voxel mask, column list.
template <bool DENSE>
__global__ __launch_bounds__(SDF_THREADS) void sdf_prep_kernel(VertLayout vl, const int32_t* __restrict__ faces_r,
                                                               const int32_t* __restrict__ faces_l, SdfWorkspace ws) {
    __shared__ float vn[NV3];
    __shared__ float red[6][SDF_THREADS];
    __shared__ unsigned needed[SDF_NCOL];
    __shared__ float box[4];
    __shared__ int wave_cnt[SDF_THREADS / WAVE + 1];
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
    for (int i = tid; i < SDF_NCOL; i += SDF_THREADS) needed[i] = DENSE ? 0xffffffffu : 0u;
    __syncthreads();
    const float cx = box[0], cy = box[1], cz = box[2], sc = box[3];
    if (tid < 4) ws.box[H * 4 + tid] = box[tid];

    // ---- normalise own vertices into [-1,1]^3
    for (int i = tid; i < NV3; i += SDF_THREADS) {
        const int k = i % 3;
        vn[i] = (vn[i] - (k == 0 ? cx : (k == 1 ? cy : cz))) / sc;
    }
    __syncthreads();

    // ---- per-iteration triangle table (SoA rows of NFP floats)
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
        T[15 * NFP + f] = ok ? 1.0f / det : __builtin_nanf("");  // NaN => never a candidate
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

    // ---- compact the non-empty columns (order = column index, deterministic)
    unsigned* gneeded = ws.needed + (size_t)H * SDF_NCOL;
    unsigned short* list = ws.col_list + (size_t)H * SDF_NCOL;
    const int lane = tid % WAVE, wave = tid / WAVE;
    int base = 0;
    for (int c0 = 0; c0 < SDF_NCOL; c0 += SDF_THREADS) {
        const int col = c0 + tid;
        const unsigned m = needed[col];
        gneeded[col] = m;
        const unsigned long long bal = __ballot(m != 0);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (m != 0) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)col;
        for (int w = 0; w < SDF_THREADS / WAVE; ++w) base += wave_cnt[w];
        __syncthreads();
    }
    if (tid == 0) ws.col_count[H] = base;
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

// ------------------------------------------------------------------------------------- eval
// grid = (chunks, 2B), block = 256 (4 waves).  One wave per needed column at a time.
__global__ __launch_bounds__(SDF_THREADS) void sdf_eval_kernel(SdfWorkspace ws, int collect_stats) {
    const int H = blockIdx.y, tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    const int waves_per_hand = gridDim.x * (SDF_THREADS / WAVE);
    const int ncol = ws.col_count[H];
    const float* T = ws.tri + (size_t)H * SDF_TRI_ROWS * NFP;
    const unsigned* needed = ws.needed + (size_t)H * SDF_NCOL;
    const unsigned short* list = ws.col_list + (size_t)H * SDF_NCOL;
    float* phi = ws.phi + (size_t)H * SDF_NVOX;
    unsigned long long st_cand = 0, st_inside = 0, st_dist = 0, st_cols = 0;

    for (int ci = blockIdx.x * (SDF_THREADS / WAVE) + wave; ci < ncol; ci += waves_per_hand) {
        const int col = list[ci], k = col / SDF_G, j = col % SDF_G;
        const unsigned need = needed[col];
        const float py = (float)(2 * j + 1) / (float)SDF_G - 1.0f;
        const float pz = (float)(2 * k + 1) / (float)SDF_G - 1.0f;
        st_cols += 1;

        // ---- step A+B: (u,v) test per triangle (column-wide), t > 0 per needed voxel for candidates
        unsigned par = 0;  // lane-local parity bits over i
        for (int f = lane; f < NFP; f += WAVE) {
            const float ay = T[9 * NFP + f], az = T[10 * NFP + f];
            const float e1y = T[11 * NFP + f], e1z = T[12 * NFP + f], e2y = T[13 * NFP + f], e2z = T[14 * NFP + f];
            const float inv = T[15 * NFP + f];
            const float sy = py - ay, sz = pz - az;
            const float u = __builtin_fmaf(sz, e2y, -(sy * e2z)) * inv;
            const float qx = __builtin_fmaf(sy, e1z, -(sz * e1y));
            const float v = qx * inv;
            const bool cand = (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && (u + v <= 1.0f);
            if (cand) {
                const float ax = T[0 * NFP + f];
                const float e1x = T[3 * NFP + f] - ax, e2x = T[6 * NFP + f] - ax;
                unsigned rem = need;
                while (rem) {
                    const int i = __ffs((int)rem) - 1;
                    rem &= rem - 1;
                    const float px = (float)(2 * i + 1) / (float)SDF_G - 1.0f;
                    const float sx = px - ax;
                    const float qy = __builtin_fmaf(sz, e1x, -(sx * e1z));
                    const float qz = __builtin_fmaf(sx, e1y, -(sy * e1x));
                    const float t = DOT3(e2x, e2y, e2z, qx, qy, qz) * inv;
                    if (t > 0.0f) par ^= 1u << i;
                }
                st_cand += 1;
            }
        }
        const unsigned inside = wave_reduce_xor(par) & need;

        // ---- step C: exact min distance for inside voxels (wave-level min reduction)
        float my_phi = 0.0f;  // lane i (< 32) keeps phi of voxel i
        unsigned rem = inside;
        while (rem) {
            const int i = __ffs((int)rem) - 1;
            rem &= rem - 1;
            const float px = (float)(2 * i + 1) / (float)SDF_G - 1.0f;
            // pass 1: wave-wide upper bound of the distance = min over triangles of |p - centroid|
            float ub = INFINITY;
            for (int f = lane; f < NFP; f += WAVE) {
                const float r = T[19 * NFP + f];
                if (r >= 0.0f) {
                    const float dx = px - T[16 * NFP + f], dy = py - T[17 * NFP + f], dz = pz - T[18 * NFP + f];
                    const float dc = sqrtf(dx * dx + dy * dy + dz * dz);
                    ub = fminf(ub, dc);  // the centroid is a point of the triangle
                }
            }
            ub = wave_reduce_min(ub);
            const float ub_lim = ub * 1.0001f + 1e-6f;
            // pass 2: exact distance only where the sphere lower bound can beat the upper bound
            float best = INFINITY;
            for (int f = lane; f < NFP; f += WAVE) {
                const float r = T[19 * NFP + f];
                if (r < 0.0f) continue;
                const float dx = px - T[16 * NFP + f], dy = py - T[17 * NFP + f], dz = pz - T[18 * NFP + f];
                const float dc = sqrtf(dx * dx + dy * dy + dz * dz);
                if (dc - r > ub_lim) continue;
                const float a[3] = {T[0 * NFP + f], T[1 * NFP + f], T[2 * NFP + f]};
                const float b[3] = {T[3 * NFP + f], T[4 * NFP + f], T[5 * NFP + f]};
                const float c[3] = {T[6 * NFP + f], T[7 * NFP + f], T[8 * NFP + f]};
                best = fminf(best, sdf_point_tri_dist2(a, b, c, px, py, pz));
                st_dist += 1;
            }
            best = wave_reduce_min(best);
            if (lane == i) my_phi = sqrtf(best);
            st_inside += 1;
        }
        if (lane < SDF_G && ((need >> lane) & 1u)) phi[(k * SDF_G + j) * SDF_G + lane] = my_phi;
    }
    if (collect_stats) {
        // per-wave totals -> global counters (diagnostics only; not on the timed path)
        unsigned long long c = st_cand, d = st_dist;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_xor(c, o);
            d += __shfl_xor(d, o);
        }
        if (lane == 0) {
            atomicAdd(&ws.stats[0], st_cols);
            atomicAdd(&ws.stats[1], c);
            atomicAdd(&ws.stats[2], st_inside);
            atomicAdd(&ws.stats[3], d);
        }
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
