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
//     (triangle, column) -- lane = triangle, looping over the needed columns of its yz bounding box --
//     and the t > 0 test of the column's voxels is a loop-free hit mask (sdf_ray_hits);
//   * the distance of an inside voxel is a wave-level min-reduction: lanes across triangles,
//     culled by a bounding-sphere lower bound against a wave-wide upper bound (exact: a culled
//     triangle can never be the minimum);
//   * inside the fused refinement loop the hands move a little per iteration: every inside voxel
//     keeps a candidate list (the triangles that can be nearest while the hand stays within a slack
//     of the pose the list was built at) and its last nearest triangle; while the lists are valid a
//     voxel is answered from its <= 192 candidates instead of all 1538 triangles (list search), the
//     full search only sees new voxels and hands that moved beyond the slack.  Same minimum, bit
//     for bit (tests/test_gpu_parity.py::test_candidate_lists_do_not_change_a_bit).
#pragma once
#include "ihmr_common.h"

#define SDF_THREADS 256
// sdf_prep_kernel<DENSE, PT>: PT threads per workgroup (one hand).  512: a thread owns two vertices, four triangles, two adjacent
// grid columns -- the form for large launches (two 1024-thread workgroups fill a CU's thread slots while waiting most of their
// cycles; 512-thread ones leave room for other kernels' workgroups: +4 % images/s with sixteen batches in flight).  1024: half the
// chain per thread -- the form for small launches, where the kernel is as long as one workgroup (one batch of 64: 29.0 -> 22.1 us).
#define SDF_PREP_THREADS_LARGE 512
#define SDF_PREP_THREADS_SMALL 1024
#ifndef SDF_PREP_SMALL_MAX_HANDS
#define SDF_PREP_SMALL_MAX_HANDS 128     // up to this many hands per launch (one batch of 64) the 1024-thread form is used; at 256 hands (IHMR-MLP, batch 128) the 512-thread form is 11 % faster end to end
#endif
#define SDF_NCOL (SDF_G * SDF_G)   // 1024 columns (k,j)
#define SDF_RAYQ 3072               // (triangle, needed column) pairs per window of the prep kernel's ray-parity queue (a hand has ~440, the largest
                                    // seen 2 900; more take a second window).  Round 6: 4096 -> 3072, the 4 KB hold the queries' cell words
#define SDF_NXCD 8                 // MI355X: 8 XCDs, workgroup b runs on XCD b % 8 (speed only, never correctness)
#ifndef SDF_DIST_WG_PER_CU
#define SDF_DIST_WG_PER_CU 4          // sdf_dist_kernel: 40 KB LDS, <= 128 VGPRs; its grid is persistent: this many workgroups per CU
#endif
#ifndef SDF_ITEM_RUN
#define SDF_ITEM_RUN 2
#endif
// round-4 measures, each exact (the same bits) and switchable at compile time for A/B builds (scripts/ab.sh):
#ifndef SDF_REFUSED_BITS
#define SDF_REFUSED_BITS 1         // voxels refused a candidate list are remembered: searched in full without the list-building half
#endif
#define SDF_ITEM 16                // inside voxels per work item of the distance kernel (one hand per item); power of two

// Per-hand, per-iteration tables (written by the prep kernel, read by the distance kernel):
//   sph[f] = (mx, my, mz, R): the triangle's minimum enclosing circle -- centre m ON the triangle (circumcentre of an acute
//            triangle, midpoint of the longest edge otherwise; the centroid for a near-degenerate one), conservative radius -- used
//            as a bounding sphere and as the in-plane circle of the plane + circle bound; padding triangles are parked at 1e18, so
//            arithmetic alone culls them.  Goes to LDS as it is (direct global -> LDS loads)
//   nrm[f] = the unit normal in 3 x 10 signed bits (n ~ q / 511) | SDF_NRM_NOPLANE for a near-degenerate triangle (sphere bound only)
//   vn4[v] = (x, y, z, 0): the hand's normalised vertices; the exact distance gathers a triangle's corners from here through the
//            packed face table fpk (12 KB per hand instead of a 77 KB per-triangle corner table)
#define SDF_NV4 784                 // vertices per hand in vn4 (778 padded to a multiple of 16)
#define SDF_NRM_NOPLANE 0x40000000u
#define SDF_NRM_EN 1.05e-3f         // max component error of the 10-bit normal (0.5 / 511) + the fp32 error of the normal itself
#define SDF_NRM_EM 2e-6f            // centre off-plane by rounding + evaluation error of the plane distance
struct SdfWorkspace {          // carved from the caller's workspace; H = 2B hands, hand id = hnd*B + b
    float* box;                // [H][4]  centre xyz, scale
    float4* sph;               // [H][NFP]
    unsigned* nrm;             // [H][NFP]
    float4* vn4;               // [H][SDF_NV4]
    const unsigned* fpk[2];    // [NFP] packed faces a | b << 10 | c << 20 of the right / left hand (constants of the model)
    int B;                     // hands [0, B) are right hands, [B, 2B) left hands
    float* phi;                // [H][32768]  (only the INSIDE voxels a sample reads are defined; the dense-grid diagnostic defines all)
    unsigned* inside_bits;     // [H][1024]   bit i of word (k,j): voxel (k,j,i) is read by a sample AND inside the mesh, i.e. phi holds its distance;
                               //             every other voxel is 0 by definition and is never written or read (round 4: was 128-byte rows of zeros)
    unsigned* qcell;           // [B][2][778] per sampling entry (hand, query vertex of the OTHER hand): the grid cell the query falls into, as the
                               //            prep kernel computed it while forming the needed-voxel mask -- SDF_QCELL_IN | (i0 + 1) | (j0 + 1) << 6 |
                               //            (k0 + 1) << 12 | (inside mask of the cell's eight corners) << 18 (round 6), or 0 for a query outside
                               //            the grid.  The fused sampler starts from these words instead of redoing the normalisation of all
                               //            1556 queries (most of which touch no inside voxel) and of reading the bitmap
    unsigned* inside_list;     // [xcd_cap] inside voxels of the whole batch: (hand << 16) | voxel id, 16-aligned run per hand
    int* inside_count;         // [SDF_NCTR] [0] entries in inside_list, [1] in inside_list_a; [SDF_CURSOR] the distance kernel's work cursor, on a
                               //            128-byte line of its own (the counters are read while the cursor is hammered)
    unsigned long long* stats; // [16] optional work counters (ihmr_opt_sdf_counters)
    int xcd_cap;
    // candidate lists of the fused refinement loop (list_mode 0 = off: single-shot callers)
    float* vn_ref;             // [H][2334]  normalised vertices of the hand when its lists were built
    int* hmode;                // [H]        this iteration: 0 = lists are being (re)built, 1 = lists are valid (written by the prep kernel)
    int* run_start;            // [H]        first entry of the hand's run in its list (this iteration)
    float* hdisp;              // [H]        how far the hand has moved from the reference pose of its lists (this iteration; 0 while rebuilding)
    int* lnext;                // [H]        next free candidate-list slot of the hand (voxels that appear after a rebuild take one)
    unsigned* lbits;           // [H][1024]  bit i of word (k,j): voxel (k,j,i) has a candidate list (cleared when the hand starts over)
    unsigned* rbits;           // [H][1024]  ... was REFUSED a list (too long) since the hand started over: searched in full every iteration, without
                               //            the list-building half of that search (its entry in inside_list carries SDF_ENT_REFUSED)
    uint2* lmap;               // [H][32768] voxel -> .x = its list | (the triangle that was nearest the last time it was evaluated) << 16,
                               //            .y = that triangle's packed corner ids (fpk[triangle]: the list search starts with the exact
                               //            distance to it -- with the ids here its corners are requested one round trip earlier)
                               //            (defined where lbits is set)
    unsigned short* lists;     // [H][SDF_LCAP_V][SDF_LCAP_L] triangle ids
    unsigned* inside_list_a;   // [xcd_cap] inside voxels of the hands whose lists are valid (aligned run per hand; inside_count[1])
    // STATIC hands (fused loop; static_mask bit hnd: the hands of that side have had bit-identical vertices since the first iteration of
    // the stage -- the right hands of a stage that moves only the translation): the hand's grid is a fixed function, so what an
    // earlier iteration of the stage found out about a voxel stays true, bit for bit.  known: the voxel's inside / outside status
    // has been determined, inside_k: it is inside (phi holds its distance).  Only voxels that are needed for the first time go
    // through the ray test (and, if inside, through the full search); box, normalised vertices and triangle records are those of
    // the first iteration.
    unsigned* known;           // [H][1024]
    unsigned* inside_k;        // [H][1024]
    int static_mask;           // ... this launch treats these sides as static (the stage's iterations after the first)
    int static_stage;          // ... the stage will (its first iteration records `known` / `inside_k` for them)
    int moving_box;            // bit hnd (subset of static_mask): the hands of that side only TRANSLATE during the stage (round 5): in the hand's own
                               // normalised frame nothing moves, so everything above is kept as for a static hand -- except the box, which is
                               // taken from the current vertices every iteration (queries are normalised with it; the sampler reads it).  NOT an
                               // exact acceleration: the kept geometry is the first iteration's, a recomputation differs from it by the rounding
                               // of the translated vertices (~1e-7 m); ihmr_opt_io::sdf_no_static_reuse = 2 switches it off alone
    int list_mode, force_rebuild;
    // conventions of the upstream module that nothing in the reference pins (ihmr_sdf_options; defaults = DESIGN.md section 4)
    int align_corners;         // grid_sample(align_corners): 0 = False (the default of the reference's pinned torch 1.6.0)
    float loss_div;            // loss[b] = sum of the 1556 sampled values / loss_div (4 = num_hands^2 of the parent project)
    int swap_xz;               // 1: a query's x addresses the field's z axis and vice versa (an upstream grid stored phi[x][y][z] and handed
                               //    to grid_sample as it is); 0 (default): phi[z][y][x], the layout grid_sample's (x, y, z) addresses
};

#define SDF_QCELL_IN 0x80000000u
#define SDF_QCELL_MASK_SHIFT 18      // bits 18-25 of a cell word: which of the cell's eight corners are inside voxels (phi holds a distance)
#define SDF_ENT_REFUSED 0x80000000u  // inside_list entry: (hand << 16) | voxel, hand < 32768; 0xffffffff = padding
#define SDF_MAX_HANDS 32768          // ... so a launch takes at most 16384 samples (every entry point checks: ihmr_hip.hip)
#define SDF_NCTR 64
#define SDF_CURSOR 32
#define SDF_NZERO 3                  // counters that have to be zero before the prep kernel: sdf_zero_counter(c, i), i < SDF_NZERO
__device__ __forceinline__ void sdf_zero_counter(int* c, int i) { c[i < 2 ? i : SDF_CURSOR] = 0; }
__host__ __device__ inline size_t sdf_xcd_cap(int H) { return (size_t)H * (SDF_NVOX + 64); }   // one batch-wide list

#define SDF_LCAP_V 1024              // candidate lists per hand (one per inside voxel, in the order of the hand's run at build time)
#define SDF_LCAP_L 192               // triangles per list (three 64-lane chunks; a voxel whose list would be longer gets none)
#define SDF_LIST_K 8                 // lanes that share one voxel's list in sdf_list_search; list element i is stored at (i % K) * (L / K) + i / K
#define SDF_LIST_ITEM (4 * (WAVE / SDF_LIST_K))   // voxels per work item of sdf_list_search (4 waves)
// which entry of a list-search item group `grp` of wave `wave` takes: INTERLEAVED over the four waves (round 6).  A hand's run is in column
// order, so a wave that takes eight consecutive entries gets eight neighbouring voxels -- all deep or all shallow -- and a partly filled
// item (a hand has ~1.4 items: half of the items are partial) fills waves 0 and 1 while 2 and 3 idle; the workgroup then waits for its
// slowest wave at the next table staging (stamps: 7.3 k of a wave-item's 20 k cycles).  A voxel's result does not depend on who
// computes it: the same bits.  -DSDF_LIST_CONTIGUOUS: rounds 3-5.
#ifdef SDF_LIST_CONTIGUOUS
#define SDF_LIST_ENTRY(wave, grp) ((wave) * (WAVE / SDF_LIST_K) + (grp))
#else
#define SDF_LIST_ENTRY(wave, grp) ((grp) * (SDF_THREADS / WAVE) + (wave))
#endif
#ifndef SDF_LIST_SLACK
#define SDF_LIST_SLACK 0.04f         // lists stay valid while no vertex of the hand has moved further than this (normalised frame)
#endif
__host__ __device__ inline size_t sdf_list_bytes(int H) {
    return (size_t)H * ((size_t)NV3 * 4 + 16 + (size_t)4 * SDF_NCOL * 4 + (size_t)SDF_NVOX * 8 + (size_t)SDF_LCAP_V * SDF_LCAP_L * 2) +
           sdf_xcd_cap(H) * sizeof(unsigned) + 1024;
}
__host__ __device__ inline size_t sdf_ws_bytes(int H, bool lists = false) {
    size_t n = lists ? sdf_list_bytes(H) : 0;
    n += (size_t)H * 4 * sizeof(float);
    n += (size_t)H * NFP * sizeof(float4);          // sph
    n += (size_t)H * NFP * sizeof(unsigned);        // nrm
    n += (size_t)H * SDF_NV4 * sizeof(float4);      // vn4
    n += (size_t)H * SDF_NVOX * sizeof(float);
    n += (size_t)H * SDF_NCOL * sizeof(unsigned);   // inside_bits
    n += sdf_xcd_cap(H) * sizeof(unsigned);
    n += 128 + SDF_NCTR * 4 + 256;
    n += ((size_t)H * NV * sizeof(unsigned) + 255) & ~(size_t)255;      // qcell
    return (n + 255) & ~(size_t)255;
}

static inline SdfWorkspace sdf_carve(void* ws, int H, bool lists = false) {
    SdfWorkspace w;
    char* p = (char*)ws;
    w.box = (float*)p; p += (size_t)H * 4 * sizeof(float);
    w.sph = (float4*)p; p += (size_t)H * NFP * sizeof(float4);
    w.vn4 = (float4*)p; p += (size_t)H * SDF_NV4 * sizeof(float4);
    w.nrm = (unsigned*)p; p += (size_t)H * NFP * sizeof(unsigned);
    w.fpk[0] = w.fpk[1] = nullptr;
    w.B = H / 2;
    w.phi = (float*)p; p += (size_t)H * SDF_NVOX * sizeof(float);
    w.inside_bits = (unsigned*)p; p += (size_t)H * SDF_NCOL * sizeof(unsigned);
    w.stats = (unsigned long long*)p; p += 128;
    w.inside_count = (int*)p; p += SDF_NCTR * 4;
    w.xcd_cap = (int)sdf_xcd_cap(H);
    w.inside_list = (unsigned*)p; p += sdf_xcd_cap(H) * sizeof(unsigned);
    w.qcell = (unsigned*)p; p += ((size_t)H * NV * sizeof(unsigned) + 255) & ~(size_t)255;
    w.vn_ref = nullptr; w.hmode = nullptr; w.run_start = nullptr; w.hdisp = nullptr; w.lnext = nullptr; w.lbits = nullptr; w.rbits = nullptr; w.lmap = nullptr; w.lists = nullptr; w.known = nullptr; w.inside_k = nullptr;
    w.static_mask = 0; w.static_stage = 0; w.moving_box = 0;
    w.inside_list_a = nullptr;
    w.list_mode = 0; w.force_rebuild = 1;
    if (lists) {
        p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
        w.vn_ref = (float*)p; p += (size_t)H * NV3 * 4;
        w.hmode = (int*)p; p += (size_t)H * 4;
        w.run_start = (int*)p; p += (size_t)H * 4;
        w.hdisp = (float*)p; p += (size_t)H * 4;
        w.lnext = (int*)p; p += (size_t)H * 4;
        p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
        w.lbits = (unsigned*)p; p += (size_t)H * SDF_NCOL * 4;
        w.rbits = (unsigned*)p; p += (size_t)H * SDF_NCOL * 4;
        w.known = (unsigned*)p; p += (size_t)H * SDF_NCOL * 4;
        w.inside_k = (unsigned*)p; p += (size_t)H * SDF_NCOL * 4;
        w.inside_list_a = (unsigned*)p; p += sdf_xcd_cap(H) * sizeof(unsigned);
        w.lmap = (uint2*)p; p += (size_t)H * SDF_NVOX * 8;
        p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
        w.lists = (unsigned short*)p;
    }
    w.align_corners = 0;
    w.loss_div = 4.0f;
    w.swap_xz = 0;
    return w;
}

// vertex addressing: hand (b, hnd) of a batch stored with arbitrary strides (floats)
struct VertLayout {
    const float* base;
    long stride_b, stride_h;
    __device__ __forceinline__ const float* hand(int b, int hnd) const { return base + b * stride_b + hnd * stride_h; }
};

// Workgroup barrier for phases that hand over LDS data only: waits for this wave's LDS operations, NOT for its global
// stores (a __syncthreads() carries a workgroup-scope release fence, i.e. s_waitcnt vmcnt(0): every barrier of the prep
// kernel would wait a full L2 round trip for the triangle records / phi stores issued before it)
#define SDF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A hand's table, global -> LDS without passing through registers (global_load_lds: the LDS address is wave-uniform base + lane *
// size, the layout stays linear).  Asynchronous: the caller goes on issuing its other loads and closes with SDF_STAGE_CLOSE():
// every wave waits for its OWN outstanding vector-memory operations (s_waitcnt vmcnt(0): the DMA writes to LDS are tracked by the
// issuing wave's vmcnt only -- a workgroup barrier alone does not wait for them), then the barrier makes all waves' pieces visible
// to all (the form of CK's block_sync_lds_direct_load).  n16 16-byte units by 256 threads.
#define SDF_STAGE_CLOSE() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
// ... the same, but the wave's YOUNGEST vector-memory load may stay in flight (loads return in order: when at most one operation is
// outstanding, every load older than the youngest -- the DMA pieces, issued first -- has returned; outstanding stores of an earlier
// phase only make the wait longer).  The caller issues exactly such a load last (a prefetch nobody waits for here).
#define SDF_STAGE_CLOSE_KEEP1() do { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); __syncthreads(); } while (0)
__device__ __forceinline__ void sdf_stage_async(const void* __restrict__ src, char* dst_lds, int n16) {
    const int tid = threadIdx.x, wave = tid / WAVE;
    for (int base = wave * WAVE; base < n16; base += SDF_THREADS) {
        if (base + (tid % WAVE) < n16)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(src) + (size_t)(base + tid % WAVE) * 16,
                                             (__attribute__((address_space(3))) void*)(dst_lds + (size_t)base * 16), 16, 0, 0);
    }
}

// exclusive prefix sum of data[0..1023] (LDS) by PT threads, thread t owning the 1024 / PT adjacent elements from (1024 / PT) * t;
// returns the total.
template <int PT>
__device__ __forceinline__ int block_excl_scan_1024(int* data, int* scratch /* >= PT / 64 ints, LDS */) {
    constexpr int CPT = SDF_NCOL / PT;
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    SDF_LDS_BARRIER();
    int m[CPT], mine = 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) { m[c] = data[CPT * tid + c]; mine += m[c]; }
    int wtot;
    const int inc = wave_incl_scan(mine, wtot);
    if (lane == WAVE - 1) scratch[wave] = wtot;
    SDF_LDS_BARRIER();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < PT / WAVE; ++w) {
        const int x = scratch[w];
        if (w < wave) base += x;
        total += x;
    }
    SDF_LDS_BARRIER();
    int run = base + inc - mine;
#pragma unroll
    for (int c = 0; c < CPT; ++c) { data[CPT * tid + c] = run; run += m[c]; }
    SDF_LDS_BARRIER();
    return total;
}

// Phase stamps (experiment builds only: -DSDF_STAMPS=1 list search, =2 full search; scripts/sdf_stamps.py): shader-clock time per
// phase of a work item, summed per wave into the spare counter slots 9..15
#ifdef SDF_STAMPS
#define SDF_TK(...) __VA_ARGS__
#define SDF_STAMP() ((long long)__builtin_readcyclecounter())
__device__ long long g_sdf_stamps[4096 * 4][8];       // per (workgroup, wave) sums, plain stores: no atomics in a stamped launch
__device__ long long g_sdf_span[4096 * 4][4];         // entry / exit stamp, XCC id, items of the LAST launch (either search)
__device__ long long g_sdf_prep[4096][8];             // sdf_prep_kernel: phase sums per hand (wave 0), [7] = launches
#else
#define SDF_TK(...)
#define SDF_STAMP() 0ll
#endif

#ifdef SDF_QMASK_CHECK
__device__ unsigned g_qmask_bad[8];
#endif
#ifdef SDF_HANDLOG
__device__ uint4* g_handlog = nullptr;
__device__ unsigned g_handlog_n = 0, g_handlog_cap = 0;
#endif
// ------------------------------------------------------------------------------------- prep + parity
// grid = 2B (block id = hand id H = hnd*B + b, so both hands of sample b sit on XCD b % 8 when B % 8 == 0),
// block = PT (512 or 1024, see above).  Everything up to the inside/outside decision of a hand happens here, out of LDS:
//   box -> normalised vertices -> needed-voxel mask (one 32-bit word per (k,j) column)
//   -> lane = triangle: sphere + abc records to HBM for the distance kernel; for every needed column whose
//      centre lies in the triangle's yz bounding box the (u,v) ray test, and for a hit the t > 0 test of the
//      column's needed voxels, XOR-ed into the column's parity word (LDS atomic; XOR is order-free)
//   -> phi = 0 for outside voxels, inside voxels appended to the batch-wide list.
// Triangle-parallel on purpose: no per-column triangle lists, no dependent LDS chains, balanced lanes.
#ifndef SDF_PREP_MIN_WAVES
#define SDF_PREP_MIN_WAVES 8       // 64 vector registers.  The kernel must not SPILL at this budget (scripts/isa.sh; tests/test_host_cpu.py checks the
#endif                             // built code object): round 6 found the 512-thread form with scratch spills faulting behind the tail launch

template <bool DENSE, int PT>
__global__ __launch_bounds__(PT, SDF_PREP_MIN_WAVES) void sdf_prep_kernel(VertLayout vl, int B, const int32_t* __restrict__ faces_r,
                                                                    const int32_t* __restrict__ faces_l, SdfWorkspace ws,
                                                                    int collect_stats) {
    TL_SCOPE(1);
    constexpr int CPT = SDF_NCOL / PT;             // adjacent grid columns owned by a thread
    constexpr int VPT = (NV + PT - 1) / PT;         // vertices owned by a thread
    __shared__ float vn[NV3];
    __shared__ unsigned needed[SDF_NCOL];
    __shared__ unsigned rowany[SDF_G];      // bit j of word k: column (k,j) has a needed voxel
    __shared__ unsigned parity[SDF_NCOL];
    __shared__ int cur[SDF_NCOL];
    __shared__ float red[6][PT / WAVE];
    __shared__ float red_disp[PT / WAVE];   // per wave: how far its vertices have moved from the reference pose of the hand's candidate lists
    __shared__ int scratch[PT / WAVE];
    __shared__ unsigned rayq[SDF_RAYQ];            // (triangle | column << 11) pairs of the ray-parity phase (carrying the packed corner
                                                   // ids instead of the triangle, 8 bytes per pair, is slower: 44.0 -> 47.1 us per 1024 hands)
    __shared__ unsigned cellw[DENSE ? 1 : SDF_NV4];   // the grid cell of every query (the other hand's vertices), kept from the normalise phase to the
                                                      // end of the kernel, where the inside mask of the cell's corners is known (qcell)
    __shared__ int blk_inside, blk_base, blk_base_a;
    const int H = blockIdx.x, hnd = H / B, b = H % B, tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    SDF_TK(long long pk_[8]; pk_[0] = SDF_STAMP();)
    const float* own = vl.hand(b, hnd);
    const float* other = vl.hand(b, 1 - hnd);
    const int32_t* faces = hnd == 0 ? faces_r : faces_l;  // SoA [3][NFP]
    // ---- bounding box (min / max are exact, any order); a thread owns vertices tid, tid + PT, ...
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float oq[VPT][3];
    struct F3 { float x, y, z; };        // a vertex as ONE 12-byte load (rows of 3 floats, 4-byte aligned)
    const bool lists_on = !DENSE && ws.list_mode != 0;
    // a STATIC hand (see SdfWorkspace::static_mask; uniform over the workgroup): its own vertices are not even read
    const bool stat = lists_on && !ws.force_rebuild && ((ws.static_mask >> hnd) & 1);
    // ... and one that only translates (SdfWorkspace::moving_box): static in its own frame, the box follows the current vertices
    const bool tbox = stat && ((ws.moving_box >> hnd) & 1);
#pragma unroll
    for (int rep = 0; rep < VPT; ++rep) {
        oq[rep][0] = oq[rep][1] = oq[rep][2] = 0.f;
        const int v = tid + rep * PT;
        if (v < NV) {
            const F3 q = reinterpret_cast<const F3*>(other)[v];
            oq[rep][0] = q.x; oq[rep][1] = q.y; oq[rep][2] = q.z;
            if (!stat || tbox) {
                const F3 o = reinterpret_cast<const F3*>(own)[v];
                if (!stat) { vn[3 * v] = o.x; vn[3 * v + 1] = o.y; vn[3 * v + 2] = o.z; }
                mn[0] = fminf(mn[0], o.x); mn[1] = fminf(mn[1], o.y); mn[2] = fminf(mn[2], o.z);
                mx[0] = fmaxf(mx[0], o.x); mx[1] = fmaxf(mx[1], o.y); mx[2] = fmaxf(mx[2], o.z);
            }
        }
    }
    // face indices of this lane's (up to four) triangles (packed: one load each): issued early, consumed after the box is known
    constexpr int TRI_IT = (NFP + PT - 1) / PT;
    unsigned fpkw[TRI_IT];                // (kept packed: four registers instead of twelve live to the end of the kernel)
#pragma unroll
    for (int it = 0; it < TRI_IT; ++it) {
        const unsigned pk = ws.fpk[hnd][min(tid + it * PT, NFP - 1)];
        fpkw[it] = pk;
    }
    // state of the temporal candidate lists, requested now and used much later: the hand's reference pose (this thread's vertices)
    // and which voxels of this thread's two columns have a list
    float rf[VPT][3];
    unsigned lb2[CPT], rb2[CPT];
#pragma unroll
    for (int rep = 0; rep < VPT; ++rep) {
        const int v = tid + rep * PT;
        rf[rep][0] = rf[rep][1] = rf[rep][2] = 0.f;
        if (lists_on && !ws.force_rebuild && !stat && v < NV) {
            const F3 r = reinterpret_cast<const F3*>(ws.vn_ref + (size_t)H * NV3)[v];
            rf[rep][0] = r.x; rf[rep][1] = r.y; rf[rep][2] = r.z;
        }
    }
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) lb2[rep] = rb2[rep] = 0u;       // (loaded at the end of the ray-parity phase: round 6, see there)

    float4 sbox = make_float4(0.f, 0.f, 0.f, 1.f);
    if (stat && !tbox) sbox = *reinterpret_cast<const float4*>(ws.box + H * 4);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = wave_reduce_min(mn[k]), c = wave_reduce_max(mx[k]);
        if (lane == 0) { red[k][wave] = a; red[3 + k][wave] = c; }
    }
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) {
        needed[tid + rep * PT] = DENSE ? 0xffffffffu : 0u;
        parity[tid + rep * PT] = 0u;
    }
    SDF_LDS_BARRIER();
    // every thread finishes the reduction itself (min / max are exact and order-free: the same box in every thread; the wave results
    // are read as broadcasts) -- no serial section of one thread, no second barrier
    float cx, cy, cz, sc;
    {
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = red[k][0]; hi[k] = red[3 + k][0];
#pragma unroll
            for (int w = 1; w < PT / WAVE; ++w) { lo[k] = fminf(lo[k], red[k][w]); hi[k] = fmaxf(hi[k], red[3 + k][w]); }
        }
        cx = (lo[0] + hi[0]) * 0.5f;
        cy = (lo[1] + hi[1]) * 0.5f;
        cz = (lo[2] + hi[2]) * 0.5f;
        sc = 0.6f * fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));  // (1 + 0.2) * 0.5 * max extent
    }
    if (stat && !tbox) { cx = sbox.x; cy = sbox.y; cz = sbox.z; sc = sbox.w; }      // (the box of the stage's first iteration: the same vertices)
    SDF_TK(pk_[1] = SDF_STAMP();)
    if ((!stat || tbox) && tid < 4) ws.box[H * 4 + tid] = tid == 0 ? cx : (tid == 1 ? cy : (tid == 2 ? cz : sc));
    // ---- normalise own vertices into [-1,1]^3; which voxels will the other hand's vertices read?
    const SdfDivisor dsc = sdf_divisor(sc);
    // (temporal candidate lists: the displacement of this thread's vertices from the lists' reference pose is taken here, where the
    // normalised coordinates are in registers, and reduced across the workgroup by the barrier this phase ends with anyway)
    const bool disp_on = lists_on && !stat && !ws.force_rebuild;
    float dmax = 0.f;
#pragma unroll
    for (int rep = 0; rep < VPT; ++rep) {
        const int v = tid + rep * PT;
        if (v >= NV) break;
        if (!stat) {
            const float nx = sdf_div(vn[3 * v] - cx, dsc), ny = sdf_div(vn[3 * v + 1] - cy, dsc), nz = sdf_div(vn[3 * v + 2] - cz, dsc);
            vn[3 * v] = nx; vn[3 * v + 1] = ny; vn[3 * v + 2] = nz;
            ws.vn4[(size_t)H * SDF_NV4 + v] = make_float4(nx, ny, nz, 0.f);     // the exact distance gathers triangle corners from here
            if (disp_on) {
                const float dx = nx - rf[rep][0], dy = ny - rf[rep][1], dz = nz - rf[rep][2];
                dmax = fmaxf(dmax, sqrtf(dx * dx + dy * dy + dz * dz));
            }
        }
        if (!DENSE) {
            const float qx0 = sdf_div(oq[rep][0] - cx, dsc), qy = sdf_div(oq[rep][1] - cy, dsc), qz0 = sdf_div(oq[rep][2] - cz, dsc);
            const float qx = ws.swap_xz ? qz0 : qx0, qz = ws.swap_xz ? qx0 : qz0;
            const float ix = sdf_unnorm(qx, ws.align_corners), iy = sdf_unnorm(qy, ws.align_corners), iz = sdf_unnorm(qz, ws.align_corners);
            const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
            // completely outside the grid (or non-finite): contributes nothing
            const bool in_grid = fx >= -1.0f && fx <= (float)(SDF_G - 1) && fy >= -1.0f && fy <= (float)(SDF_G - 1) && fz >= -1.0f &&
                                 fz <= (float)(SDF_G - 1);
            // the query's cell for the fused sampler (entry hnd * 778 + v of sample b): parked in LDS; written to qcell at the end of
            // the kernel together with the inside mask of the cell's eight corners
            cellw[v] = in_grid ? (SDF_QCELL_IN | (unsigned)((int)fx + 1) | ((unsigned)((int)fy + 1) << 6) | ((unsigned)((int)fz + 1) << 12)) : 0u;
            if (in_grid) {
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
    }
    if (disp_on) {
        dmax = wave_reduce_max(dmax);
        if (lane == 0) red_disp[wave] = dmax;
    }
    SDF_LDS_BARRIER();
    // which columns of a row hold a needed voxel at all: a wave covers two rows per pass, one ballot gives both words
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) {
        const unsigned long long any = __ballot(needed[tid + rep * PT] != 0u);
        if (lane == 0) { rowany[2 * (wave + (PT / WAVE) * rep)] = (unsigned)any; rowany[2 * (wave + (PT / WAVE) * rep) + 1] = (unsigned)(any >> 32); }
    }
    SDF_LDS_BARRIER();
    // ---- temporal candidate lists (fused refinement loop): how far has this hand moved, in its own normalised frame, since its
    //      lists were built?  Within the slack they stay exact (sdf_dist_kernel); beyond it, or when the caller says so (first
    //      iteration of a stage: the parameters may have jumped), the hand starts over: reference frame := now, map cleared.
    SDF_TK(pk_[2] = SDF_STAMP();)
    bool lists_reused = false;
    if (stat && tid == 0) ws.hmode[H] = -1;        // (its new voxels are searched in full, without candidate lists)
    if (lists_on && !stat) {
        float* ref = ws.vn_ref + (size_t)H * NV3;
        bool reuse = !ws.force_rebuild;
        float moved = 0.f;
        if (reuse) {
            float m = red_disp[0];
            for (int w = 1; w < PT / WAVE; ++w) m = fmaxf(m, red_disp[w]);
            reuse = m <= SDF_LIST_SLACK - 1e-4f;        // (a NaN compares false: rebuild)
            moved = m * 1.0001f + 1e-6f;
        }
        lists_reused = reuse;
        if (tid == 0) ws.hdisp[H] = reuse ? moved : 0.f;
        if (!reuse) {
#pragma unroll
            for (int rep = 0; rep < VPT; ++rep) {
                const int v = tid + rep * PT;
                if (v < NV) { ref[3 * v] = vn[3 * v]; ref[3 * v + 1] = vn[3 * v + 1]; ref[3 * v + 2] = vn[3 * v + 2]; }
            }
#pragma unroll
            for (int rep = 0; rep < CPT; ++rep) { ws.lbits[(size_t)H * SDF_NCOL + tid + rep * PT] = 0u; ws.rbits[(size_t)H * SDF_NCOL + tid + rep * PT] = 0u; }
        }
        if (tid == 0) ws.hmode[H] = reuse ? 1 : 0;
    }
    // ---- a static hand: which needed voxels are needed for the FIRST time in this stage?  Only those go through the ray test (their
    //      masks replace `needed` there: `nmask`); none (common once the other hand has settled): the test is skipped altogether.
    unsigned* const newm = reinterpret_cast<unsigned*>(cur);      // (cur is free until the publish phase)
    const unsigned* const nmask = stat ? newm : needed;
    bool stat_skip = false;
    unsigned kn2[CPT], ik2[CPT];       // this thread's columns: voxels whose status is known / known to be inside (static hands)
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) kn2[rep] = ik2[rep] = 0u;
    if (stat) {
#pragma unroll
        for (int rep = 0; rep < CPT; ++rep) {
            kn2[rep] = ws.known[(size_t)H * SDF_NCOL + CPT * tid + rep];
            ik2[rep] = ws.inside_k[(size_t)H * SDF_NCOL + CPT * tid + rep];
        }
        unsigned anyn = 0u;
#pragma unroll
        for (int rep = 0; rep < CPT; ++rep) {
            const unsigned nw = needed[CPT * tid + rep] & ~kn2[rep];
            newm[CPT * tid + rep] = nw;
            anyn |= nw;
        }
        const unsigned long long bal = __ballot(anyn != 0u);
        if (lane == 0) scratch[wave] = bal != 0ull ? 1 : 0;
        SDF_LDS_BARRIER();
        int any = 0;
#pragma unroll
        for (int w = 0; w < PT / WAVE; ++w) any |= scratch[w];
        stat_skip = any == 0;
        if (!stat_skip) {
            // the hand's normalised vertices as the first iteration stored them (the bits a recomputation would give)
#pragma unroll
            for (int rep = 0; rep < VPT; ++rep) {
                const int v = tid + rep * PT;
                if (v < NV) {
                    const float4 p4 = ws.vn4[(size_t)H * SDF_NV4 + v];
                    vn[3 * v] = p4.x; vn[3 * v + 1] = p4.y; vn[3 * v + 2] = p4.z;
                }
            }
            // columns that hold a new voxel
#pragma unroll
            for (int rep = 0; rep < CPT; ++rep) {
                const unsigned long long anyc = __ballot(newm[tid + rep * PT] != 0u);
                if (lane == 0) { rowany[2 * (wave + (PT / WAVE) * rep)] = (unsigned)anyc; rowany[2 * (wave + (PT / WAVE) * rep) + 1] = (unsigned)(anyc >> 32); }
            }
        }
        SDF_LDS_BARRIER();          // (scratch is free again; vn / rowany visible)
    }
    // ---- lane = triangle: records for the distance kernel + ray parity of the needed voxels it can hit
    SDF_TK(pk_[3] = SDF_STAMP();)
    float4* sph = ws.sph + (size_t)H * NFP;
    unsigned* nrm = ws.nrm + (size_t)H * NFP;
    unsigned long long st_tests = 0;
    SDF_TK(pk_[4] = SDF_STAMP();)
    // ---- ray parity as DENSE (triangle, needed column) pairs.  A triangle-parallel loop over the needed columns of each triangle's
    //      yz box is bound by its slowest lane (stamps: 5 us on average, 16 us for the slowest hand of a launch -- a palm triangle
    //      covers 30 columns, most cover 0-2, and every iteration is a dependent LDS round trip).  So the lanes only ENUMERATE their
    //      pairs (a store per column) into an LDS queue, and the pairs are then dealt evenly: thread p takes pairs p, p + PT, ...,
    //      re-derives the triangle's constants (same expressions: the same bits) and does the (u,v) test and the hit mask.
    if (!stat_skip) {
        const unsigned* fpk = ws.fpk[hnd];
        int cnt[TRI_IT], kr[TRI_IT];       // per triangle: needed columns in its box, k range packed (+ the j mask below)
        unsigned jmk[TRI_IT];
        int mine = 0;
#pragma unroll
        for (int it = 0; it < TRI_IT; ++it) {
            const int f = tid + it * PT;
            cnt[it] = 0; jmk[it] = 0u; kr[it] = 0;
            if (f >= NF) continue;
            const int fa = (int)(fpkw[it] & 1023u), fb = (int)((fpkw[it] >> 10) & 1023u), fc = (int)(fpkw[it] >> 20);
            const float ay = vn[3 * fa + 1], az = vn[3 * fa + 2], by = vn[3 * fb + 1], bz = vn[3 * fb + 2], cy2 = vn[3 * fc + 1], cz2 = vn[3 * fc + 2];
            const float e1y = by - ay, e1z = bz - az, e2y = cy2 - ay, e2z = cz2 - az;
            const float det = __builtin_fmaf(e1z, e2y, -(e1y * e2z));
            if (!(fabsf(det) >= 1e-12f)) continue;                // degenerate in yz: the +x ray never counts it
            int j0, j1, k0, k1;
            tri_col_range(ay, by, cy2, az, bz, cz2, j0, j1, k0, k1);
            if (j1 < j0 || k1 < k0) continue;
            const unsigned jmask = (j1 - j0 == 31 ? 0xffffffffu : ((1u << (j1 - j0 + 1)) - 1u)) << j0;
            int n = 0;
            for (int k = k0; k <= k1; ++k) n += __popc(rowany[k] & jmask);
            cnt[it] = n; jmk[it] = jmask; kr[it] = k0 | (k1 << 8);
            mine += n;
        }
        // exclusive scan of the per-thread counts over the workgroup (wave scan + wave totals through LDS)
        int wtot;
        const int inc = wave_incl_scan(mine, wtot);
        if (lane == WAVE - 1) scratch[wave] = wtot;
        SDF_LDS_BARRIER();
        int base_t = inc - mine, P = 0;
#pragma unroll
        for (int wv = 0; wv < PT / WAVE; ++wv) {
            const int x = scratch[wv];
            if (wv < wave) base_t += x;
            P += x;
        }
        for (int win = 0; win < P; win += SDF_RAYQ) {             // (one window unless the hand has more than SDF_RAYQ pairs)
            if (win > 0) SDF_LDS_BARRIER();                        // the previous window's readers are done
            int o = base_t - win;
#pragma unroll
            for (int it = 0; it < TRI_IT; ++it) {
                if (cnt[it] == 0) continue;
                if (o + cnt[it] <= 0 || o >= SDF_RAYQ) { o += cnt[it]; continue; }
                const int f = tid + it * PT, k1 = kr[it] >> 8;
                for (int k = kr[it] & 255; k <= k1; ++k) {
                    unsigned cols = rowany[k] & jmk[it];
                    while (cols) {
                        const int j = __ffs((int)cols) - 1;
                        cols &= cols - 1u;
                        if (o >= 0 && o < SDF_RAYQ) rayq[o] = (unsigned)f | ((unsigned)(k * SDF_G + j) << 11);
                        ++o;
                    }
                }
            }
            SDF_LDS_BARRIER();
            const int nwin = min(P - win, SDF_RAYQ);
            for (int q = tid; q < nwin; q += PT) {
                const unsigned e = rayq[q];
                const int col = (int)(e >> 11);
                const unsigned pk = fpk[e & 2047u];
                const int fa = (int)(pk & 1023u), fb = (int)((pk >> 10) & 1023u), fc = (int)(pk >> 20);
                const float a[3] = {vn[3 * fa], vn[3 * fa + 1], vn[3 * fa + 2]};
                const float bb[3] = {vn[3 * fb], vn[3 * fb + 1], vn[3 * fb + 2]};
                const float c[3] = {vn[3 * fc], vn[3 * fc + 1], vn[3 * fc + 2]};
                const unsigned need = nmask[col];
                bool uv_pass;
                const unsigned hits = sdf_ray_column_hits(a, bb, c, col, need, uv_pass);       // (ihmr_pure.h)
                st_tests += 1;
                if (!uv_pass) continue;
                st_tests += __popc(need);
                if (hits) atomicXor(&parity[col], hits);
            }
        }
    }
    // which voxels of this thread's columns have a candidate list / were refused one: needed by the publish phase only, so requested
    // here, behind the ray-parity phase -- rounds 3-5 requested the words at the top of the kernel and carried them in 2 CPT registers
    // through every phase (the 512-thread form then spilled to scratch at its 64-register budget).  A hand that starts over has just
    // cleared its words (other threads' stores): nothing is read, the words are zero.
    if (lists_reused) {
#pragma unroll
        for (int rep = 0; rep < CPT; ++rep) {
            lb2[rep] = ws.lbits[(size_t)H * SDF_NCOL + CPT * tid + rep];
            rb2[rep] = ws.rbits[(size_t)H * SDF_NCOL + CPT * tid + rep];
        }
    }
    SDF_LDS_BARRIER();
    SDF_TK(pk_[5] = SDF_STAMP();)
    // ---- publish: a thread owns CPT adjacent columns; phi = 0 for the outside voxels a sample reads, inside voxels
    //      into the batch-wide lists
    float* phi = ws.phi + (size_t)H * SDF_NVOX;
    unsigned need2[CPT], inside2[CPT], pub2[CPT];     // needed / inside (= phi defined) / to be evaluated by the distance kernel now
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) {
        const int col = CPT * tid + rep;
        need2[rep] = needed[col];
        if (stat) {           // what earlier iterations of the stage found + the voxels tested just now
            const unsigned nw = stat_skip ? 0u : newm[col];
            pub2[rep] = parity[col] & nw;
            kn2[rep] |= nw;
            ik2[rep] |= pub2[rep];
            inside2[rep] = ik2[rep] & need2[rep];
        } else {
            inside2[rep] = parity[col] & need2[rep];
            pub2[rep] = inside2[rep];
            kn2[rep] = need2[rep];
            ik2[rep] = inside2[rep];
        }
        // the dense-grid diagnostic hands the whole grid out: phi = 0 for its outside voxels (the column's 128-byte row is zeroed, the
        // inside voxels are overwritten by the distance kernel).  The samplers never read an outside voxel: inside_bits below.
        if (DENSE && (need2[rep] & ~inside2[rep])) {
            float4* row = reinterpret_cast<float4*>(phi + col * SDF_G);
#pragma unroll
            for (int q = 0; q < SDF_G / 4; ++q) row[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // inside voxels with a candidate list (low half) / without (high half): one scan for both
        cur[col] = __popc(pub2[rep] & lb2[rep]) | (__popc(pub2[rep] & ~lb2[rep]) << 16);
        // (the parity word has been consumed: from here on the LDS array holds the column's INSIDE word, for the query masks below)
        if (!DENSE) parity[col] = inside2[rep];
    }
    // which voxels hold a distance: one word per column, a thread's CPT adjacent words in one store
    if (CPT == 2) *reinterpret_cast<uint2*>(ws.inside_bits + (size_t)H * SDF_NCOL + 2 * tid) = make_uint2(inside2[0], inside2[CPT - 1]);
    else ws.inside_bits[(size_t)H * SDF_NCOL + tid] = inside2[0];
    // a hand that is static in the stage's later iterations starts from what its first iteration knows; a static hand adds what it learnt
    if (lists_on && (stat ? !stat_skip : (((ws.static_stage >> hnd) & 1) != 0))) {
#pragma unroll
        for (int rep = 0; rep < CPT; ++rep) {
            ws.known[(size_t)H * SDF_NCOL + CPT * tid + rep] = kn2[rep];
            ws.inside_k[(size_t)H * SDF_NCOL + CPT * tid + rep] = ik2[rep];
        }
    }
    const unsigned blk_both = (unsigned)block_excl_scan_1024<PT>(cur, scratch);
    // Two batch-wide lists, each with an aligned run per hand (tail padded with an invalid marker) so that a work item belongs to
    // exactly one hand.  inside_list_a: voxels with a valid candidate list, for sdf_list_search; inside_list: the others (every
    // inside voxel of a single-shot call or of a hand whose lists are being rebuilt), for the full search of sdf_dist_kernel.
    const int n_a = (int)(blk_both & 0xffffu), n_b = (int)(blk_both >> 16);
    const int pad_a = (n_a + SDF_LIST_ITEM - 1) & ~(SDF_LIST_ITEM - 1), pad_b = (n_b + SDF_ITEM - 1) & ~(SDF_ITEM - 1);
    if (tid == 0) {              // (the two reservations from different waves: their round trips overlap)
        blk_inside = n_a + n_b;
        blk_base = n_b > 0 ? atomicAdd(&ws.inside_count[0], pad_b) : 0;
        if (lists_on && !stat) {
            ws.run_start[H] = blk_base;
            if (!lists_reused) ws.lnext[H] = n_b;      // a rebuild hands out slots 0 .. n_b - 1 by position in the run
        }
    }
    if (tid == WAVE) blk_base_a = n_a > 0 ? atomicAdd(&ws.inside_count[1], pad_a) : 0;
#ifdef SDF_HANDLOG       // experiment builds only (scripts/experiments/hand_work_log.py): per launch and hand, how many voxels go to which search
    if (tid == 2 * WAVE && g_handlog) {
        const unsigned i = atomicAdd(&g_handlog_n, 1u);
        if (i < g_handlog_cap) g_handlog[i] = make_uint4((unsigned)H, (unsigned)n_a, (unsigned)n_b, (lists_reused ? 1u : 0u) | (stat ? 2u : 0u));
    }
#endif
    SDF_LDS_BARRIER();
    unsigned* const run_b = ws.inside_list + blk_base;
    unsigned* const run_a = ws.inside_list_a + blk_base_a;     // (never touched when n_a == 0: null for single-shot callers)
    if (tid < pad_b - n_b) run_b[n_b + tid] = 0xffffffffu;
    if (tid < pad_a - n_a) run_a[n_a + tid] = 0xffffffffu;
#pragma unroll
    for (int rep = 0; rep < CPT; ++rep) {
        const int col = CPT * tid + rep;
        const unsigned both = (unsigned)cur[col];
        int oa = (int)(both & 0xffffu), ob = (int)(both >> 16);
        unsigned rem = pub2[rep];
        while (rem) {
            const int i = __ffs((int)rem) - 1;
            rem &= rem - 1;
            const unsigned ent = ((unsigned)H << 16) | (unsigned)(col * SDF_G + i);
            if ((lb2[rep] >> i) & 1u) run_a[oa++] = ent;
            else run_b[ob++] = ent | (((rb2[rep] >> i) & 1u) ? SDF_ENT_REFUSED : 0u);
        }
    }
    // ---- the fused sampler's cell words (round 6): the query's cell AND which of its eight corners hold a distance (bits 18-25: corner
    //      2 c4 + di, c4 = (j - j0) + 2 (k - k0)) -- the inside words of the four columns are in LDS here (the scan's barriers have
    //      published them).  The sampler of the tail launch then needs neither the hand's 4 KB bitmap nor a workgroup barrier before
    //      it can request the phi values: one dependent round trip fewer per iteration.  The cell word waited in LDS (this thread's
    //      own slot) instead of a register held across the ray-parity phase.
    if (!DENSE) {
        unsigned* const qc = ws.qcell + ((size_t)b * 2 + hnd) * NV;
#pragma unroll
        for (int rep = 0; rep < VPT; ++rep) {
            const int v = tid + rep * PT;
            if (v < NV) {
                const unsigned c = cellw[v];
                unsigned m = 0u;
                if (c & SDF_QCELL_IN) {
                    const int i0 = (int)(c & 63u) - 1, j0 = (int)((c >> 6) & 63u) - 1, k0 = (int)((c >> 12) & 63u) - 1;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                        if (j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                            const unsigned wbits = parity[k * SDF_G + j];
                            const unsigned b0 = i0 >= 0 ? ((wbits >> (i0 & 31)) & 1u) : 0u, b1 = i0 + 1 < SDF_G ? ((wbits >> ((i0 + 1) & 31)) & 1u) : 0u;
                            m |= (b0 << (2 * c4)) | (b1 << (2 * c4 + 1));
                        }
                    }
                }
                qc[v] = c | (m << SDF_QCELL_MASK_SHIFT);
            }
        }
    }
    // ---- lane = triangle: records for the distance kernel (minimum enclosing circle + normal), LAST: only the distance kernel reads them,
    //      and only for a hand that has handed it an inside voxel -- 22 % of the hands of a refinement have none (round 5: the records are
    //      28 % of the kernel's instructions).  A hand that will be static in the stage's later iterations keeps its first iteration's
    //      records and computes them whatever it found.
    const bool need_records = !stat && (blk_both != 0u || (lists_on && ((ws.static_stage >> hnd) & 1) != 0));
#pragma unroll 1
    for (int it = 0; it < (need_records ? TRI_IT : 0); ++it) {
        const int f = tid + it * PT;
        if (f >= NFP) break;
        const int fa = (int)(fpkw[it] & 1023u), fb = (int)((fpkw[it] >> 10) & 1023u), fc = (int)(fpkw[it] >> 20);
        const float a[3] = {vn[3 * fa], vn[3 * fa + 1], vn[3 * fa + 2]};
        const float bb[3] = {vn[3 * fb], vn[3 * fb + 1], vn[3 * fb + 2]};
        const float c[3] = {vn[3 * fc], vn[3 * fc + 1], vn[3 * fc + 2]};
        const float e1x = bb[0] - a[0], e1y = bb[1] - a[1], e1z = bb[2] - a[2];
        const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
        // ---- record for the distance kernel: minimum enclosing circle (centre m on the triangle, conservative radius) + normal.
        //      Used for conservative culling only -- the minimum itself is evaluated exactly from the corners
        {
            const float e3x = c[0] - bb[0], e3y = c[1] - bb[1], e3z = c[2] - bb[2];
            const float la = e3x * e3x + e3y * e3y + e3z * e3z;        // squared edge opposite a
            const float lb = e2x * e2x + e2y * e2y + e2z * e2z;        // ... opposite b
            const float lc = e1x * e1x + e1y * e1y + e1z * e1z;        // ... opposite c
            const float nx = __builtin_fmaf(e1y, e2z, -(e1z * e2y)), ny = __builtin_fmaf(e1z, e2x, -(e1x * e2z)),
                        nz = __builtin_fmaf(e1x, e2y, -(e1y * e2x));
            const float n2 = nx * nx + ny * ny + nz * nz;
            // well-conditioned: sin^2 of the angle at a >= 1e-4 (the normal's direction is then good to ~1e-5 in fp32)
            const bool well = f < NF && n2 >= 1e-4f * (lb * lc) && n2 > 1e-30f;
            const float wa = la * (lb + lc - la), wb = lb * (la + lc - lb), wc = lc * (la + lb - lc);
            float mx, my, mz;
            if (!well) {                                   // centroid: on the triangle whatever its shape
                mx = (a[0] + bb[0] + c[0]) * (1.0f / 3.0f); my = (a[1] + bb[1] + c[1]) * (1.0f / 3.0f); mz = (a[2] + bb[2] + c[2]) * (1.0f / 3.0f);
            } else if (wa <= 0.f) {                        // angle at a >= 90 degrees: midpoint of the opposite edge
                mx = 0.5f * (bb[0] + c[0]); my = 0.5f * (bb[1] + c[1]); mz = 0.5f * (bb[2] + c[2]);
            } else if (wb <= 0.f) {
                mx = 0.5f * (a[0] + c[0]); my = 0.5f * (a[1] + c[1]); mz = 0.5f * (a[2] + c[2]);
            } else if (wc <= 0.f) {
                mx = 0.5f * (a[0] + bb[0]); my = 0.5f * (a[1] + bb[1]); mz = 0.5f * (a[2] + bb[2]);
            } else {                                       // acute: circumcentre as a CONVEX combination of the corners (weights in (0,1))
                const float inv_w = 1.0f / (wa + wb + wc);
                const float ua = wa * inv_w, ub_ = wb * inv_w, uc = wc * inv_w;
                mx = ua * a[0] + ub_ * bb[0] + uc * c[0]; my = ua * a[1] + ub_ * bb[1] + uc * c[1]; mz = ua * a[2] + ub_ * bb[2] + uc * c[2];
            }
            float r2 = 0.f;
            {
                float dx = a[0] - mx, dy = a[1] - my, dz = a[2] - mz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
                dx = bb[0] - mx; dy = bb[1] - my; dz = bb[2] - mz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
                dx = c[0] - mx; dy = c[1] - my; dz = c[2] - mz; r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
            }
            unsigned nw = SDF_NRM_NOPLANE;
            if (well) {
                const float inv_n = 511.0f / sqrtf(n2);
                const int qx = (int)rintf(nx * inv_n), qy = (int)rintf(ny * inv_n), qz = (int)rintf(nz * inv_n);
                nw = ((unsigned)qx & 1023u) | (((unsigned)qy & 1023u) << 10) | (((unsigned)qz & 1023u) << 20);
            }
            // padding triangles (f >= NF): parked at 1e18 with radius 0, so the distance kernel culls them by arithmetic alone
            const bool real = f < NF;
            sph[f] = make_float4(real ? mx : 1e18f, real ? my : 1e18f, real ? mz : 1e18f, real ? sqrtf(r2) * 1.0001f + 1e-6f : 0.0f);
            nrm[f] = real ? nw : SDF_NRM_NOPLANE;
        }
    }
    SDF_TK(if (tid == 0 && H < 4096) { pk_[6] = SDF_STAMP(); for (int k = 0; k < 6; ++k) g_sdf_prep[H][k] += pk_[k + 1] - pk_[k]; g_sdf_prep[H][6] += (long long)blk_inside; g_sdf_prep[H][7] += 1; })
    if (collect_stats) {
        unsigned long long c = st_tests;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        unsigned long long nv = 0;
#pragma unroll
        for (int rep = 0; rep < CPT; ++rep) nv += (unsigned long long)__popc(need2[rep]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o);
        if (lane == 0) { atomicAdd(&ws.stats[0], c); atomicAdd(&ws.stats[3], nv); }   // ray tests, needed voxels
        if (tid == 0) atomicAdd(&ws.stats[2], (unsigned long long)blk_inside);            // inside voxels
    }
}

// ------------------------------------------------------------------------------------- distance kernel: shared pieces
// Both searches of sdf_dist_kernel work the same way on a wave's current voxels (slots): conservative culling against the hand's
// table in LDS -> dense (slot, triangle) pairs in the wave's queue -> sdf_refine_pairs (plane + circle bound) -> sdf_exact_pairs
// (closest-point distance, 64-bit LDS atomic min on (distance bits, triangle)).  `best[slot]` always holds a distance some
// triangle attains or exceeds (its upper bound), so every cull against it is exact: a culled triangle cannot be the minimum.
typedef float sdf_v2f __attribute__((ext_vector_type(2)));
#define SDF_QCAP 512                                   // pairs a wave queues before it evaluates them
#define SDF_VSLOTS 8                                   // voxels a wave works on at a time (full search 4, list search 8)
#define SDF_NRM_N 1540                                 // normals staged to LDS (NF rounded up to a multiple of 4)
#define SDF_WAVE_LDS (SDF_QCAP * 4 + SDF_VSLOTS * 8 + SDF_VSLOTS * 2)
#define SDF_DIST_LDS (NFP * 16 + SDF_NRM_N * 4 + (SDF_THREADS / WAVE) * SDF_WAVE_LDS)
struct SdfWaveLds {
    unsigned* q;                  // [SDF_QCAP] (slot << 16) | triangle
    unsigned long long* best;     // [SDF_VSLOTS] (squared distance bits << 32) | triangle   (bits of a float >= 0: unsigned order = float order)
    unsigned short* vox;          // [SDF_VSLOTS] voxel id of the slot
};
__device__ __forceinline__ SdfWaveLds sdf_wave_lds(char* smem, int wave) {
    char* p = smem + NFP * 16 + SDF_NRM_N * 4 + wave * SDF_WAVE_LDS;
    return SdfWaveLds{reinterpret_cast<unsigned*>(p), reinterpret_cast<unsigned long long*>(p + SDF_QCAP * 4),
                      reinterpret_cast<unsigned short*>(p + SDF_QCAP * 4 + SDF_VSLOTS * 8)};
}
// LDS hand-off between the lanes of one wave (DS operations of a wave execute in order; the fences only pin the compiler)
#define SDF_WAVE_SYNC()                                              \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       \
    } while (0)

// Plane + circle lower bound for the queued pairs (the sphere cull leaves ~30 triangles per voxel, this ~9; one LDS gather and
// ~35 instructions per pair against a global gather and ~150 for the exact distance).  A triangle lies in its plane inside the
// circle (m, R) of its record, so for every point q of it |p - q|^2 = h^2 + |P - q|^2 >= h^2 + max(rho - R, 0)^2 with h the
// distance of p from the plane, P its foot point and rho = |P - m| = sqrt(|p - m|^2 - h^2).  h is evaluated from the 10-bit
// normal: |h' - h| <= EN * |p - m|_1 + EM (SDF_NRM_EN: rounding of the normal's components; SDF_NRM_EM: m off the plane by
// rounding), so hlo = max(|h'| - e, 0) <= |h| <= |h'| + e = hhi and lb^2 = hlo^2 + max(sqrt(max(d2 - hhi^2, 0)) - R, 0)^2 is a
// lower bound; a near-degenerate triangle has no normal (SDF_NRM_NOPLANE) and keeps the sphere bound.  Pairs with lb above the
// slot's upper bound are dropped, the rest compacted in place.  Returns the new count.
__device__ __forceinline__ int sdf_refine_pairs(const float4* tab_s, const unsigned* nrm_s, const SdfWaveLds& w, int npair, int lane) {
    int nout = 0;
    for (int base = 0; base < npair; base += WAVE) {
        const bool live = base + lane < npair;
        const unsigned pr = live ? w.q[base + lane] : 0u;
        const int g = (int)(pr >> 16), f = (int)(pr & 0xffffu);
        const float4 sp = tab_s[f];
        const unsigned nw = nrm_s[f < SDF_NRM_N ? f : 0];
        const int id = (int)w.vox[g];
        const float ub2 = __uint_as_float((unsigned)(w.best[g] >> 32));
        float px, py, pz;
        sdf_vox_centre(id, px, py, pz);
        const float dx = px - sp.x, dy = py - sp.y, dz = pz - sp.z;
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const bool plane = (nw & SDF_NRM_NOPLANE) == 0u;
        const float nx = (float)((int)(nw << 22) >> 22), ny = (float)((int)(nw << 12) >> 22), nz = (float)((int)(nw << 2) >> 22);
        const float h = plane ? fabsf(__builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx))) * (1.0f / 511.0f) : 0.0f;
        const float e = plane ? __builtin_fmaf(SDF_NRM_EN, fabsf(dx) + fabsf(dy) + fabsf(dz), SDF_NRM_EM) : 0.0f;
        const float hlo = fmaxf(h - e, 0.0f), hhi = h + e;
        const float rho = sqrtf(fmaxf(__builtin_fmaf(-hhi, hhi, d2), 0.0f)) * 0.9999f;
        const float rlo = fmaxf(rho - sp.w, 0.0f);
        const float lb2 = __builtin_fmaf(rlo, rlo, hlo * hlo);
        const float ub = sqrtf(ub2) * 1.0001f + 1e-6f;
        const bool keep = live && !(lb2 > ub * ub * 1.00001f);
        const unsigned long long m = __ballot(keep);
        const int pos = nout + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) w.q[pos] = pr;            // pos <= base + lane: never ahead of an entry that is still to be read
        nout += __popcll(m);
    }
    return nout;
}

// Exact closest-point distances of the queued pairs, 64 per round; a triangle's corners are gathered from the hand's normalised
// vertices through the packed face table.  Software pipeline, two rounds deep: the corners of round r + 1 and the (pair, face) of
// round r + 2 are requested before the distances of round r are computed.
__device__ __forceinline__ void sdf_exact_pairs(const float4* __restrict__ vn4, const unsigned* __restrict__ fpk, const SdfWaveLds& w,
                                                int npair, int lane) {
    if (npair <= 0) return;
    unsigned pr0 = lane < npair ? w.q[lane] : 0u;
    unsigned pk0 = fpk[pr0 & 0xffffu];
    unsigned pr1 = WAVE + lane < npair ? w.q[WAVE + lane] : 0u;
    unsigned pk1 = fpk[pr1 & 0xffffu];
    float4 A = vn4[pk0 & 1023u], Bv = vn4[(pk0 >> 10) & 1023u], Cv = vn4[pk0 >> 20];
    for (int base = 0; base < npair; base += WAVE) {
        const unsigned pr = pr0;
        const float a[3] = {A.x, A.y, A.z}, b[3] = {Bv.x, Bv.y, Bv.z}, c[3] = {Cv.x, Cv.y, Cv.z};
        const bool live = base + lane < npair;
        pr0 = pr1;
        if (base + WAVE < npair) { A = vn4[pk1 & 1023u]; Bv = vn4[(pk1 >> 10) & 1023u]; Cv = vn4[pk1 >> 20]; }
        if (base + 2 * WAVE < npair) {
            const int i2 = base + 2 * WAVE + lane;
            pr1 = i2 < npair ? w.q[i2] : 0u;
            pk1 = fpk[pr1 & 0xffffu];
        }
        if (live) {
            const int g = (int)(pr >> 16);
            float qx, qy, qz;
            sdf_vox_centre((int)w.vox[g], qx, qy, qz);
            atomicMin(&w.best[g], ((unsigned long long)__float_as_uint(sdf_point_tri_dist2(a, b, c, qx, qy, qz)) << 32) | (pr & 0xffffu));
        }
    }
}

// the hand's table -> LDS (asynchronous; close with SDF_STAGE_CLOSE)
__device__ __forceinline__ void sdf_stage_table(const SdfWorkspace& ws, int H, char* smem) {
    sdf_stage_async(ws.sph + (size_t)H * NFP, smem, NFP);
    sdf_stage_async(ws.nrm + (size_t)H * NFP, smem + NFP * 16, SDF_NRM_N / 4);
}

// work counters / phase stamps of one workgroup's pass over its work units (registers; flushed once at the end of the kernel)
#define SDF_CNT(...) do { if (STATS) { __VA_ARGS__; } } while (0)      // (STATS: template parameter of the distance kernel)
struct SdfAcc {
    unsigned dist = 0, full = 0, build = 0, fresh = 0, ref = 0, sph = 0, vox = 0, refused = 0;      // (per workgroup and launch: 32 bits are plenty)
    SDF_TK(long long tk[7] = {0, 0, 0, 0, 0, 0, 0};)      // items, front, (full: sphere passes), (list: walk), refine, exact, -
};

// What a list-search item needs first from memory -- its lane's list entry, the entry's map word and the corner ids of the triangle
// that was nearest last time -- is two dependent round trips before the item's own loads (list pieces, corners) can even be
// requested; a workgroup of the persistent grid lives for ~2.5 items, so those round trips were a fifth of its lifetime (stamps, round 5:
// the front of a list item 11.3 k of its 26 k cycles).  Every item therefore requests them for the NEXT list item of its workgroup
// (`ni`; the first three units of a workgroup are static, later ones are known a unit ahead) beside its own loads: they ride on
// round trips the item pays anyway, and the next item starts with its pieces and corners.  `item` says whose data `pre` holds;
// an item that finds another one loads its own (sdf_list_pre_sync: the first item of a workgroup).
struct SdfPre {
    unsigned ent;      // this lane's entry of inside_list_a (the K lanes of a group hold the same), 0xffffffff = padding
    unsigned lw, pk;   // lmap[voxel] of the entry (zeros for padding)
    int H, item;       // the item's hand (uniform); the item (uniform; -1: nothing valid)
};
// The map words are requested in one place (sdf_list_pre_issue) and turned into `pre` in another (sdf_list_pre_finish: the first
// arithmetic on them, i.e. where the wave waits for them) -- at the END of the requesting item, before its stores: the load has had
// the whole item to land, and the next item starts on registers nothing is pending on (a value still in flight across the loop's
// back-edge makes hipcc wait for EVERYTHING outstanding at its first use).
struct SdfPreLoad { unsigned ent; uint2 lw; int H; };
// (loads only from addresses that are valid whatever `e0` / `ent` hold: a padding lane reads voxel 0 of hand H)
__device__ __forceinline__ SdfPreLoad sdf_list_pre_issue(const SdfWorkspace& ws, unsigned e0, unsigned ent) {
    const int H = __builtin_amdgcn_readfirstlane((int)(e0 >> 16));      // entry 0 of an item is always valid
    return SdfPreLoad{ent, ws.lmap[(size_t)H * SDF_NVOX + (ent != 0xffffffffu ? (ent & 0xffffu) : 0u)], H};
}
__device__ __forceinline__ void sdf_list_pre_finish(const SdfPreLoad& l, int item, SdfPre& pre) {
    const bool has = l.ent != 0xffffffffu;
    pre.ent = l.ent; pre.lw = has ? l.lw.x : 0u; pre.pk = has ? l.lw.y : 0u; pre.H = l.H; pre.item = item;
}
__device__ __forceinline__ void sdf_list_pre_sync(const SdfWorkspace& ws, int item, SdfPre& pre) {
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const unsigned* g = ws.inside_list_a + (size_t)item * SDF_LIST_ITEM;
    const unsigned e0 = g[0], ent = g[SDF_LIST_ENTRY(wave, lane / SDF_LIST_K)];
    sdf_list_pre_finish(sdf_list_pre_issue(ws, e0, ent), item, pre);
}

// ------------------------------------------------------------------------------------- distance: full search
// (one of the two searches of sdf_dist_kernel, below; grid-strided over its slots.)  The inside voxels of
// the whole batch sit in one list (balanced work matters more here than L2 affinity: the per-sample counts vary
// by 3x).  A work item = SDF_ITEM (16) consecutive list entries = inside voxels of ONE hand: the workgroup stages
// that hand's table in LDS (32 KB), then each wave takes 4 voxels, two at a time through the
// sphere passes (lanes across triangles, packed fp32 on the voxel pair): upper bound = nearest circle centre (a point of its
// triangle), sphere cull, scan-compacted survivors of the wave's four voxels as dense pairs -> refine -> exact (above).
// While a hand's candidate lists are being (re)built it also writes them.
// One work item of the full search: `item` = 16 consecutive entries of inside_list (one hand); half = -1: the whole item (four
// voxels per wave, two passes), 0 / 1: one half of it (two workgroups share the item when there are few: two voxels per wave = ONE
// pass -- a small launch is as long as its longest item).  curH = the hand whose table the workgroup's LDS holds.
template <bool STATS>
__device__ __forceinline__ void sdf_full_item(const SdfWorkspace& ws, int item, int half, char* smem, int& curH, SdfAcc& acc, int ni,
                                              SdfPre& pre) {
    const float4* const sph_s = reinterpret_cast<const float4*>(smem);                                  // [NFP] (centre, radius)
    const unsigned* const nrm_s = reinterpret_cast<const unsigned*>(smem + NFP * 16);                   // [SDF_NRM_N]
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    const SdfWaveLds w = sdf_wave_lds(smem, wave);
    const unsigned* glist = ws.inside_list;
    unsigned& st_dist = acc.dist; unsigned& st_full = acc.full; unsigned& st_build = acc.build;
    unsigned& st_new = acc.fresh; unsigned& st_ref = acc.ref;
    SDF_TK(long long* const tk = acc.tk;)
    const int vpw = half < 0 ? SDF_ITEM / 4 : SDF_ITEM / 8;                  // voxels per wave
    {
        SDF_TK(const long long tk0 = SDF_STAMP();)
        const unsigned ent_l = lane < SDF_ITEM ? glist[item * SDF_ITEM + lane] : 0xffffffffu;
        const int H = (int)(((unsigned)__builtin_amdgcn_readlane((int)ent_l, 0) & ~SDF_ENT_REFUSED) >> 16);   // entry 0 of an item is always valid
        // this wave's entries of the item: e0 + 4 q, q < vpw -- interleaved over the waves, so that the voxels of a partly filled item
        // (a hand's few late voxels) spread over all four waves: one sphere pass each instead of two on the first wave
        const int e0 = (half > 0 ? SDF_ITEM / 2 : 0) + wave;
        if (half > 0 && (unsigned)__builtin_amdgcn_readlane((int)ent_l, SDF_ITEM / 2) == 0xffffffffu) return;   // (padding; uniform)
        // 0: the hand's candidate lists are being (re)built -- by this search; 1: they are valid and these voxels have none; -1: no
        // candidate lists (single-shot callers)
        const int mode = ws.list_mode ? ws.hmode[H] : -1;
        const int run_start = ws.list_mode ? ws.run_start[H] : 0;
        // a voxel that appears while the hand's lists are valid gets a list too (next iteration it is answered from it): the hand has
        // moved `disp` from the reference pose already and may move up to the slack from it, i.e. up to slack + disp from where it is
        // now -- the list bound is widened by twice that
        const float widen = mode == 1 ? 2.0f * (SDF_LIST_SLACK + ws.hdisp[H]) : 2.0f * SDF_LIST_SLACK;
        if (H != curH) {             // uniform over the workgroup (same item for all waves)
            SDF_LDS_BARRIER();       // (the previous table's readers are done; LDS traffic only)
            sdf_stage_table(ws, H, smem);
            SDF_STAGE_CLOSE();
            curH = H;
        }
        const float4* vn4 = ws.vn4 + (size_t)H * SDF_NV4;
        const unsigned* fpk = ws.fpk[H >= ws.B ? 1 : 0];
        unsigned ent4[4];             // this wave's four list entries (uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) ent4[q] = q < vpw ? (unsigned)__builtin_amdgcn_readlane((int)ent_l, e0 + 4 * q) : 0xffffffffu;
        if (lane < 4) {
            w.vox[lane] = (unsigned short)((lane == 0 ? ent4[0] : (lane == 1 ? ent4[1] : (lane == 2 ? ent4[2] : ent4[3]))) & 0xffffu);
            w.best[lane] = (0x7f800000ull << 32) | 0xffffull;
        }
        // list slots of this wave's voxels: by position in the hand's run while the hand rebuilds; a voxel that appears later takes a
        // fresh one when -- and only when -- its list is written (most late voxels are refused a list for its length, every
        // iteration anew: handing them slots up front would use the hand's 1024 up within ~200 iterations).  Lane q keeps the slot of
        // the wave's q-th voxel.
        int my_slot = SDF_LCAP_V;
        int npair = 0;
        SDF_TK(const long long tk1 = SDF_STAMP(), tk_r0 = tk[4] + tk[5];)
        auto flush = [&]() {
            SDF_WAVE_SYNC();
            SDF_TK(const long long f0 = SDF_STAMP();)
            SDF_CNT(st_ref += (unsigned)(lane == 0 ? npair : 0));
            npair = sdf_refine_pairs(sph_s, nrm_s, w, npair, lane);
            SDF_WAVE_SYNC();
            SDF_TK(const long long f1 = SDF_STAMP();)
            sdf_exact_pairs(vn4, fpk, w, npair, lane);
            SDF_CNT(st_dist += (unsigned)(lane == 0 ? npair : 0));
            npair = 0;
            SDF_WAVE_SYNC();
            SDF_TK(const long long f2 = SDF_STAMP(); tk[4] += f1 - f0; tk[5] += f2 - f1;)
        };
        int nmine = 0;    // voxels of this wave
        // ---- the full search for a pair of voxels (packed fp32): table read once for both
        auto pair_pass = [&](unsigned ent0, unsigned ent1r, int vs0, int lidx0) {
            const int nvox = ent1r != 0xffffffffu ? 2 : 1;
            // a voxel that was refused a list since the hand started over (too many triangles within the list bound: a voxel deep inside the
            // other hand) would be refused again: the pair skips the list-building half of the search (pmode -1) when all of it is such
            const int pmode = (SDF_REFUSED_BITS && (ent0 & SDF_ENT_REFUSED) && (nvox == 1 || (ent1r & SDF_ENT_REFUSED))) ? -1 : mode;
            const int id0 = (int)(ent0 & 0xffffu), id1 = (int)((nvox == 2 ? ent1r : ent0) & 0xffffu);
            const sdf_v2f PX = {(float)(2 * (id0 & 31) + 1) / (float)SDF_G - 1.0f, (float)(2 * (id1 & 31) + 1) / (float)SDF_G - 1.0f};
            const sdf_v2f PY = {(float)(2 * ((id0 >> 5) & 31) + 1) / (float)SDF_G - 1.0f,
                                (float)(2 * ((id1 >> 5) & 31) + 1) / (float)SDF_G - 1.0f};
            const sdf_v2f PZ = {(float)(2 * (id0 >> 10) + 1) / (float)SDF_G - 1.0f, (float)(2 * (id1 >> 10) + 1) / (float)SDF_G - 1.0f};
            // d2[t] = |p - m|^2 (packed fp32 on the voxel pair); used for conservative culling only, the minimum over the survivors
            // is computed exactly afterwards
            sdf_v2f d2[NFP / WAVE];
            sdf_v2f ub2 = {INFINITY, INFINITY};
#pragma unroll
            for (int t = 0; t < NFP / WAVE; ++t) {
                const float4 sp = sph_s[lane + WAVE * t];
                const sdf_v2f dx = PX - sdf_v2f{sp.x, sp.x}, dy = PY - sdf_v2f{sp.y, sp.y}, dz = PZ - sdf_v2f{sp.z, sp.z};
                d2[t] = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
                ub2 = __builtin_elementwise_min(ub2, d2[t]);
            }
            // the circle centre is a point of its triangle: dist <= |p - m|
            const float ub_a = sqrtf(wave_reduce_min(ub2.x)) * 1.0001f + 1e-6f;
            const float ub_b = sqrtf(wave_reduce_min(ub2.y)) * 1.0001f + 1e-6f;
            const sdf_v2f ub_lim = {ub_a, ub_b};
            // the slots' upper bound for the refine pass (no triangle id yet: any exact distance will be below it)
            if (lane < nvox) w.best[vs0 + lane] = ((unsigned long long)__float_as_uint(lane ? ub_b * ub_b : ub_a * ub_a) << 32) | 0xffffull;
            unsigned keep_a = 0, keep_b = 0, list_a = 0, list_b = 0;
            if (pmode >= 0) {   // lists are being built: the same cull with the bound widened by twice the motion the list has to cover
#pragma unroll
                for (int t = 0; t < NFP / WAVE; ++t) {
                    const float r = sph_s[lane + WAVE * t].w;
                    const sdf_v2f lim = ub_lim + r;
                    const sdf_v2f lim2 = lim * lim * sdf_v2f{1.00001f, 1.00001f};
                    const sdf_v2f limw = lim + sdf_v2f{widen, widen};
                    const sdf_v2f limw2 = limw * limw * sdf_v2f{1.00001f, 1.00001f};
                    keep_a |= !(d2[t].x > lim2.x) ? (1u << t) : 0u;
                    keep_b |= !(d2[t].y > lim2.y) ? (1u << t) : 0u;
                    list_a |= !(d2[t].x > limw2.x) ? (1u << t) : 0u;
                    list_b |= !(d2[t].y > limw2.y) ? (1u << t) : 0u;
                }
            } else {
#pragma unroll
                for (int t = 0; t < NFP / WAVE; ++t) {
                    const float r = sph_s[lane + WAVE * t].w;
                    const sdf_v2f lim = ub_lim + r;
                    const sdf_v2f lim2 = lim * lim * sdf_v2f{1.00001f, 1.00001f};
                    // cull iff |p - m| - radius > upper bound (exact: such a triangle cannot be the minimum)
                    keep_a |= !(d2[t].x > lim2.x) ? (1u << t) : 0u;
                    keep_b |= !(d2[t].y > lim2.y) ? (1u << t) : 0u;
                }
            }
            for (int v = 0; v < nvox; ++v) {
                unsigned keepmask = v ? keep_b : keep_a;
                const int vs = vs0 + v;
                int cnt;
                const int mine = __popc(keepmask);
                int off = wave_incl_scan(mine, cnt) - mine;
                if (cnt <= SDF_QCAP) {
                    if (npair + cnt > SDF_QCAP) flush();
                    off += npair;
                    while (keepmask) {
                        const int t = __ffs((int)keepmask) - 1;
                        keepmask &= keepmask - 1;
                        w.q[off++] = ((unsigned)vs << 16) | (unsigned)(lane + WAVE * t);
                    }
                    npair += cnt;
                } else {   // (degenerate geometry) more survivors than queue slots: every lane walks its own triangles
                    const float px = v ? PX.y : PX.x, py = v ? PY.y : PY.x, pz = v ? PZ.y : PZ.x;
                    while (keepmask) {
                        const int t = __ffs((int)keepmask) - 1;
                        keepmask &= keepmask - 1;
                        const int f = lane + WAVE * t;
                        const unsigned pk = fpk[f];
                        const float4 A = vn4[pk & 1023u], Bv = vn4[(pk >> 10) & 1023u], Cv = vn4[pk >> 20];
                        const float a[3] = {A.x, A.y, A.z}, b[3] = {Bv.x, Bv.y, Bv.z}, c[3] = {Cv.x, Cv.y, Cv.z};
                        atomicMin(&w.best[vs], ((unsigned long long)__float_as_uint(sdf_point_tri_dist2(a, b, c, px, py, pz)) << 32) | (unsigned)f);
                        SDF_CNT(st_dist += 1);
                    }
                }
                if (pmode >= 0) {
                    // the voxel's candidate list: every triangle whose bounding sphere comes within the (widened) bound; its slot =
                    // the voxel's position in the hand's run (known without atomics) or a fresh one.  Too long a list, or too many voxels: none.
                    unsigned lm = v ? list_b : list_a;
                    int lcnt;
                    const int lmine = __popc(lm);
                    int loff = wave_incl_scan(lmine, lcnt) - lmine;
                    int lidx = mode == 0 ? lidx0 + 4 * v : SDF_LCAP_V;
                    if (mode == 1 && lcnt <= SDF_LCAP_L) {
                        int sl = 0;
                        if (lane == 0) sl = atomicAdd(&ws.lnext[H], 1);
                        lidx = __builtin_amdgcn_readfirstlane(sl);
                    }
                    if (lane == vs) my_slot = lidx;
                    SDF_CNT(acc.refused += (unsigned)(lane == 0 && !(lcnt <= SDF_LCAP_L && lidx < SDF_LCAP_V) ? 1 : 0));
                    if (!(lcnt <= SDF_LCAP_L && lidx < SDF_LCAP_V)) {
                        if (lane == 0) {
                            const int vid = v ? id1 : id0;
                            atomicOr(&ws.rbits[(size_t)H * SDF_NCOL + (vid >> 5)], 1u << (vid & 31));
                        }
                    } else {
                        unsigned short* dst = ws.lists + ((size_t)H * SDF_LCAP_V + lidx) * SDF_LCAP_L;
                        constexpr int LQ = SDF_LCAP_L / SDF_LIST_K;     // element i at (i % K) * LQ + i / K: each of the K reader lanes gets a contiguous piece
                        while (lm) {
                            const int t = __ffs((int)lm) - 1;
                            lm &= lm - 1;
                            dst[(loff % SDF_LIST_K) * LQ + loff / SDF_LIST_K] = (unsigned short)(lane + WAVE * t);
                            ++loff;
                        }
                        // the rest of the list is parked padding: a reader needs no length
                        for (int pos = lcnt + lane; pos < SDF_LCAP_L; pos += WAVE) dst[(pos % SDF_LIST_K) * LQ + pos / SDF_LIST_K] = (unsigned short)(NFP - 1);
                        if (lane == 0) {       // (the voxel's lmap word is written with the result, at the end of the item)
                            const int vid = v ? id1 : id0;
                            atomicOr(&ws.lbits[(size_t)H * SDF_NCOL + (vid >> 5)], 1u << (vid & 31));
                        }
                    }
                }
                ++nmine;
            }
        };
        // the wave's (up to) four voxels as two jobs of the search (ONE call site: the routine is large)
        unsigned j_ent0[2], j_ent1[2];
        int nj = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            j_ent0[k] = 2 * k < vpw ? (unsigned)__builtin_amdgcn_readlane((int)ent_l, e0 + 8 * k) : 0xffffffffu;
            j_ent1[k] = 2 * k < vpw ? (unsigned)__builtin_amdgcn_readlane((int)ent_l, e0 + 8 * k + 4) : 0xffffffffu;
            if (j_ent0[k] != 0xffffffffu) nj = k + 1;     // padding sits only at the tail of a hand's run: a wave's valid entries are a prefix of its four
        }
        const int lidx_base = mode == 0 ? item * SDF_ITEM - run_start + e0 : 0;
        SDF_WAVE_SYNC();
#pragma unroll 1
        for (int k = 0; k < nj; ++k)
            pair_pass(k == 0 ? j_ent0[0] : j_ent0[1], k == 0 ? j_ent1[0] : j_ent1[1], 2 * k, lidx_base + 8 * k);
        // the entries of the workgroup's next list-search item (SdfPre; uniform branch): requested now -- the sphere passes' registers
        // are free again --, their map words after the queue has been worked off
        unsigned n_e0 = 0u, n_ent = 0xffffffffu;
        if (ni >= 0) {
            const unsigned* g = ws.inside_list_a + (size_t)ni * SDF_LIST_ITEM;
            n_e0 = g[0]; n_ent = g[SDF_LIST_ENTRY(wave, lane / SDF_LIST_K)];
        }
        flush();
        SdfPreLoad nl{0xffffffffu, make_uint2(0u, 0u), 0};
        if (ni >= 0) nl = sdf_list_pre_issue(ws, n_e0, n_ent);
        SDF_TK(tk[0] += 1; tk[1] += tk1 - tk0; tk[2] += (SDF_STAMP() - tk1) - (tk[4] + tk[5] - tk_r0);)
        SDF_CNT(st_full += (unsigned)(lane == 0 ? nmine : 0));
        SDF_CNT(st_build += (unsigned)(lane == 0 && mode == 0 ? nmine : 0));
        SDF_CNT(st_new += (unsigned)(lane == 0 && mode == 1 ? nmine : 0));
        if (lane < nmine) {
            const unsigned ent = lane == 0 ? ent4[0] : (lane == 1 ? ent4[1] : (lane == 2 ? ent4[2] : ent4[3]));
            const unsigned long long bst = w.best[lane];
            ws.phi[(size_t)H * SDF_NVOX + (ent & 0xffffu)] = sqrtf(__uint_as_float((unsigned)(bst >> 32)));
            // list of the voxel (if it got one: lbits) and the nearest triangle, which starts its next evaluation (sdf_list_search)
            const unsigned tri = (unsigned)(bst & 0xffffu) < (unsigned)NF ? (unsigned)(bst & 0xffffu) : 0u;
            const int lslot = mode == 0 ? lidx_base + 4 * lane : my_slot;
            if (mode >= 0 && lslot < SDF_LCAP_V)
                ws.lmap[(size_t)H * SDF_NVOX + (ent & 0xffffu)] = make_uint2((unsigned)lslot | (tri << 16), fpk[tri]);
        }
        sdf_list_pre_finish(nl, ni, pre);       // (always assigned: nothing of `pre` is live across the sphere passes)
        SDF_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------- distance from candidate lists
// The inside voxels that HAVE a valid candidate list (inside_list_a; item = SDF_LIST_ITEM consecutive entries of one hand).
// grid-strided over the items, block = 256 (4 waves), 4 workgroups per CU (<= 128 VGPRs, 40 KB LDS): the workgroup stages the
// hand's CURRENT table (LDS, once per run of two items), a wave takes WAVE / K voxels, K lanes per voxel, each lane
// walking its contiguous piece of the voxel's list (<= 192 triangle ids written when the lists were built: 16 bytes = 8 ids per
// load, all requested up front).  Upper bound of a voxel = the exact distance to the triangle that was nearest the last time (any
// triangle gives a valid bound, this one is nearly always the answer): no reduction pass, and only the triangles whose spheres
// reach inside that bound are queued -> refine -> exact.  Same minimum as the full search, bit for bit: the list holds every
// triangle that can be nearest while the hand stays within SDF_LIST_SLACK of its reference pose (see pair_pass), and a culled
// triangle lies farther than a distance some triangle attains.
#define SDF_LIST_VPW (WAVE / SDF_LIST_K)          // voxels per wave
#define SDF_LIST_PIECE (SDF_LCAP_L / SDF_LIST_K / 8)   // 16-byte loads per lane
template <bool STATS>
__device__ __forceinline__ void sdf_list_item(const SdfWorkspace& ws, int item, int ni, char* smem, int& curH, SdfAcc& acc, SdfPre& pre) {
    const float4* const tab_s = reinterpret_cast<const float4*>(smem);                                  // [NFP] (centre, radius)
    const unsigned* const nrm_s = reinterpret_cast<const unsigned*>(smem + NFP * 16);                   // [SDF_NRM_N]
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    const SdfWaveLds w = sdf_wave_lds(smem, wave);
    const int grp = lane / SDF_LIST_K, sub = lane % SDF_LIST_K;
    const unsigned* glist = ws.inside_list_a;
    unsigned& st_dist = acc.dist; unsigned& st_sph = acc.sph; unsigned& st_vox = acc.vox;
    unsigned& st_ref = acc.ref;
    SDF_TK(long long* const tk = acc.tk;)
    {
        // this item's entries, map words and bound-triangle corner ids: requested by the workgroup's previous item (SdfPre), or now
        SDF_TK(const long long tk0 = SDF_STAMP();)
        if (pre.item != item) sdf_list_pre_sync(ws, item, pre);          // (uniform; the first item of a workgroup)
        const int H = pre.H;
        const bool stage = H != curH;          // uniform over the workgroup
        // a new hand's table is requested first (global -> LDS, asynchronous) ...
        if (stage) {
            SDF_LDS_BARRIER();          // (the previous table's readers are done; LDS traffic only)
            sdf_stage_table(ws, H, smem);
        }
        asm volatile("" ::: "memory");         // (the DMA requests are the wave's oldest: SDF_STAGE_CLOSE_KEEP1 below)
        // ... then the entries of the workgroup's NEXT list item (none: this item's again -- valid addresses, nobody uses them) ...
        const int nj = ni >= 0 ? ni : item;
        const unsigned n_e0 = glist[(size_t)nj * SDF_LIST_ITEM], n_ent = glist[(size_t)nj * SDF_LIST_ITEM + SDF_LIST_ENTRY(wave, grp)];
        // ... then this lane's voxel (the same for the K lanes of a group) and everything it needs from memory: all of it in flight
        // while the table goes to LDS
        const unsigned ent = pre.ent;
        const bool has = ent != 0xffffffffu;
        const unsigned vox = has ? (ent & 0xffffu) : 0u;
        uint2* const lword = ws.lmap + (size_t)H * SDF_NVOX + vox;
        const unsigned lw = pre.lw;
        const int nr = (int)(lw >> 16);
        const float4* vn4 = ws.vn4 + (size_t)H * SDF_NV4;
        const unsigned* fpk = ws.fpk[H >= ws.B ? 1 : 0];
        // (a padding lane reads list 0 / vertex 0 of the hand and discards them: no load of this phase sits behind a branch, so that
        // hipcc counts the outstanding loads exactly -- vmcnt(n), not vmcnt(0), at their first uses)
        const uint4* piece = reinterpret_cast<const uint4*>(ws.lists + ((size_t)H * SDF_LCAP_V + (lw & 0xffffu)) * SDF_LCAP_L) + sub * SDF_LIST_PIECE;
        uint4 ids4[SDF_LIST_PIECE];
#pragma unroll
        for (int c = 0; c < SDF_LIST_PIECE; ++c) ids4[c] = piece[c];
        const unsigned pk0 = pre.pk;
        const float4 A0 = vn4[pk0 & 1023u], B0 = vn4[(pk0 >> 10) & 1023u], C0 = vn4[pk0 >> 20];
        // the next item's map words: its entries were requested before this item's pieces and corners and land with them; the map
        // words themselves are this phase's youngest load and stay in flight behind the item's work
        __builtin_amdgcn_sched_barrier(0);
        const SdfPreLoad nl = sdf_list_pre_issue(ws, n_e0, n_ent);
        __builtin_amdgcn_sched_barrier(0);
        if (stage) {
            SDF_STAGE_CLOSE_KEEP1();    // the table has landed (waits for this wave's pieces and corners too: they were all in flight)
            curH = H;
        }
        constexpr unsigned PK = (unsigned)(NFP - 1) | ((unsigned)(NFP - 1) << 16);     // parked padding
        if (!has) {
#pragma unroll
            for (int c = 0; c < SDF_LIST_PIECE; ++c) ids4[c] = make_uint4(PK, PK, PK, PK);
        }
        float px, py, pz;
        sdf_vox_centre((int)vox, px, py, pz);
        SDF_TK(const long long tk1 = SDF_STAMP();)
        float ub;
        {
            const float a[3] = {A0.x, A0.y, A0.z}, b[3] = {B0.x, B0.y, B0.z}, c[3] = {C0.x, C0.y, C0.z};
            const float ub2 = sdf_point_tri_dist2(a, b, c, px, py, pz);
            if (sub == 0) {
                w.vox[grp] = (unsigned short)vox;
                w.best[grp] = ((unsigned long long)__float_as_uint(ub2) << 32) | (unsigned)nr;
            }
            ub = sqrtf(ub2) * 1.0001f + 1e-6f;
        }
        int npair = 0;
        SDF_TK(const long long tk2 = SDF_STAMP(), tk_r0 = tk[4] + tk[5];)
        auto flush = [&]() {
            SDF_WAVE_SYNC();
            SDF_TK(const long long f0 = SDF_STAMP();)
            SDF_CNT(st_ref += (unsigned)(lane == 0 ? npair : 0));
            npair = sdf_refine_pairs(tab_s, nrm_s, w, npair, lane);
            SDF_WAVE_SYNC();
            SDF_TK(const long long f1 = SDF_STAMP();)
            sdf_exact_pairs(vn4, fpk, w, npair, lane);
            SDF_CNT(st_dist += (unsigned)(lane == 0 ? npair : 0));
            npair = 0;
            SDF_WAVE_SYNC();
            SDF_TK(const long long f2 = SDF_STAMP(); tk[4] += f1 - f0; tk[5] += f2 - f1;)
        };
#pragma unroll
        for (int c = 0; c < SDF_LIST_PIECE; ++c) {
            const uint4 cur = ids4[c];
            // a piece is compact (parked padding only at its tail): done when no lane has a candidate left
            const unsigned long long more = __ballot((cur.x & 0xffffu) != (unsigned)(NFP - 1));
            if (!more) break;
            SDF_CNT(st_sph += (unsigned)(lane == 0 ? 8 * __popcll(more) : 0));
            const unsigned ids[8] = {cur.x & 0xffffu, cur.x >> 16, cur.y & 0xffffu, cur.y >> 16, cur.z & 0xffffu, cur.z >> 16, cur.w & 0xffffu, cur.w >> 16};
            float4 sp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) sp[k] = tab_s[ids[k]];         // eight gathers in flight
            unsigned km = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dx = px - sp[k].x, dy = py - sp[k].y, dz = pz - sp[k].z;
                const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
                const float lim = ub + sp[k].w;
                // cull iff |p - m| - radius > upper bound (exact: such a triangle cannot be the minimum); parked padding: never kept
                // (both conditions evaluated, no short circuit: as `&&` this compiles to a branch with exec-mask bookkeeping per candidate)
                const unsigned keep = (unsigned)!(d2 > lim * lim * 1.00001f) & (unsigned)((int)ids[k] != nr);
                km |= keep << k;
            }
            // survivors -> the wave's queue: one scan over the per-lane counts
            int cnt;
            const int mine = __popc(km);
            int off = wave_incl_scan(mine, cnt) - mine;
            if (cnt) {
                if (npair + cnt > SDF_QCAP) flush();
                off += npair;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((km >> k) & 1u) w.q[off++] = ((unsigned)grp << 16) | ids[k];
                npair += cnt;
            }
        }
        flush();
        SDF_TK(tk[0] += 1; tk[1] += tk1 - tk0; tk[3] += (SDF_STAMP() - tk2) - (tk[4] + tk[5] - tk_r0);)
        sdf_list_pre_finish(nl, ni, pre);          // (before this item's stores: nothing else is in flight)
        if (has && sub == 0) {
            const unsigned long long bst = w.best[grp];
            ws.phi[(size_t)H * SDF_NVOX + vox] = sqrtf(__uint_as_float((unsigned)(bst >> 32)));
            // the nearest triangle starts the voxel's next evaluation: its id and corner ids go to the map word when they change (rarely)
            const unsigned nt = (unsigned)(bst & 0xffffu);
            if ((int)nt != nr) *lword = make_uint2((lw & 0xffffu) | (nt << 16), fpk[nt]);
        }
        if (STATS) {
            const unsigned long long mh = __ballot(has && sub == 0);
            st_vox += (unsigned)(lane == 0 ? __popcll(mh) : 0);
        }
        SDF_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------- distance: the launch
// A PERSISTENT grid (as many workgroups as the GPU holds at once: 4 per CU, 256 threads, 40 KB LDS, <= 128 VGPRs) that pulls work
// units from one queue -- the full-search items first (the heavy ones: ~8 us against ~4.5 us), then the list-search items in pairs
// (consecutive items mostly belong to one hand, whose table is then staged once).  The first unit of a workgroup is its own index,
// every further one comes from an atomic cursor that is requested BEFORE the current unit is processed and read after it (the
// round trip hides behind the work).  Why: with one or two items per workgroup and a grid of 4096 the launch lasted 45-65 us
// while the sum of all workgroups' lifetimes was 22 us of the chip -- workgroups live ~5 us (max 23), and the dispatcher refilled
// the freed slots too slowly to keep more than ~60 % of the wave slots busy (stamps: scripts/sdf_stamps.py).  Both searches use
// the same table format, so a workgroup that moves from a hand's full-search item to its list items keeps the table.
template <bool STATS>
__global__ __launch_bounds__(SDF_THREADS, 4) void sdf_dist_kernel(SdfWorkspace ws) {
    TL_SCOPE(2);
#ifdef SDF_SPIN      // experiment: pure residency (no instruction issued) added to every workgroup's life -- does throughput follow the occupied wave-time?
    for (int i_ = 0; i_ < SDF_SPIN; ++i_) __builtin_amdgcn_s_sleep(127);
#endif
    __shared__ __attribute__((aligned(16))) char smem[SDF_DIST_LDS];
    __shared__ int s_next[2];
#ifdef SDF_STAMPS
    const long long span_t0 = (long long)wall_clock64();      // constant 100 MHz, the same on every CU (the shader clocks are not aligned)
    const long long tk_in = SDF_STAMP();
#endif
    const int tid = threadIdx.x, nwg = (int)gridDim.x;
#ifndef SDF_CHUNK_LIST
#define SDF_CHUNK_LIST 2
#endif
#ifndef SDF_TAPER
#define SDF_TAPER 3
#endif
    // Work units, heavy first: full-search items (halves of them when there are few: two workgroups share an item), then the
    // list-search items, in pairs when there are many (consecutive items mostly share a hand: one table staging).
    // Units b and nwg + b belong to workgroup b (no atomics: a small launch never touches the cursor, and the workgroups of a large
    // one do not all hit it at t = 0 -- 1024 simultaneous atomics on one address take 12 us); every further unit comes from the cursor,
    // requested at the START of the unit before (the workgroups are out of step by then) and read after it.
    // (Eight queues, one per XCD, with stealing from the fullest: the same kernel time -- 57.8 against 58.5 us per 512 samples -- and a
    // lower rate with two sequences in flight; static per-XCD lists: +12 %.  The tables do not stay in an XCD's L2 across kernels.)
    const int n_full = (ws.inside_count[0] + SDF_ITEM - 1) / SDF_ITEM;
    const int n_list = ws.list_mode ? (ws.inside_count[1] + SDF_LIST_ITEM - 1) / SDF_LIST_ITEM : 0;
    const int split = 2 * n_full + n_list <= nwg ? 2 : 1;
    const int pair = n_list + n_full > nwg ? SDF_CHUNK_LIST : 1;
    // (the last SDF_TAPER-th of the list items go one by one: short units at the end of the queue shorten the tail of the launch:
    // 48.5 -> 47.7 us per 512 samples)
    const int n_paired = pair > 1 ? (n_list - n_list / SDF_TAPER) / pair * pair : 0;
    const int u_full = n_full * split, u_pair = u_full + n_paired / (pair > 1 ? pair : 1), total = u_pair + (n_list - n_paired);
    // (Tried: the late voxels of a hand with valid lists -- ~6 per hand and iteration, one full-search item each today -- searched
    // by the workgroup of the hand's first list item, whose table is staged anyway ("riders"): 13 % less work, but the units that
    // carry riders are 15-20 us chains -- 47.7 -> 57.8 us per 512 samples even when they go first, 16.4 -> 27.5 us per 64.)
    int* const cursor = ws.inside_count + SDF_CURSOR;                     // zero on entry (zeroed with the list counters)
    SdfAcc acc;
    SdfPre pre{0xffffffffu, 0u, 0u, -1, -1};
    int curH = -1;
    // first list-search item of a unit (-1: a full-search unit, or none) and how many it has
    auto list_first = [&](int u) { return (u < u_full || u >= total) ? -1 : (u < u_pair ? (u - u_full) * pair : n_paired + (u - u_pair)); };
    // Units b, nwg + b and 2 nwg + b belong to workgroup b; every further one comes from the cursor, requested a unit AHEAD (at the top of
    // the unit before the one it follows) so that a workgroup always knows its next unit: the last item of a unit requests the first
    // loads of the next unit's first item (SdfPre).
    int unit = (int)blockIdx.x, unext = nwg + (int)blockIdx.x, round = 0;
    int nxt = 0;
    auto request = [&]() { if (round > 0 && tid == 0) nxt = 3 * nwg + atomicAdd(cursor, 1); };          // the unit after `unext`
    auto advance = [&]() {          // uniform over the workgroup
        unit = unext;
        if (round == 0) { unext = 2 * nwg + (int)blockIdx.x; }
        else {
            if (tid == 0) s_next[round & 1] = nxt;
            __syncthreads();
            unext = s_next[round & 1];
        }
        ++round;
    };
    // (a workgroup's units come in ascending order: its full-search units first -- two loops, so that the list search's prefetched
    // state is not live across the full search, which has no register to spare)
    while (unit < u_full) {
        request();
        sdf_full_item<STATS>(ws, unit / split, split == 2 ? unit % 2 : -1, smem, curH, acc, list_first(unext), pre);
        advance();
    }
    while (unit < total) {
        request();
        const int nfirst = list_first(unext);
        const int i0 = list_first(unit), i1 = i0 + (unit < u_pair ? pair : 1);
        for (int i = i0; i < i1; ++i) sdf_list_item<STATS>(ws, i, i + 1 < i1 ? i + 1 : nfirst, smem, curH, acc, pre);
        advance();
    }
    const int lane = tid % WAVE;
    if (STATS) {
        unsigned long long d = acc.dist;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) {
            atomicAdd(&ws.stats[1], d + (unsigned long long)acc.vox);                        // exact point-triangle distances (list search: + one per voxel for the bound)
            atomicAdd(&ws.stats[4], (unsigned long long)acc.full * (unsigned long long)NF + (unsigned long long)acc.sph);   // bounding-sphere tests
            atomicAdd(&ws.stats[5], (unsigned long long)acc.vox);                            // voxels answered from their lists
            atomicAdd(&ws.stats[6], (unsigned long long)acc.fresh);                          // voxels of hands with valid lists that have none
            atomicAdd(&ws.stats[7], (unsigned long long)acc.build);                          // voxels whose lists were (re)built
            atomicAdd(&ws.stats[8], (unsigned long long)acc.ref);                            // plane + circle tests
            atomicAdd(&ws.stats[9], (unsigned long long)acc.refused);                        // voxels searched in full that got NO list (too long / no slot left)
            atomicAdd(&ws.stats[10], (unsigned long long)acc.full);                          // voxels searched in full, whatever the reason
        }
    }
#ifdef SDF_STAMPS
    if (lane == 0 && blockIdx.x < 4096) {
        long long* d_ = g_sdf_stamps[blockIdx.x * 4 + tid / WAVE];
        for (int k = 0; k < 6; ++k) d_[k] += acc.tk[k];
        d_[6] += SDF_STAMP() - tk_in; d_[7] += 1;
        long long* e_ = g_sdf_span[blockIdx.x * 4 + tid / WAVE];
        e_[0] = span_t0; e_[1] = (long long)wall_clock64(); e_[2] = (long long)blockIdx.x; e_[3] = round;
    }
#endif
}

// ------------------------------------------------------------------------------------- sample
// One workgroup of SDF_SAMPLE_THREADS per sample b; threads tid < nworkers share the 1556 entries.  Entry
// e = hnd*778 + v samples phi of hand `hnd` at vertex v of hand 1-hnd.
// Writes per_vert / origin_scale (B,1556), dval (B,1556,3) = d per_vert / d vertex, loss (B)
// (x mask[b] = [hand_type_array sum > 1.5] when hand_type != nullptr, loss_utils.py:186-188).
// If gverts != nullptr: fused-path gradient gverts[(1-hnd), b, v, :] = gs * dval  (layout (2,B,778,3)).
#define SDF_SAMPLE_THREADS 512
#define SDF_SAMPLE_NIT 4             // entries per thread: ceil(1556 / 448) with the fused kernel's 448 sampling threads
__device__ __forceinline__ void sdf_sample_block(const VertLayout& vl, const SdfWorkspace& ws, float robustifier,
                                                 float* __restrict__ loss, float* __restrict__ per_vert,
                                                 float* __restrict__ origin, float* __restrict__ dval,
                                                 float* __restrict__ gverts, int B, float gs,
                                                 const float* __restrict__ hand_type, float* red16, int b, int nworkers,
                                                 float* g_lds_r = nullptr, float* g_lds_l = nullptr) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    // A thread owns the entries tid, tid + nworkers, ... (at most SDF_SAMPLE_NIT).  Three phases over ALL of them, so that the two
    // dependent round trips (vertex + box, then the eight grid corners) are paid once per thread, not once per entry: as a plain
    // loop the stores of one entry keep the loads of the next from being issued early.  Arithmetic and summation order per entry
    // are those of the loop.
    bool on[SDF_SAMPLE_NIT];
    int hn[SDF_SAMPLE_NIT], vx[SDF_SAMPLE_NIT];
    float4 bx[SDF_SAMPLE_NIT];
    float qv[SDF_SAMPLE_NIT][3];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        const int e = tid + it * nworkers;
        on[it] = e < 2 * NV && tid < nworkers;
        const int ee = on[it] ? e : 0;
        hn[it] = ee / NV; vx[it] = ee % NV;
        bx[it] = *reinterpret_cast<const float4*>(ws.box + (hn[it] * B + b) * 4);
        const float* q = vl.hand(b, 1 - hn[it]) + 3 * vx[it];
        qv[it][0] = q[0]; qv[it][1] = q[1]; qv[it][2] = q[2];
    }
    __builtin_amdgcn_sched_barrier(0);
    float pv[SDF_SAMPLE_NIT][8], ixs[SDF_SAMPLE_NIT][3];
    unsigned ibw[SDF_SAMPLE_NIT][4];
    bool inr[SDF_SAMPLE_NIT];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        const float cx = bx[it].x, cy = bx[it].y, cz = bx[it].z, sc = bx[it].w;
        const SdfDivisor dsc = sdf_divisor(sc);
        const float nx0 = sdf_div(qv[it][0] - cx, dsc), nz0 = sdf_div(qv[it][2] - cz, dsc);
        const float ix = sdf_unnorm(ws.swap_xz ? nz0 : nx0, ws.align_corners), iy = sdf_unnorm(sdf_div(qv[it][1] - cy, dsc), ws.align_corners),
                    iz = sdf_unnorm(ws.swap_xz ? nx0 : nz0, ws.align_corners);
        ixs[it][0] = ix; ixs[it][1] = iy; ixs[it][2] = iz;
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
        inr[it] = on[it] && x0 >= -1.0f && x0 <= (float)(SDF_G - 1) && y0 >= -1.0f && y0 <= (float)(SDF_G - 1) && z0 >= -1.0f &&
                  z0 <= (float)(SDF_G - 1);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) pv[it][c8] = 0.f;
        ibw[it][0] = ibw[it][1] = ibw[it][2] = ibw[it][3] = 0u;
        if (inr[it]) {
            // which of the cell's eight corners hold a distance at all (inside the mesh): one bitmap word per (k, j) row -- a 4 KB table
            // per hand; most query vertices lie outside the other hand and read nothing else
            const int j0 = (int)y0, k0 = (int)z0;
            const unsigned* ib = ws.inside_bits + (size_t)(hn[it] * B + b) * SDF_NCOL;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                if (j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) ibw[it][c4] = ib[k * SDF_G + j];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        if (inr[it]) {
            const int i0 = (int)floorf(ixs[it][0]), j0 = (int)floorf(ixs[it][1]), k0 = (int)floorf(ixs[it][2]);
            const float* phi = ws.phi + (size_t)(hn[it] * B + b) * SDF_NVOX;
            // the two x-neighbours of a cell are adjacent in memory: one load for the pair when both are inside the grid
            // (4-byte aligned 8-byte load: the hardware takes dword-aligned global accesses of any width), single loads at
            // the border of the grid; a voxel outside the mesh is 0 without a load
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                const unsigned wbits = ibw[it][c4];
                const bool b0 = i0 >= 0 && ((wbits >> (i0 & 31)) & 1u), b1 = i0 + 1 < SDF_G && ((wbits >> ((i0 + 1) & 31)) & 1u);
                const float* row = phi + (k * SDF_G + j) * SDF_G;
                if (b0 && b1) {
                    typedef float sdf_f2u __attribute__((ext_vector_type(2), aligned(4)));
                    const sdf_f2u two = *reinterpret_cast<const sdf_f2u*>(row + i0);
                    pv[it][2 * c4] = two.x; pv[it][2 * c4 + 1] = two.y;
                } else if (b0) {
                    pv[it][2 * c4] = row[i0];
                } else if (b1) {
                    pv[it][2 * c4 + 1] = row[i0 + 1];
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        if (!on[it]) continue;
        const int e = tid + it * nworkers, hnd = hn[it], v = vx[it];
        const float sc = bx[it].w;
        const float ix = ixs[it][0], iy = ixs[it][1], iz = ixs[it][2];
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
        float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        if (inr[it]) {
            const int i0 = (int)x0, j0 = (int)y0, k0 = (int)z0;
            const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
            const float wx1 = fx, wx0 = (x0 + 1.0f) - ix, wy1 = fy, wy0 = (y0 + 1.0f) - iy, wz1 = fz, wz0 = (z0 + 1.0f) - iz;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const int di = c8 & 1, dj = (c8 >> 1) & 1, dk = c8 >> 2;
                const int i = i0 + di, j = j0 + dj, k = k0 + dk;
                if (i >= 0 && i < SDF_G && j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                    const float p = pv[it][c8];
                    const float wx = di ? wx1 : wx0, wy = dj ? wy1 : wy0, wz = dk ? wz1 : wz0;
                    val += p * (wx * wy * wz);
                    gx += (di ? p : -p) * (wy * wz);
                    gy += (dj ? p : -p) * (wx * wz);
                    gz += (dk ? p : -p) * (wx * wy);
                }
            }
        }
        // chain: ix = ((x+1)*G - 1)/2 (or (x+1)/2*(G-1)), x = (q - c)/s  =>  d ix / d q = G / (2 s)  (or (G-1) / (2 s))
        const float chain = (0.5f * (float)(ws.align_corners ? SDF_G - 1 : SDF_G)) / sc;
        gx *= chain; gy *= chain; gz *= chain;
        if (ws.swap_xz) { const float t = gx; gx = gz; gz = t; }       // back to the query vertex's own axes
        if (robustifier > 0.f) {
            const float r = val / robustifier, fr = r * r;
            const float dfr = 2.0f * r / robustifier;       // d fr / d val
            const float dout = dfr / ((fr + 1.0f) * (fr + 1.0f));
            val = fr / (fr + 1.0f);
            gx *= dout; gy *= dout; gz *= dout;
        }
        if (per_vert) {
            per_vert[(size_t)b * 2 * NV + e] = val;
            origin[(size_t)b * 2 * NV + e] = val * sc;
        }
        if (dval) {
            float* d = dval + ((size_t)b * 2 * NV + e) * 3;
            d[0] = gx; d[1] = gy; d[2] = gz;
        }
        if (gverts) {
            float* g = gverts + (((size_t)(1 - hnd) * B + b) * NV + v) * 3;
            g[0] = gs * gx; g[1] = gs * gy; g[2] = gs * gz;
        }
        if (g_lds_r) {     // fused tail: straight into the LBS backward's LDS record of the hand the vertex belongs to (raw hand frame:
            float* g = (hnd ? g_lds_r : g_lds_l) + 3 * v;       // the left hand's x negated, as lbs_bwd1_hand's staging does)
            const float g0 = gs * gx;
            g[0] = hnd ? g0 : -g0; g[1] = gs * gy; g[2] = gs * gz;
        }
        acc += val;
    }
    // fixed-order block sum: DPP inside each wave, the 16 wave totals through LDS
    const float wsum = wave_reduce_sum_dpp(acc);
    if (tid % WAVE == 0) red16[tid / WAVE] = wsum;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < SDF_SAMPLE_THREADS / WAVE; ++wv) tot += red16[wv];
        float mask = 1.0f;
        if (hand_type) mask = (hand_type[b * 2] + hand_type[b * 2 + 1]) > 1.5f ? 1.f : 0.f;
        loss[b] = tot / ws.loss_div * mask;  // parent project: sum / num_hands^2
    }
}

// The sampler of the fused tail (opt_tail_kernel): the same values as sdf_sample_block, bit for bit, from what the collision kernels
// of the iteration have already worked out.
//   * the prep kernel stored every query's grid cell (SdfWorkspace::qcell) while forming the needed-voxel mask: the 1556 normalisations
//     (a third of the old sampler's instructions) are not redone;
//   * ... and, since round 6, which of the cell's eight corners are inside voxels (bits 18-25 of the cell word: the prep kernel had the
//     bitmap in LDS when it wrote the word) -- rounds 4-5 staged both hands' 4 KB bitmaps in LDS and needed a workgroup barrier before the
//     phi values could be requested;
//   * a query none of whose eight cell corners is an inside voxel -- 94 % of them -- has value 0 and gradient 0: the old code computed
//     exactly +0 for it (0 * w sums), so it loads nothing more and skips the arithmetic; the others load their
//     vertex, box and phi values in ONE round trip (the phi addresses follow from the cell word) and run sdf_sample_block's
//     expressions.  A degenerate hand (box scale outside [1e-6, 1e6]: the oracle's infinities / NaNs) takes the full path for every entry.
// Two dependent global round trips (cell words, then vertex + phi of the few entries that need them) and no barrier before the block sum.
#ifdef TAIL_STAMPS
__device__ long long g_samp_stamps[4096][8];
#define SAMP_TK(k) do { samp_t_[k] = (long long)__builtin_readcyclecounter(); } while (0)
#define SAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define SAMP_TK(k)
#define SAMP_DRAIN()
#endif
__device__ __forceinline__ void sdf_sample_fused(const VertLayout& vl, const SdfWorkspace& ws, float* __restrict__ loss, int B, float gs,
                                                 const float* __restrict__ hand_type, float* red16, int b, int nworkers,
                                                 float* g_lds_r, float* g_lds_l) {
    const int tid = threadIdx.x;
    float acc = 0.f;
#ifdef TAIL_STAMPS
    long long samp_t_[8];
    SAMP_TK(0);
#endif
    // ---- the cell words of this thread's entries tid, tid + nworkers, ... and both boxes
    unsigned cw[SDF_SAMPLE_NIT];
    bool on[SDF_SAMPLE_NIT];
    int hn[SDF_SAMPLE_NIT], vx[SDF_SAMPLE_NIT];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        const int e = tid + it * nworkers;
        on[it] = e < 2 * NV;
        const int ee = on[it] ? e : 0;
        hn[it] = ee / NV; vx[it] = ee % NV;
        cw[it] = ws.qcell[(size_t)b * 2 * NV + ee];
    }
    const float4 box0 = *reinterpret_cast<const float4*>(ws.box + (size_t)b * 4), box1 = *reinterpret_cast<const float4*>(ws.box + ((size_t)B + b) * 4);
    __builtin_amdgcn_sched_barrier(0);
    SAMP_TK(1);
    SAMP_DRAIN();
    SAMP_TK(2);
    // ---- which of a cell's eight corners hold a distance: bits 18-25 of the cell word (the prep kernel had the bitmap in LDS)
    const bool fast0 = box0.w >= 1e-6f && box0.w <= 1e6f, fast1 = box1.w >= 1e-6f && box1.w <= 1e6f;
    unsigned m8[SDF_SAMPLE_NIT];
    bool nz[SDF_SAMPLE_NIT];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        unsigned m = (on[it] && (cw[it] & SDF_QCELL_IN)) ? ((cw[it] >> SDF_QCELL_MASK_SHIFT) & 0xffu) : 0u;
#ifdef SDF_QMASK_CHECK      // experiment builds only: the mask recomputed from the hand's bitmap in global memory; mismatches counted, the bitmap's used
        {
            const unsigned c = cw[it];
            const int i0 = (int)(c & 63u) - 1, j0 = (int)((c >> 6) & 63u) - 1, k0 = (int)((c >> 12) & 63u) - 1;
            unsigned mr = 0u;
            if (on[it] && (c & SDF_QCELL_IN)) {
                const unsigned* ib = ws.inside_bits + (size_t)(hn[it] * B + b) * SDF_NCOL;
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                    if (j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                        const unsigned wbits = ib[k * SDF_G + j];
                        const unsigned b0 = i0 >= 0 ? ((wbits >> (i0 & 31)) & 1u) : 0u, b1 = i0 + 1 < SDF_G ? ((wbits >> ((i0 + 1) & 31)) & 1u) : 0u;
                        mr |= (b0 << (2 * c4)) | (b1 << (2 * c4 + 1));
                    }
                }
            }
            if (mr != m) { atomicAdd(&g_qmask_bad[0], 1u); if (c >> 26 & 31u) atomicAdd(&g_qmask_bad[1], 1u); g_qmask_bad[2] = c; g_qmask_bad[3] = mr; }
            atomicAdd(&g_qmask_bad[4], 1u);
            m = mr;
        }
#endif
        m8[it] = m;
        nz[it] = on[it] && (m != 0u || !(hn[it] ? fast1 : fast0));
    }
    SAMP_TK(3);
    // ---- the entries that touch an inside voxel: vertex + phi values, one round trip
    float qv[SDF_SAMPLE_NIT][3], pv[SDF_SAMPLE_NIT][8];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        qv[it][0] = qv[it][1] = qv[it][2] = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) pv[it][c8] = 0.f;
        if (nz[it]) {
            const float* q = vl.hand(b, 1 - hn[it]) + 3 * vx[it];
            qv[it][0] = q[0]; qv[it][1] = q[1]; qv[it][2] = q[2];
            const unsigned c = cw[it];
            const int i0 = (int)(c & 63u) - 1, j0 = (int)((c >> 6) & 63u) - 1, k0 = (int)((c >> 12) & 63u) - 1;
            const float* phi = ws.phi + (size_t)(hn[it] * B + b) * SDF_NVOX;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                const bool b0 = (m8[it] >> (2 * c4)) & 1u, b1 = (m8[it] >> (2 * c4 + 1)) & 1u;
                const float* row = phi + (k * SDF_G + j) * SDF_G;
                if (b0 && b1) {
                    typedef float sdf_f2u __attribute__((ext_vector_type(2), aligned(4)));
                    const sdf_f2u two = *reinterpret_cast<const sdf_f2u*>(row + i0);
                    pv[it][2 * c4] = two.x; pv[it][2 * c4 + 1] = two.y;
                } else if (b0) {
                    pv[it][2 * c4] = row[i0];
                } else if (b1) {
                    pv[it][2 * c4 + 1] = row[i0 + 1];
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    SAMP_DRAIN();
    SAMP_TK(4);
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        if (!on[it]) continue;
        const int hnd = hn[it], v = vx[it];
        float val = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        if (nz[it]) {           // sdf_sample_block's expressions, operation for operation
            const float4 bx = hnd ? box1 : box0;
            const float cx = bx.x, cy = bx.y, cz = bx.z, sc = bx.w;
            const SdfDivisor dsc = sdf_divisor(sc);
            const float nx0 = sdf_div(qv[it][0] - cx, dsc), nz0 = sdf_div(qv[it][2] - cz, dsc);
            const float ix = sdf_unnorm(ws.swap_xz ? nz0 : nx0, ws.align_corners), iy = sdf_unnorm(sdf_div(qv[it][1] - cy, dsc), ws.align_corners),
                        iz = sdf_unnorm(ws.swap_xz ? nx0 : nz0, ws.align_corners);
            const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
            const bool inr = x0 >= -1.0f && x0 <= (float)(SDF_G - 1) && y0 >= -1.0f && y0 <= (float)(SDF_G - 1) && z0 >= -1.0f && z0 <= (float)(SDF_G - 1);
            if (inr) {
                const int i0 = (int)x0, j0 = (int)y0, k0 = (int)z0;
                const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
                const float wx1 = fx, wx0 = (x0 + 1.0f) - ix, wy1 = fy, wy0 = (y0 + 1.0f) - iy, wz1 = fz, wz0 = (z0 + 1.0f) - iz;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const int di = c8 & 1, dj = (c8 >> 1) & 1, dk = c8 >> 2;
                    const int i = i0 + di, j = j0 + dj, k = k0 + dk;
                    if (i >= 0 && i < SDF_G && j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                        const float p = pv[it][c8];
                        const float wx = di ? wx1 : wx0, wy = dj ? wy1 : wy0, wz = dk ? wz1 : wz0;
                        val += p * (wx * wy * wz);
                        gx += (di ? p : -p) * (wy * wz);
                        gy += (dj ? p : -p) * (wx * wz);
                        gz += (dk ? p : -p) * (wx * wy);
                    }
                }
            }
            const float chain = (0.5f * (float)(ws.align_corners ? SDF_G - 1 : SDF_G)) / sc;
            gx *= chain; gy *= chain; gz *= chain;
            if (ws.swap_xz) { const float t = gx; gx = gz; gz = t; }
        }
        // (an entry that takes no part: val = +0 and gradient = +0, what the arithmetic above gives for eight zero corners and a finite chain)
        float* g = (hnd ? g_lds_r : g_lds_l) + 3 * v;       // raw hand frame: the left hand's x negated, as lbs_bwd1_hand's staging does
        const float g0 = gs * gx;
        g[0] = hnd ? g0 : -g0; g[1] = gs * gy; g[2] = gs * gz;
        acc += val;
    }
    SAMP_TK(5);
    // fixed-order block sum: DPP inside each wave, the wave totals through LDS
    const float wsum = wave_reduce_sum_dpp(acc);
    if (tid % WAVE == 0) red16[tid / WAVE] = wsum;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < SDF_SAMPLE_THREADS / WAVE; ++wv) tot += red16[wv];
        float mask = 1.0f;
        if (hand_type) mask = (hand_type[b * 2] + hand_type[b * 2 + 1]) > 1.5f ? 1.f : 0.f;
        loss[b] = tot / ws.loss_div * mask;  // parent project: sum / num_hands^2
    }
#ifdef TAIL_STAMPS
    SAMP_TK(6);
    if (tid == 0 && b < 4096) { for (int k = 0; k < 6; ++k) g_samp_stamps[b][k] += samp_t_[k + 1] - samp_t_[k]; g_samp_stamps[b][7] += 1; }
#endif
}

// The sampler of IHMR-MLP's evaluations (opt_sample_loss_kernel with a keep / reject decision behind it; round 6): sdf_sample_block's
// values -- loss, per-vertex depth, origin-scale depth -- from the prep kernel's cell words, as sdf_sample_fused takes them: an entry none
// of whose eight cell corners is an inside voxel (94 %) is the exact +0 the full arithmetic gives and costs one word; the others load
// vertex, box and phi in one round trip and run sdf_sample_block's expressions.  No gradient: an evaluation of MLPModel.test() has no
// backward (the training step and every other caller go through sdf_sample_block).  Called by ALL threads of the workgroup; threads
// tid >= nworkers (the loss wave) take part in the block sum only.
__device__ __forceinline__ void sdf_sample_cells(const VertLayout& vl, const SdfWorkspace& ws, float* __restrict__ loss,
                                                 float* __restrict__ per_vert, float* __restrict__ origin, int B,
                                                 const float* __restrict__ hand_type, float* red16, int b, int nworkers) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    unsigned cw[SDF_SAMPLE_NIT];
    bool on[SDF_SAMPLE_NIT];
    int hn[SDF_SAMPLE_NIT], vx[SDF_SAMPLE_NIT];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        const int e = tid + it * nworkers;
        on[it] = e < 2 * NV && tid < nworkers;
        const int ee = on[it] ? e : 0;
        hn[it] = ee / NV; vx[it] = ee % NV;
        cw[it] = ws.qcell[(size_t)b * 2 * NV + ee];
    }
    const float4 box0 = *reinterpret_cast<const float4*>(ws.box + (size_t)b * 4), box1 = *reinterpret_cast<const float4*>(ws.box + ((size_t)B + b) * 4);
    __builtin_amdgcn_sched_barrier(0);
    const bool fast0 = box0.w >= 1e-6f && box0.w <= 1e6f, fast1 = box1.w >= 1e-6f && box1.w <= 1e6f;
    unsigned m8[SDF_SAMPLE_NIT];
    bool nz[SDF_SAMPLE_NIT];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        const unsigned m = (on[it] && (cw[it] & SDF_QCELL_IN)) ? ((cw[it] >> SDF_QCELL_MASK_SHIFT) & 0xffu) : 0u;
        m8[it] = m;
        nz[it] = on[it] && (m != 0u || !(hn[it] ? fast1 : fast0));
    }
    float qv[SDF_SAMPLE_NIT][3], pv[SDF_SAMPLE_NIT][8];
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        qv[it][0] = qv[it][1] = qv[it][2] = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) pv[it][c8] = 0.f;
        if (nz[it]) {
            const float* q = vl.hand(b, 1 - hn[it]) + 3 * vx[it];
            qv[it][0] = q[0]; qv[it][1] = q[1]; qv[it][2] = q[2];
            const unsigned c = cw[it];
            const int i0 = (int)(c & 63u) - 1, j0 = (int)((c >> 6) & 63u) - 1, k0 = (int)((c >> 12) & 63u) - 1;
            const float* phi = ws.phi + (size_t)(hn[it] * B + b) * SDF_NVOX;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int j = j0 + (c4 & 1), k = k0 + (c4 >> 1);
                const bool b0 = (m8[it] >> (2 * c4)) & 1u, b1 = (m8[it] >> (2 * c4 + 1)) & 1u;
                const float* row = phi + (k * SDF_G + j) * SDF_G;
                if (b0 && b1) {
                    typedef float sdf_f2u __attribute__((ext_vector_type(2), aligned(4)));
                    const sdf_f2u two = *reinterpret_cast<const sdf_f2u*>(row + i0);
                    pv[it][2 * c4] = two.x; pv[it][2 * c4 + 1] = two.y;
                } else if (b0) {
                    pv[it][2 * c4] = row[i0];
                } else if (b1) {
                    pv[it][2 * c4 + 1] = row[i0 + 1];
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < SDF_SAMPLE_NIT; ++it) {
        if (!on[it]) continue;
        const int e = tid + it * nworkers, hnd = hn[it];
        const float sc = (hnd ? box1 : box0).w;
        float val = 0.f;
        if (nz[it]) {           // sdf_sample_block's expressions, operation for operation (value only)
            const float4 bx = hnd ? box1 : box0;
            const SdfDivisor dsc = sdf_divisor(sc);
            const float nx0 = sdf_div(qv[it][0] - bx.x, dsc), nz0 = sdf_div(qv[it][2] - bx.z, dsc);
            const float ix = sdf_unnorm(ws.swap_xz ? nz0 : nx0, ws.align_corners), iy = sdf_unnorm(sdf_div(qv[it][1] - bx.y, dsc), ws.align_corners),
                        iz = sdf_unnorm(ws.swap_xz ? nx0 : nz0, ws.align_corners);
            const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
            const bool inr = x0 >= -1.0f && x0 <= (float)(SDF_G - 1) && y0 >= -1.0f && y0 <= (float)(SDF_G - 1) && z0 >= -1.0f && z0 <= (float)(SDF_G - 1);
            if (inr) {
                const int i0 = (int)x0, j0 = (int)y0, k0 = (int)z0;
                const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
                const float wx1 = fx, wx0 = (x0 + 1.0f) - ix, wy1 = fy, wy0 = (y0 + 1.0f) - iy, wz1 = fz, wz0 = (z0 + 1.0f) - iz;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const int di = c8 & 1, dj = (c8 >> 1) & 1, dk = c8 >> 2;
                    const int i = i0 + di, j = j0 + dj, k = k0 + dk;
                    if (i >= 0 && i < SDF_G && j >= 0 && j < SDF_G && k >= 0 && k < SDF_G) {
                        const float wx = di ? wx1 : wx0, wy = dj ? wy1 : wy0, wz = dk ? wz1 : wz0;
                        val += pv[it][c8] * (wx * wy * wz);
                    }
                }
            }
        }
        per_vert[(size_t)b * 2 * NV + e] = val;
        origin[(size_t)b * 2 * NV + e] = val * sc;
        acc += val;
    }
    // fixed-order block sum: DPP inside each wave, the wave totals through LDS (as sdf_sample_block)
    const float wsum = wave_reduce_sum_dpp(acc);
    if (tid % WAVE == 0) red16[tid / WAVE] = wsum;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < SDF_SAMPLE_THREADS / WAVE; ++wv) tot += red16[wv];
        float mask = 1.0f;
        if (hand_type) mask = (hand_type[b * 2] + hand_type[b * 2 + 1]) > 1.5f ? 1.f : 0.f;
        loss[b] = tot / ws.loss_div * mask;
    }
}

// seam B: grid = B, block = SDF_SAMPLE_THREADS (512)
__global__ __launch_bounds__(SDF_SAMPLE_THREADS) void sdf_sample_kernel(VertLayout vl, SdfWorkspace ws, float robustifier,
                                                                 float* __restrict__ loss, float* __restrict__ per_vert,
                                                                 float* __restrict__ origin, float* __restrict__ dval, int B) {
    __shared__ float red16[SDF_SAMPLE_THREADS / WAVE];
    sdf_sample_block(vl, ws, robustifier, loss, per_vert, origin, dval, nullptr, B, 0.f, nullptr, red16, blockIdx.x, SDF_SAMPLE_THREADS);
    if (blockIdx.x == 0 && threadIdx.x < SDF_NZERO) sdf_zero_counter(ws.inside_count, (int)threadIdx.x);   // ready for the next call's prep kernel
}
