// libihmr_hip.so -- C ABI (include/ihmr_hip.h) over the gfx950 kernels.  Single translation unit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC ihmr_hip.hip -o libihmr_hip.so
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "ihmr_common.h"
#include "mano_lbs.h"
#include "sdf_collision.h"
#include "preprocess.h"
#include "mlp_infer.h"
#include "refine.h"
#include "encoder.h"
#include "evaluate.h"
#include "train.h"
#include "train_conv.h"
#include "copy_pack.h"

// bench.py's per-kernel timer -- the library's ONE piece of process-global state (include/ihmr_hip.h says so): the pointer and the
// list of pending event pairs are shared by every stream and thread of the process, guarded by g_timer_mutex; while no timer is set
// (the default) nothing here is touched.  (a,b) brackets one in-loop launch of kernel slot `k` (k < 0: an EMPTY pair, the cost of
// two event records)
static ihmr_kernel_timer* g_timer = nullptr;
static std::mutex g_timer_mutex;
struct TimedPair { hipEvent_t a, b; int k; };
static std::vector<TimedPair> g_pending;
static int timed_mark(hipEvent_t* ev, hipStream_t st) {
    HIP_TRY(hipEventCreate(ev));
    HIP_TRY(hipEventRecord(*ev, st));
    return 0;
}
// events around a group of back-to-back launches: timed_begin (empty pair + first mark), timed_next(k) after each launch
static int timed_begin(hipEvent_t* cur, hipStream_t st) {
    hipEvent_t e0, e1;
    if (int rc = timed_mark(&e0, st)) return rc;
    if (int rc = timed_mark(&e1, st)) return rc;
    { std::lock_guard<std::mutex> lk(g_timer_mutex); g_pending.push_back(TimedPair{e0, e1, -1}); }
    return timed_mark(cur, st);
}
static int timed_next(hipEvent_t* cur, int k, hipStream_t st) {
    hipEvent_t e;
    if (int rc = timed_mark(&e, st)) return rc;
    { std::lock_guard<std::mutex> lk(g_timer_mutex); g_pending.push_back(TimedPair{*cur, e, k}); }
    return timed_mark(cur, st);      // (a fresh start mark: every event belongs to exactly one pair)
}

// ------------------------------------------------------------------------------------------ model
template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static void derive_shape_constants(const float* Jreg, const float* v_template, const float* shapedirs,
                                   std::vector<float>& sd_t, std::vector<float>& J_t, std::vector<float>& J_sd) {
    sd_t.assign((size_t)10 * NV3, 0.f);
    for (int v = 0; v < NV; ++v)
        for (int k = 0; k < 3; ++k)
            for (int l = 0; l < 10; ++l) sd_t[(size_t)l * NV3 + 3 * v + k] = shapedirs[(v * 3 + k) * 10 + l];
    J_t.assign(48, 0.f);
    J_sd.assign(480, 0.f);
    for (int j = 0; j < NJ; ++j)
        for (int k = 0; k < 3; ++k) {
            double acc = 0.0;
            for (int v = 0; v < NV; ++v) acc += (double)Jreg[j * NV + v] * (double)v_template[3 * v + k];
            J_t[j * 3 + k] = (float)acc;
            for (int l = 0; l < 10; ++l) {
                double a2 = 0.0;
                for (int v = 0; v < NV; ++v) a2 += (double)Jreg[j * NV + v] * (double)shapedirs[(v * 3 + k) * 10 + l];
                J_sd[(j * 3 + k) * 10 + l] = (float)a2;
            }
        }
}

static void pack4(const float* src_rows /*[rows][2334]*/, int rows, std::vector<float>& dst) {
    dst.assign((size_t)rows * NVP * 4, 0.f);
    for (int r = 0; r < rows; ++r)
        for (int v = 0; v < NV; ++v)
            for (int k = 0; k < 3; ++k) dst[((size_t)r * NVP + v) * 4 + k] = src_rows[(size_t)r * NV3 + 3 * v + k];
}

template <typename T>
static int upload_as(T** dst, const std::vector<float>& src) {
    HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ihmr_mano_create(const ihmr_mano_arrays* h, ihmr_mano** out) {
    if (!h || !out) return -1;
    ihmr_mano* m = new ihmr_mano();
    memset(m, 0, sizeof(*m));
    std::vector<float> vt(h->v_template, h->v_template + NV3), pd(h->posedirs, h->posedirs + (size_t)NPF * NV3),
        wts(h->lbs_weights, h->lbs_weights + NV * NJ), jr(h->J_regressor, h->J_regressor + NJ * NV);
    std::vector<float> sd_t, J_t, J_sd;
    derive_shape_constants(h->J_regressor, h->v_template, h->shapedirs, sd_t, J_t, J_sd);
    std::vector<float> pm(48, 0.f);
    for (int i = 0; i < 45; ++i) pm[3 + i] = h->hands_mean[i];
    std::vector<int32_t> par(h->parents, h->parents + NJ), depth(NJ, 0), tips(h->tip_ids, h->tip_ids + IHMR_NUM_TIPS);
    int maxd = 0;
    for (int j = 1; j < NJ; ++j) {
        if (par[j] < 0 || par[j] >= j) { delete m; return -1; }
        depth[j] = depth[par[j]] + 1;
        if (depth[j] > maxd) maxd = depth[j];
    }
    std::vector<int32_t> start(NJ + 1, 0), wv;
    std::vector<float> ww;
    for (int j = 0; j < NJ; ++j) {
        for (int v = 0; v < NV; ++v)
            if (wts[v * NJ + j] != 0.f) { wv.push_back(v); ww.push_back(wts[v * NJ + j]); }
        start[j + 1] = (int32_t)wv.size();
    }
    // by vertex: the (up to) four non-zero weights in joint order (a zero weight adds an exact zero to the skinning sums, so
    // leaving the zeros out changes no bit); an asset with a denser vertex keeps the 16-joint loops
    std::vector<float> w4w((size_t)NV * 4, 0.f);
    std::vector<uint32_t> w4j(NV, 0u);
    int sparse4 = 1;
    for (int v = 0; v < NV; ++v) {
        int n = 0;
        for (int j = 0; j < NJ; ++j)
            if (wts[v * NJ + j] != 0.f) {
                if (n < 4) { w4w[(size_t)v * 4 + n] = wts[v * NJ + j]; w4j[v] |= (uint32_t)j << (8 * n); }
                ++n;
            }
        if (n > 4) sparse4 = 0;
    }
    std::vector<int32_t> seg_q, jseg(NJ + 1, 0);
    for (int j = 0; j < NJ; ++j) {
        jseg[j] = (int32_t)seg_q.size();
        for (int q = start[j]; q < start[j + 1]; q += LBS_SEG) seg_q.push_back(q);
    }
    jseg[NJ] = (int32_t)seg_q.size();
    seg_q.push_back((int32_t)wv.size());
    // segment s of joint j ends at min(next segment start, end of the joint's list)
    std::vector<int32_t> fsoa((size_t)3 * NFP, 0);
    for (int f = 0; f < NFP; ++f)
        for (int k = 0; k < 3; ++k) {
            const int32_t id = h->faces[(f < NF ? f : 0) * 3 + k];
            if (id < 0 || id >= NV) { delete m; return -1; }
            fsoa[(size_t)k * NFP + f] = id;
        }
    std::vector<uint32_t> fpk(NFP, 0u);
    for (int f = 0; f < NFP; ++f)
        fpk[f] = (uint32_t)fsoa[f] | ((uint32_t)fsoa[(size_t)NFP + f] << 10) | ((uint32_t)fsoa[(size_t)2 * NFP + f] << 20);
    std::vector<float> pd4, sd4, vt4;
    pack4(pd.data(), NPF, pd4);
    pack4(sd_t.data(), 10, sd4);
    pack4(vt.data(), 1, vt4);
    int rc = 0;
    rc |= upload_as(&m->pd4, pd4); rc |= upload_as(&m->sd4, sd4); rc |= upload_as(&m->vt4, vt4);
    rc |= upload(&m->v_template, vt); rc |= upload(&m->shapedirs_t, sd_t); rc |= upload(&m->posedirs, pd);
    rc |= upload(&m->J_template, J_t); rc |= upload(&m->J_shapedirs, J_sd); rc |= upload(&m->weights, wts);
    rc |= upload_as(&m->w4_w, w4w); rc |= upload(&m->w4_j, w4j);
    m->sparse4 = sparse4;
    rc |= upload(&m->pose_mean, pm); rc |= upload(&m->parents, par); rc |= upload(&m->depth, depth);
    rc |= upload(&m->tip_ids, tips); rc |= upload(&m->wj_start, start); rc |= upload(&m->wj_vert, wv);
    rc |= upload(&m->wj_w, ww); rc |= upload(&m->seg_q, seg_q); rc |= upload(&m->jseg_start, jseg); rc |= upload(&m->faces, fsoa); rc |= upload(&m->faces_pk, fpk); rc |= upload(&m->J_regressor, jr);
    m->max_depth = maxd;
    m->nnz = (int)wv.size();
    m->nseg = (int)seg_q.size() - 1;
    if (m->nseg > LBS_SEG_CAP) { ihmr_mano_destroy(m); return -1; }
    // lbs_bwd1 keeps the per-segment partial sums in dynamic LDS (48 B per segment on top of ~36 KB static): dense
    // weight matrices (up to ~960 segments) go past the default 64 KB per workgroup, so raise the cap to what is needed --
    // for every instantiation that is launched with dynamic LDS, and checked: a kernel whose static + dynamic LDS does not fit
    // the device is never launched (m->tail_fits = 0: ihmr_opt_run_stage falls back to the three separate launches)
    {
        const int dyn = m->nseg * 12 * (int)sizeof(float);
        hipError_t e1 = hipFuncSetAttribute((const void*)lbs_bwd1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipError_t e2 = hipFuncSetAttribute((const void*)lbs_bwd1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e1 != hipSuccess || e2 != hipSuccess) { ihmr_mano_destroy(m); return (int)(e1 != hipSuccess ? e1 : e2); }
        int dev = 0, lds_max = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        const void* tails[3] = {(const void*)opt_tail_kernel<true, true>, (const void*)opt_tail_kernel<true>, (const void*)opt_tail_kernel<false>};
        m->tail_fits = 1;
        const int tail_dyn = opt_tail_dynamic_lds(m->nseg);
        for (const void* k : tails) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, k) != hipSuccess || (long)fa.sharedSizeBytes + tail_dyn > (long)lds_max ||
                hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, tail_dyn) != hipSuccess)
                m->tail_fits = 0;
        }
        (void)hipGetLastError();
    }
    if (rc) return rc;
    *out = m;
    return 0;
}

extern "C" int ihmr_mano_destroy(ihmr_mano* m) {
    if (!m) return 0;
    void* ptrs[] = {m->v_template, m->shapedirs_t, m->posedirs, m->J_template, m->J_shapedirs, m->weights, m->pose_mean,
                    m->parents, m->depth, m->tip_ids, m->wj_start, m->wj_vert, m->wj_w, m->faces, m->J_regressor, m->pd4, m->sd4, m->vt4, m->seg_q, m->jseg_start,
                    m->w4_w, m->w4_j, m->faces_pk};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
    return 0;
}

extern "C" int ihmr_mano_update_shapedirs(ihmr_mano* m, const float* shapedirs_host) {
    if (!m || !shapedirs_host) return -1;
    std::vector<float> jr((size_t)NJ * NV), vt(NV3), sd_t, J_t, J_sd;
    HIP_TRY(hipMemcpy(jr.data(), m->J_regressor, jr.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vt.data(), m->v_template, vt.size() * 4, hipMemcpyDeviceToHost));
    derive_shape_constants(jr.data(), vt.data(), shapedirs_host, sd_t, J_t, J_sd);
    HIP_TRY(hipMemcpy(m->shapedirs_t, sd_t.data(), sd_t.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->J_shapedirs, J_sd.data(), J_sd.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> sd4;
    pack4(sd_t.data(), 10, sd4);
    HIP_TRY(hipMemcpy(m->sd4, sd4.data(), sd4.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------------ seam A
extern "C" size_t ihmr_mano_workspace_bytes(int N) { return lbs_ws_bytes(N); }

// the skinning launch (REUSE: the workspace holds v_posed of the current pose and shape parameters, see lbs_skin_kernel)
// small launches: four instead of eight hands per skin workgroup (half the chain per thread)
template <bool TWO_HAND, int MODE>
static void lbs_skin_launch_mode(const ihmr_mano* m, int N, int B, float* verts, float* joints, const LbsWork& wk, float* pose_off, hipStream_t st) {
    const bool small = N <= LBS_SMALL_MAX_HANDS;
    const int hg8 = 8 * (small ? LBS_HG_SMALL : LBS_HG);
    const dim3 skin_grid(8, 4 * ((N + hg8 - 1) / hg8));
    if (small) hipLaunchKernelGGL((lbs_skin_kernel<TWO_HAND, MODE, LBS_HG_SMALL>), skin_grid, dim3(LBS_THREADS), 0, st, *m, (const float*)wk.skel, N, B,
                                  verts, joints, wk.v_posed, pose_off);
    else hipLaunchKernelGGL((lbs_skin_kernel<TWO_HAND, MODE, LBS_HG>), skin_grid, dim3(LBS_THREADS), 0, st, *m, (const float*)wk.skel, N, B, verts,
                            joints, wk.v_posed, pose_off);
}
// skin modes of the callers: FULL (both blends), REUSE (v_posed kept: lbs_skin_kernel), FULL_STORE_P (both blends + the pose offsets P stored:
// the first iteration of a stage that moves the shape but not the finger pose), KEEP_P (the later iterations of such a stage: no pose rows read)
#define LBS_SKIN_FULL 0
#define LBS_SKIN_REUSE 1
#define LBS_SKIN_KEEP_P 2
#define LBS_SKIN_FULL_STORE_P 3
static int g_force_full_skin = 0;    // checker switch: ihmr_debug_force_full_skin
template <bool TWO_HAND>
static void lbs_skin_launch(const ihmr_mano* m, int mode, int N, int B, float* verts, float* joints, const LbsWork& wk, hipStream_t st) {
    if (g_force_full_skin && (mode == LBS_SKIN_KEEP_P || mode == LBS_SKIN_FULL_STORE_P)) mode = LBS_SKIN_FULL;
    if (mode == LBS_SKIN_REUSE) lbs_skin_launch_mode<TWO_HAND, LBS_MODE_REUSE>(m, N, B, verts, joints, wk, nullptr, st);
    else if (mode == LBS_SKIN_KEEP_P) lbs_skin_launch_mode<TWO_HAND, LBS_MODE_KEEP_P>(m, N, B, verts, joints, wk, wk.pose_off, st);
    else lbs_skin_launch_mode<TWO_HAND, LBS_MODE_FULL>(m, N, B, verts, joints, wk, mode == LBS_SKIN_FULL_STORE_P ? wk.pose_off : nullptr, st);
}
extern "C" int ihmr_debug_force_full_skin(int force) {
    const int prev = g_force_full_skin;
    g_force_full_skin = force ? 1 : 0;
    return prev;
}

static void lbs_forward_launch(const ihmr_mano* m, bool two_hand, const float* orient, const float* pose, const float* betas,
                               const float* trans, int N, int B, float* verts, float* joints, const LbsWork& wk, hipStream_t st) {
    if (two_hand) {
        hipLaunchKernelGGL(lbs_skel_kernel<true>, dim3(N), dim3(192), 0, st, *m, orient, pose, betas, trans, B, wk.skel, joints);
        lbs_skin_launch<true>(m, LBS_SKIN_FULL, N, B, verts, joints, wk, st);
    } else {
        hipLaunchKernelGGL(lbs_skel_kernel<false>, dim3(N), dim3(192), 0, st, *m, orient, pose, betas, trans, B, wk.skel, joints);
        lbs_skin_launch<false>(m, LBS_SKIN_FULL, N, B, verts, joints, wk, st);
    }
}

static int g_bwd2_streaming = 0;     // checker switch: ihmr_debug_force_lbs_bwd2_streaming
extern "C" int ihmr_debug_force_lbs_bwd2_streaming(int force) {
    const int prev = g_bwd2_streaming;
    g_bwd2_streaming = force ? 1 : 0;
    return prev;
}

static void lbs_backward_launch(const ihmr_mano* m, bool two_hand, int N, int B, const float* d_verts, const float* d_joints,
                                float* d_orient, float* d_pose, float* d_betas, float* d_trans, int need_mask, const LbsWork& wk,
                                hipStream_t st, bool bwd1_done = false) {
    const size_t part_lds = (size_t)m->nseg * 12 * sizeof(float);
    if (bwd1_done) {}        // (the per-hand part ran inside opt_tail_kernel)
    else if (two_hand)
        hipLaunchKernelGGL(lbs_bwd1_kernel<true>, dim3(N), dim3(LBS_THREADS), part_lds, st, *m, wk, B, d_verts, d_joints, d_orient,
                           d_betas, d_trans, need_mask);
    else
        hipLaunchKernelGGL(lbs_bwd1_kernel<false>, dim3(N), dim3(LBS_THREADS), part_lds, st, *m, wk, B, d_verts, d_joints, d_orient,
                           d_betas, d_trans, need_mask);
    if (need_mask & 2) {
        // LDS-tiled form from LBS_B2_MIN_HANDS = 256 hands on (one batch of 64 samples = 128 hands: 2 x 25 workgroups are too few; the
        // streaming form stays there); the same bits either way
        if (N >= LBS_B2_MIN_HANDS && !g_bwd2_streaming) hipLaunchKernelGGL(lbs_bwd2_lds_kernel, dim3((N + 63) / 64, LBS_KG), dim3(320), 0, st, *m, wk, N);
        else hipLaunchKernelGGL(lbs_bwd2_kernel, dim3(5, (N + 31) / 32, LBS_KG), dim3(64), 0, st, *m, wk, N);
        if (two_hand) hipLaunchKernelGGL(lbs_bwd3_kernel<true>, dim3(N), dim3(64), 0, st, wk, N, B, d_pose);
        else hipLaunchKernelGGL(lbs_bwd3_kernel<false>, dim3(N), dim3(64), 0, st, wk, N, B, d_pose);
    }
}

extern "C" int ihmr_mano_lbs_fwd(const ihmr_mano* m, const float* orient, const float* pose, const float* betas, int N,
                                 float* verts, float* joints, void* workspace, void* stream) {
    if (!m || N <= 0 || !workspace) return -1;
    lbs_forward_launch(m, false, orient, pose, betas, nullptr, N, 0, verts, joints, lbs_carve(workspace, N), (hipStream_t)stream);
    return (int)hipGetLastError();
}

extern "C" int ihmr_mano_lbs_bwd(const ihmr_mano* m, int N, const void* workspace, const float* d_verts, const float* d_joints,
                                 float* d_orient, float* d_pose, float* d_betas, int need_mask, void* stream) {
    if (!m || N <= 0 || !workspace) return -1;
    lbs_backward_launch(m, false, N, 0, d_verts, d_joints, d_orient, d_pose, d_betas, nullptr, need_mask,
                        lbs_carve(const_cast<void*>(workspace), N), (hipStream_t)stream);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ seam B
// (tail: the SoA and the packed copies of the caller's two face arrays)
// Compute units of the CURRENT device, cached per device id in relaxed atomics (concurrent first calls store the same value: no lock, no
// data race).  The persistent grid of sdf_dist_kernel and the Stream-K worker count of ihmr_conv_igemm derive from it; the latter fixes
// the K partition, so results are bit-stable per device MODEL (same CU count), not across models.
#include <atomic>
static int device_cu_count(int* cus_out) {
    static std::atomic<int> cache[64];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int cus = cache[dev & 63].load(std::memory_order_relaxed);
    if (cus == 0) {
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus <= 0) cus = 256;
        cache[dev & 63].store(cus, std::memory_order_relaxed);
    }
    *cus_out = cus;
    return 0;
}

extern "C" size_t ihmr_sdf_workspace_bytes(int B) { return sdf_ws_bytes(2 * B) + (size_t)2 * NFP * 4 * 4 + 256; }

static int g_collect_stats = 0;

static int sdf_launch(const VertLayout& vl, const int32_t* faces_r_soa, const int32_t* faces_l_soa, const uint32_t* fpk_r,
                      const uint32_t* fpk_l, int B, SdfWorkspace ws, float robustifier, float* loss, float* per_vert, float* origin, float* dval, bool dense, hipStream_t st) {
    // the inside-voxel counter is zero on entry (faces_to_soa_kernel on seam B, the skeleton kernel of the iteration
    // on seam C); the prep kernel appends to it
    // an inside-list entry is (hand << 16) | voxel with bit 31 reserved (SDF_ENT_REFUSED): hand ids stay below 32768
    if (B <= 0 || 2 * B > SDF_MAX_HANDS) return -1;
    // small launches: the 1024-thread form (half the chain per thread), see sdf_collision.h
    const bool small = 2 * B <= SDF_PREP_SMALL_MAX_HANDS;
    ws.fpk[0] = fpk_r; ws.fpk[1] = fpk_l; ws.B = B;
    const bool timed = g_timer != nullptr;
    hipEvent_t tcur;
    if (timed) { if (int rc = timed_begin(&tcur, st)) return rc; }
    if (dense)
        hipLaunchKernelGGL((sdf_prep_kernel<true, SDF_PREP_THREADS_LARGE>), dim3(2 * B), dim3(SDF_PREP_THREADS_LARGE), 0, st, vl, B, faces_r_soa,
                           faces_l_soa, ws, g_collect_stats);
    else if (small)
        hipLaunchKernelGGL((sdf_prep_kernel<false, SDF_PREP_THREADS_SMALL>), dim3(2 * B), dim3(SDF_PREP_THREADS_SMALL), 0, st, vl, B, faces_r_soa,
                           faces_l_soa, ws, g_collect_stats);
    else
        hipLaunchKernelGGL((sdf_prep_kernel<false, SDF_PREP_THREADS_LARGE>), dim3(2 * B), dim3(SDF_PREP_THREADS_LARGE), 0, st, vl, B, faces_r_soa,
                           faces_l_soa, ws, g_collect_stats);
    // persistent grid: as many workgroups as the GPU holds at once; they pull work units from a queue (sdf_collision.h)
    int cus = 0;
    if (int rc = device_cu_count(&cus)) return rc;
    const int dist_blocks = SDF_DIST_WG_PER_CU * cus;
    if (timed) { if (int rc = timed_next(&tcur, IHMR_TIMED_SDF_PREP, st)) return rc; }
    int nblk = dist_blocks;
#ifdef IHMR_TUNING_BUILD
    if (const char* e = getenv("IHMR_DIST_BLOCKS_PER_SAMPLE")) nblk = std::max(64, std::min(dist_blocks, atoi(e) * B));
#endif
    if (g_collect_stats) hipLaunchKernelGGL(sdf_dist_kernel<true>, dim3(nblk), dim3(SDF_THREADS), 0, st, ws);
    else hipLaunchKernelGGL(sdf_dist_kernel<false>, dim3(nblk), dim3(SDF_THREADS), 0, st, ws);
    // (the launch the refinement really runs is the one that is timed: a second launch would find the work cursor spent)
    if (timed) { if (int rc = timed_next(&tcur, IHMR_TIMED_SDF_DIST, st)) return rc; (void)hipEventDestroy(tcur); }
    if (loss)
        hipLaunchKernelGGL(sdf_sample_kernel, dim3(B), dim3(SDF_SAMPLE_THREADS), 0, st, vl, ws, robustifier, loss, per_vert, origin,
                           dval, B);
    return (int)hipGetLastError();
}

// seam-B callers hand over (F,3) int32 AoS faces on the device; the SoA copy lives in the workspace tail
__global__ void faces_to_soa_kernel(const int32_t* __restrict__ aos, int32_t* __restrict__ soa, uint32_t* __restrict__ packed, int* zero8) {
    if (zero8 && blockIdx.x == 0 && threadIdx.x < SDF_NZERO) sdf_zero_counter(zero8, (int)threadIdx.x);  // inside-voxel counters, work cursors
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= NFP) return;
    const int s = f < NF ? f : 0;
    uint32_t pk = 0;
    for (int k = 0; k < 3; ++k) {
        const int32_t id = min(max(aos[s * 3 + k], 0), NV - 1);      // (an out-of-range id of the caller cannot leave the vertex array)
        soa[k * NFP + f] = id;
        pk |= (uint32_t)id << (10 * k);
    }
    packed[f] = pk;
}

extern "C" int ihmr_sdf_collision_ex(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                                     float robustifier, const ihmr_sdf_options* options, float* loss, float* per_vert,
                                     float* origin_scale, float* dval, void* workspace, void* stream) {
    if (!workspace || B <= 0 || 2 * B > SDF_MAX_HANDS) return -1;
    hipStream_t st = (hipStream_t)stream;
    SdfWorkspace ws = sdf_carve(workspace, 2 * B);
    if (options) {
        ws.align_corners = options->align_corners ? 1 : 0;
        if (options->loss_divisor > 0.f) ws.loss_div = options->loss_divisor;
        ws.swap_xz = options->swap_xz ? 1 : 0;
    }
    int32_t* soa = (int32_t*)((char*)workspace + sdf_ws_bytes(2 * B));
    uint32_t* pk = (uint32_t*)(soa + 6 * NFP);
    hipLaunchKernelGGL(faces_to_soa_kernel, dim3((NFP + 255) / 256), dim3(256), 0, st, faces_right, soa, pk, ws.inside_count);
    hipLaunchKernelGGL(faces_to_soa_kernel, dim3((NFP + 255) / 256), dim3(256), 0, st, faces_left, soa + 3 * NFP, pk + NFP, (int*)nullptr);
    VertLayout vl{hand_verts, (long)2 * NV3, (long)NV3};
    return sdf_launch(vl, soa, soa + 3 * NFP, pk, pk + NFP, B, ws, robustifier, loss, per_vert, origin_scale, dval, false, st);
}

extern "C" int ihmr_sdf_collision(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                                  float robustifier, float* loss, float* per_vert, float* origin_scale, float* dval,
                                  void* workspace, void* stream) {
    return ihmr_sdf_collision_ex(faces_right, faces_left, hand_verts, B, robustifier, nullptr, loss, per_vert, origin_scale, dval,
                                 workspace, stream);
}

extern "C" int ihmr_sdf_dense_grid(const int32_t* faces_right, const int32_t* faces_left, const float* hand_verts, int B,
                                   float* phi, void* workspace, void* stream) {
    if (!workspace || B <= 0 || 2 * B > SDF_MAX_HANDS) return -1;
    hipStream_t st = (hipStream_t)stream;
    SdfWorkspace ws = sdf_carve(workspace, 2 * B);
    int32_t* soa = (int32_t*)((char*)workspace + sdf_ws_bytes(2 * B));
    uint32_t* pk = (uint32_t*)(soa + 6 * NFP);
    hipLaunchKernelGGL(faces_to_soa_kernel, dim3((NFP + 255) / 256), dim3(256), 0, st, faces_right, soa, pk, ws.inside_count);
    hipLaunchKernelGGL(faces_to_soa_kernel, dim3((NFP + 255) / 256), dim3(256), 0, st, faces_left, soa + 3 * NFP, pk + NFP, (int*)nullptr);
    VertLayout vl{hand_verts, (long)2 * NV3, (long)NV3};
    int rc = sdf_launch(vl, soa, soa + 3 * NFP, pk, pk + NFP, B, ws, 0.f, nullptr, nullptr, nullptr, nullptr, true, st);
    if (rc) return rc;
    // workspace order is hand = hnd*B + b; the caller's grid is (B,2,...)
    for (int b = 0; b < B; ++b)
        for (int hnd = 0; hnd < 2; ++hnd)
            HIP_TRY(hipMemcpyAsync(phi + ((size_t)b * 2 + hnd) * SDF_NVOX, ws.phi + ((size_t)hnd * B + b) * SDF_NVOX,
                                   (size_t)SDF_NVOX * 4, hipMemcpyDeviceToDevice, st));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ seam C
extern "C" size_t ihmr_opt_workspace_bytes(int B) { return opt_ws_bytes(B); }

// `prev` = the Adam step of the previous iteration (group < 0: none), applied at the head of the skeleton kernel
static const ParamStep kNoStep{0, 0.f, 0.f, 1.f, -1, 0, 0};
// skin_mode: LBS_SKIN_REUSE = the workspace holds v_posed of the current pose and shape parameters (see lbs_skin_kernel); < 0: no skin launch
// lists: temporal candidate lists of the collision kernels -- 0 off (single-shot callers), 1 reuse while valid, 2 rebuild now, 3 = the first
// iteration of a stage whose caller vouches for the lists of the previous stage (ihmr_opt_stage::keep_lists): the static-hand bookkeeping
// starts over, a hand's lists stay while its own displacement test passes
// the collision workspace of the fused loop with the switches of this call
// static_mask: bit 0 / 1 = the right / left hands have had bit-identical vertices since the stage's first iteration (SdfWorkspace::static_mask)
static SdfWorkspace opt_sdf_ws(const ihmr_opt_io* io, const OptWork& wk, int B, int lists, int static_mask = 0) {
    SdfWorkspace ws = sdf_carve(wk.sdf_ws, 2 * B, true);
    ws.list_mode = (lists != 0 && !io->sdf_no_candidate_lists) ? 1 : 0;
    ws.force_rebuild = lists == 2 ? 1 : 0;
    ws.static_stage = (ws.list_mode && io->sdf_no_static_reuse != 1) ? (static_mask & 3) : 0;
    ws.static_mask = lists >= 2 ? 0 : ws.static_stage;
    ws.moving_box = (static_mask >> 2) & ws.static_stage;      // (bits 2-3 of the caller's mask: sides that only translate)
    ws.align_corners = io->sdf_align_corners ? 1 : 0;
    if (io->sdf_loss_divisor > 0.f) ws.loss_div = io->sdf_loss_divisor;
    ws.swap_xz = io->sdf_swap_xz ? 1 : 0;
    return ws;
}

// head = the Adam + skeleton launch, skin = the skinning launch, tail = the sampling + loss launch (a caller that fuses them into other launches skips them)
static int opt_forward(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, const OptWork& wk, int B,
                       const ihmr_opt_weights& w, const ParamStep& prev, hipStream_t st, int need_cam = 0, int skin_mode = LBS_SKIN_FULL,
                       int lists = 0, bool head = true, bool tail = true, int static_mask = 0, const MlpSelect* sel = nullptr) {
    if (head)
        hipLaunchKernelGGL(opt_adam_skel_kernel, dim3(B), dim3(384), 0, st, *m, *io, wk, B, prev, sdf_carve(wk.sdf_ws, 2 * B, true).inside_count);
    // (skin_mode < 0: the tail launch of the previous iteration has skinned the stored v_posed with the new skeletons, opt_tail_kernel<true, true>)
    if (skin_mode >= 0) lbs_skin_launch<true>(m, skin_mode, 2 * B, B, io->verts, wk.joints_raw, wk.lbs, st);
    SdfWorkspace ws = opt_sdf_ws(io, wk, B, lists, static_mask);
    VertLayout vl{io->verts, (long)NV3, (long)B * NV3};
    int rc = sdf_launch(vl, m->faces, m_left ? m_left->faces : m->faces, m->faces_pk, m_left ? m_left->faces_pk : m->faces_pk, B, ws, 0.f,
                        nullptr, nullptr, nullptr, nullptr, false, st);
    if (rc) return rc;
    // collision sampling (loss_batch[2], masked by hand type; gradient -> g_verts) and the joint losses in one launch
    MlpSelect no_sel;
    memset(&no_sel, 0, sizeof(no_sel));
    if (tail) hipLaunchKernelGGL(opt_sample_loss_kernel, dim3(B), dim3(SDF_SAMPLE_THREADS), 0, st, *io, wk, B, w, vl, ws, need_cam, sel ? *sel : no_sel);
    return (int)hipGetLastError();
}

extern "C" int ihmr_opt_forward_losses(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                       const ihmr_opt_weights* w, void* stream) {
    if (!m || !io || !w || B <= 0 || 2 * B > SDF_MAX_HANDS) return -1;
    hipStream_t st = (hipStream_t)stream;
    OptWork wk = opt_carve(io->workspace, B);
    int rc = opt_forward(m, m_left, io, wk, B, *w, kNoStep, st);
    if (rc) return rc;
    return (int)hipGetLastError();
}

extern "C" int ihmr_opt_run_stage(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                  const ihmr_opt_weights* w, const ihmr_opt_stage* sg, void* stream) {
    if (!m || !io || !w || !sg || B <= 0 || 2 * B > SDF_MAX_HANDS || sg->n_iters <= 0 || sg->save_freq <= 0) return -1;
    if (sg->param_mask <= 0 || sg->param_mask > 255 || sg->select_loss < 0 || sg->select_loss > 2) return -1;
    if (sg->optimizer != IHMR_OPTIM_ADAM && sg->optimizer != IHMR_OPTIM_SGD) return -1;
    hipStream_t st = (hipStream_t)stream;
    OptWork wk = opt_carve(io->workspace, B);
    const int pm = sg->param_mask;
    // what the LBS backward has to deliver (bit0 orient, bit1 pose, bit2 betas, bit3 trans)
    const int need_mask = ((pm & (IHMR_PB_ORIENT_R | IHMR_PB_ORIENT_L)) ? 1 : 0) | ((pm & (IHMR_PB_POSE_R | IHMR_PB_POSE_L)) ? 2 : 0) |
                          ((pm & (IHMR_PB_SHAPE_R | IHMR_PB_SHAPE_L)) ? 4 : 0) | ((pm & IHMR_PB_TRANS) ? 8 : 0);
    const int need_cam = (pm & IHMR_PB_CAM) ? 1 : 0, sgd = sg->optimizer == IHMR_OPTIM_SGD;
    int S = 0;
    ParamStep step{0, 0.f, 0.f, 1.f, -1, 1, 0};   // iteration 0: no step yet, zero the optimizer state
    // a stage that moves neither the finger pose nor the shape keeps v_posed: computed in its first iteration, reused after
    const bool vposed_fixed = (pm & (IHMR_PB_POSE_R | IHMR_PB_POSE_L | IHMR_PB_SHAPE_R | IHMR_PB_SHAPE_L)) == 0;
    // ... and a stage that moves the shape but not the finger pose keeps the pose offsets P: stored by its first iteration's skinning, reused
    // after (lbs_skin_kernel MODE KEEP_P: the 1.8 MB pose basis is not read again; the same bits, test_skin_keeps_pose_offsets_bit_identically)
    const bool pose_fixed = (pm & (IHMR_PB_POSE_R | IHMR_PB_POSE_L)) == 0;
    const int keep_mode = vposed_fixed ? LBS_SKIN_REUSE : (pose_fixed ? LBS_SKIN_KEEP_P : LBS_SKIN_FULL);
    const int first_mode = (!vposed_fixed && pose_fixed) ? LBS_SKIN_FULL_STORE_P : LBS_SKIN_FULL;
    // The tail of an iteration -- sampling + losses, LBS backward of both hands, and in the stages that do not move the finger pose also the
    // optimizer step + next skeletons -- is ONE launch per sample (opt_tail_kernel): 4 launches per iteration instead of 6 (finger-pose
    // stage, whose backward continues with a batch-wide GEMM: 7 instead of 8); in a stage that keeps v_posed the same launch also skins
    // the next iteration's vertices: 3 launches (skin_mode < 0 below)
    const bool fused_tail = need_mask != 0 && !io->no_fused_tail && m->tail_fits;
    // Hands whose vertices cannot change during this stage (SdfWorkspace::static_mask): the right hand when none of its own blocks is
    // refined; the left hand when neither its own blocks, nor the translation, nor the right hand's shape (the left hand is shifted by
    // trans + J_r[0] - J_l[0], optimize_model.py:217-224) is.  opt_default's translation stage: the right hands.
    int static_mask = ((pm & (IHMR_PB_ORIENT_R | IHMR_PB_POSE_R | IHMR_PB_SHAPE_R)) ? 0 : 1) |
                      ((pm & (IHMR_PB_ORIENT_L | IHMR_PB_POSE_L | IHMR_PB_SHAPE_L | IHMR_PB_TRANS | IHMR_PB_SHAPE_R)) ? 0 : 2);
    // Round 5: a left hand that the stage only TRANSLATES (the translation stage of opt_default: trans alone moves) is static in its own
    // normalised frame -- its box follows it, everything inside the box stays: treated as static with a moving box (SdfWorkspace::moving_box;
    // bits 2-3 of the mask).  The kept grid is the first iteration's; a recomputation would differ by the rounding of the translated
    // vertices, so unlike the static reuse this is not bit-identical to the from-scratch path (sdf_no_static_reuse = 2 switches it off).
    if ((pm & IHMR_PB_TRANS) && !(pm & (IHMR_PB_ORIENT_L | IHMR_PB_POSE_L | IHMR_PB_SHAPE_L | IHMR_PB_SHAPE_R)) && io->sdf_no_static_reuse == 0)
        static_mask |= 2 | (2 << 2);
    const bool pose_stage = (need_mask & 2) != 0;
    const size_t tail_lds = (size_t)opt_tail_dynamic_lds(m->nseg);
    const int lists_first = sg->keep_lists ? 3 : 2;
    for (int it = 0; it < sg->n_iters; ++it) {
        // the first iteration of a stage starts the candidate lists over (the workspace is the caller's memory: whatever it holds, a stage
        // is self-contained) -- unless the caller vouches for them (keep_lists): then a hand keeps its lists while the prep kernel's
        // displacement test against the reference pose they were built at passes, whether an optimizer step or the previous stage's
        // select step moved the hand
        const double t = (double)(it + 1);
        const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
        const ParamStep next{pm, w->shape_reg, sgd ? sg->lr : (float)((double)sg->lr / bc1), (float)sqrt(bc2),
                             (it % sg->save_freq == 0) ? S++ : -1, 0, sgd};
        if (fused_tail) {
            // head (optimizer step of the previous iteration + skeletons): stand-alone in the first iteration (zero the optimizer state,
            // first skeletons) and in the finger-pose stage; otherwise the tail of iteration it - 1 has done it
            // ... and in a stage that keeps v_posed (translation, orientation) the tail has skinned the next vertices as well: 3 launches
            int rc = opt_forward(m, m_left, io, wk, B, *w, step, st, need_cam, it == 0 ? first_mode : (vposed_fixed ? -1 : keep_mode), it == 0 ? lists_first : 1,
                                 /*head=*/it == 0 || pose_stage, /*tail=*/false, static_mask);
            if (rc) return rc;
            SdfWorkspace ws = opt_sdf_ws(io, wk, B, it == 0 ? lists_first : 1, static_mask);
            VertLayout vl{io->verts, (long)NV3, (long)B * NV3};
            hipEvent_t tcur;
            const bool timed = g_timer != nullptr;
            if (timed) { if (int rc2 = timed_begin(&tcur, st)) return rc2; }
            if (vposed_fixed && it + 1 < sg->n_iters)
                hipLaunchKernelGGL((opt_tail_kernel<true, true>), dim3(B), dim3(SDF_SAMPLE_THREADS), tail_lds, st, *m, *io, wk, B, *w, vl, ws, need_cam, need_mask,
                                   next, ws.inside_count);
            else if (!pose_stage && it + 1 < sg->n_iters)
                hipLaunchKernelGGL(opt_tail_kernel<true>, dim3(B), dim3(SDF_SAMPLE_THREADS), tail_lds, st, *m, *io, wk, B, *w, vl, ws, need_cam, need_mask,
                                   next, ws.inside_count);
            else
                hipLaunchKernelGGL(opt_tail_kernel<false>, dim3(B), dim3(SDF_SAMPLE_THREADS), tail_lds, st, *m, *io, wk, B, *w, vl, ws, need_cam, need_mask,
                                   next, ws.inside_count);
            if (timed) { if (int rc2 = timed_next(&tcur, IHMR_TIMED_OPT_TAIL, st)) return rc2; (void)hipEventDestroy(tcur); }
            if (pose_stage)
                lbs_backward_launch(m, true, 2 * B, B, wk.g_verts, wk.g_joints, wk.g_orient, wk.g_pose, wk.g_shape, wk.g_trans, need_mask,
                                    wk.lbs, st, /*bwd1_done=*/true);
        } else {
            int rc = opt_forward(m, m_left, io, wk, B, *w, step, st, need_cam, it == 0 ? first_mode : keep_mode, it == 0 ? lists_first : 1, true, true, static_mask);   // applies the step of iteration it - 1 first
            if (rc) return rc;
            if (need_mask)
                lbs_backward_launch(m, true, 2 * B, B, wk.g_verts, wk.g_joints, wk.g_orient, wk.g_pose, wk.g_shape, wk.g_trans, need_mask,
                                    wk.lbs, st);
        }
        step = next;
    }
    hipLaunchKernelGGL(opt_adam_kernel, dim3(B), dim3(128), 0, st, *io, wk, B, step);
    hipLaunchKernelGGL(opt_select_kernel, dim3((B + 63) / 64), dim3(64), 0, st, *io, B, S, *sg);
    return (int)hipGetLastError();
}

extern "C" int ihmr_opt_set_params(const ihmr_opt_io* io, const float* final_params, int B, void* stream) {
    if (!io || !final_params || B <= 0) return -1;
    hipLaunchKernelGGL(opt_unpack_params_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, *io, final_params, B);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ IHMR-MLP inference glue (mlp_infer.h)
// [256 B header | hidden activations (B, 512 + 256 + 128) | raw joints of every sample's accepted state (B, 42, 3)]
extern "C" size_t ihmr_mlp_workspace_bytes(int B) { return ((size_t)B * (512 + 256 + 128 + 126) * sizeof(float) + 256 + 255) & ~(size_t)255; }
static float* mlp_acc_joints(void* workspace, int B) { return (float*)((char*)workspace + 256) + (size_t)B * (512 + 256 + 128); }

static int mlp_fill_select(MlpSelect& s, const ihmr_mlp_tables* t, const ihmr_mlp_stage* stage, int mode, void* workspace) {
    memset(&s, 0, sizeof(s));
    s.mode = mode;
    if (mode == 2) {
        if (!stage || stage->n_filter < 0 || stage->n_filter > 4 || stage->select_loss < 0 || stage->select_loss > 2) return -1;
        s.n_filter = stage->n_filter;
        for (int f = 0; f < stage->n_filter; ++f) {
            if (stage->filter_loss[f] < 0 || stage->filter_loss[f] > 2) return -1;
            s.filter_loss[f] = stage->filter_loss[f];
            s.filter_factor[f] = stage->filter_factor[f];
        }
        s.select_loss = stage->select_loss;
    }
    s.idx = (const long long*)t->idx; s.new_params = t->new_params; s.img_feat = t->img_feat;
    s.data_idxs_all = t->data_idxs_all; s.img_feat_all = t->img_feat_all; s.prev_final = t->prev_final; s.prev_loss = t->prev_loss;
    s.final_out = t->final_params; s.kept = t->kept;
    return 0;
}

extern "C" int ihmr_mlp_stage_head(const ihmr_mlp_net* net, const ihmr_mlp_tables* t, const ihmr_opt_io* io, int B, void* workspace,
                                   void* stream) {
    if (!net || !t || !io || !workspace || B <= 0 || net->k_out <= 0 || net->k_out > 122) return -1;
    MlpHeadArgs a;
    memset(&a, 0, sizeof(a));
    a.feat = t->img_feat; a.prev = t->final_params;
    for (int l = 0; l < 4; ++l) { a.w[l] = net->w[l]; a.b[l] = net->b[l]; a.ldw[l] = net->ldw[l]; if (!a.w[l] || !a.b[l]) return -1; }
    if (a.ldw[0] < 512 || a.ldw[1] < 256 || a.ldw[2] < 128 || a.ldw[3] < ((net->k_out + 15) & ~15)) return -1;
    a.kout = net->k_out;
    for (int j = 0; j < net->k_out; ++j) {
        if (net->col[j] < 0 || net->col[j] >= 122) return -1;
        a.col[j] = (unsigned char)net->col[j];
    }
    float* h = (float*)((char*)workspace + 256);
    a.h[0] = h; a.h[1] = h + (size_t)B * 512; a.h[2] = a.h[1] + (size_t)B * 256;
    a.new_params = t->new_params;
    a.B = B;
    const int tr = (B + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_layer_kernel<0>, dim3(std::min(tr * 32, MLPI_MAX_WG)), dim3(MLPI_THREADS), 0, st, a, *io);
    hipLaunchKernelGGL(mlp_layer_kernel<1>, dim3(std::min(tr * 16, MLPI_MAX_WG)), dim3(MLPI_THREADS), 0, st, a, *io);
    hipLaunchKernelGGL(mlp_layer_kernel<2>, dim3(std::min(tr * 8, MLPI_MAX_WG)), dim3(MLPI_THREADS), 0, st, a, *io);
    hipLaunchKernelGGL(mlp_layer_kernel<3>, dim3(std::min(tr * ((a.kout + 15) / 16), MLPI_MAX_WG)), dim3(MLPI_THREADS), 0, st, a, *io);
    return (int)hipGetLastError();
}

extern "C" int ihmr_mlp_forward_select(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B, const ihmr_opt_weights* w,
                                       const ihmr_mlp_tables* t, const ihmr_mlp_stage* stage, int mode, void* workspace, void* stream) {
    if (!m || !io || !w || !workspace || B <= 0 || 2 * B > SDF_MAX_HANDS || mode < 0 || mode > 3 || (mode && !t)) return -1;
    MlpSelect sel;
    memset(&sel, 0, sizeof(sel));
    // mode 3 (round 6) = mode 2 for a stage that moves neither finger pose nor shape while the workspace still holds v_posed of exactly
    // these finger poses and shapes (the caller's bookkeeping: ihmr_amd/mlp_model.py): the skinning launch skips both blends
    // (lbs_skin_kernel MODE REUSE: the stored values are the bits a recomputation gives)
    const int skin_mode = mode == 3 ? LBS_SKIN_REUSE : LBS_SKIN_FULL;
    if (mode == 3) mode = 2;
    if (mode) { if (int rc = mlp_fill_select(sel, t, stage, mode, workspace)) return rc; }
    OptWork wk = opt_carve(io->workspace, B);
    if (mode) { sel.joints_now = wk.joints_raw; sel.acc_joints = mlp_acc_joints(workspace, B); }
    // The evaluations of one test() call move the hands by a stage's residual at a time: the collision kernels keep their per-voxel
    // candidate lists from one evaluation to the next (valid while a hand stays within the slack of the pose its lists were built at,
    // checked per hand and evaluation; rebuilt otherwise) -- started over by the evaluation that opens the batch.  No static-hand
    // reuse here: a rejected update falls back to parameters the vertex buffers no longer hold.
    return opt_forward(m, m_left, io, wk, B, *w, kNoStep, (hipStream_t)stream, 0, skin_mode, mode == 1 ? 2 : 1, true, true, 0, mode ? &sel : nullptr);
}

// the evaluation of a stage that moved ONLY the camera: one small launch (mlp_camera_select_kernel, refine.h)
extern "C" int ihmr_mlp_camera_select(const ihmr_opt_io* io, int B, const ihmr_opt_weights* w, const ihmr_mlp_tables* t,
                                      const ihmr_mlp_stage* stage, void* workspace, void* stream) {
    if (!io || !w || !t || !stage || !workspace || B <= 0) return -1;
    MlpSelect sel;
    if (int rc = mlp_fill_select(sel, t, stage, 2, workspace)) return rc;
    OptWork wk = opt_carve(io->workspace, B);
    sel.joints_now = nullptr; sel.acc_joints = nullptr;          // (the accepted joints do not change: the camera moves no joint)
    wk.joints_raw = mlp_acc_joints(workspace, B);                 // the loss wave reads the accepted state's raw joints
    hipLaunchKernelGGL(mlp_camera_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *io, wk, B, *w, sel);
    return (int)hipGetLastError();
}

// skeletons + skinning of the parameters in `io` only (no collision term, no losses): the annotation's meshes of the export
extern "C" int ihmr_opt_forward_verts(const ihmr_mano* m, const ihmr_opt_io* io, int B, void* stream) {
    if (!m || !io || B <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    OptWork wk = opt_carve(io->workspace, B);
    hipLaunchKernelGGL(opt_adam_skel_kernel, dim3(B), dim3(384), 0, st, *m, *io, wk, B, kNoStep, sdf_carve(wk.sdf_ws, 2 * B, true).inside_count);
    lbs_skin_launch<true>(m, LBS_SKIN_FULL, 2 * B, B, io->verts, wk.joints_raw, wk.lbs, st);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ stage graphs
// A stage is ~8 launches x n_iters with no host decision inside: capture it once into a hipGraph (per model
// instance and stage -- the io pointers and the per-iteration Adam constants are baked into the nodes) and
// replay it, so the host cost per stage drops from ~n_iters x 8 launches to one graph launch.
struct ihmr_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

template <typename F>
static int capture_graph(F&& enqueue, ihmr_graph** out) {
    hipStream_t cs;
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    ihmr_kernel_timer* keep = g_timer;
    g_timer = nullptr;  // no event records inside a capture
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    int rc = e == hipSuccess ? enqueue(cs) : (int)e;
    hipGraph_t g = nullptr;
    hipError_t e2 = hipStreamEndCapture(cs, &g);
    g_timer = keep;
    (void)hipStreamDestroy(cs);
    if (rc) return rc;
    if (e2 != hipSuccess) return (int)e2;
    ihmr_graph* h = new ihmr_graph();
    h->graph = g;
    hipError_t e3 = hipGraphInstantiate(&h->exec, g, nullptr, nullptr, 0);
    if (e3 != hipSuccess) { (void)hipGraphDestroy(g); delete h; return (int)e3; }
    *out = h;
    return 0;
}

extern "C" int ihmr_opt_stage_graph_create(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                           const ihmr_opt_weights* w, const ihmr_opt_stage* stage, ihmr_graph** out) {
    if (!out) return -1;
    return capture_graph([&](hipStream_t cs) { return ihmr_opt_run_stage(m, m_left, io, B, w, stage, (void*)cs); }, out);
}

extern "C" int ihmr_opt_forward_graph_create(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                             const ihmr_opt_weights* w, ihmr_graph** out) {
    if (!out) return -1;
    return capture_graph([&](hipStream_t cs) { return ihmr_opt_forward_losses(m, m_left, io, B, w, (void*)cs); }, out);
}

extern "C" int ihmr_graph_launch(ihmr_graph* g, void* stream) {
    if (!g) return -1;
    return (int)hipGraphLaunch(g->exec, (hipStream_t)stream);
}

extern "C" int ihmr_graph_destroy(ihmr_graph* g) {
    if (!g) return 0;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return 0;
}

// diagnostics (synchronises): one forward + losses with the SDF work counters switched on.
// out[0] = (voxel, triangle) ray tests, out[1] = exact point-triangle distances, out[2] = inside voxels,
// out[3] = needed voxels -- totals of ONE sdf_prep_kernel + sdf_dist_kernel launch pair.
extern "C" int ihmr_opt_sdf_stats(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                  const ihmr_opt_weights* w, unsigned long long* out4, void* stream) {
    if (!m || !io || !w || !out4 || B <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    OptWork wk = opt_carve(io->workspace, B);
    SdfWorkspace ws = sdf_carve(wk.sdf_ws, 2 * B, true);
    HIP_TRY(hipMemsetAsync(ws.stats, 0, 128, st));
    ihmr_kernel_timer* keep = g_timer;
    g_timer = nullptr;
    g_collect_stats = 1;
    int rc = opt_forward(m, m_left, io, wk, B, *w, kNoStep, st);
    g_collect_stats = 0;
    g_timer = keep;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(out4, ws.stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// diagnostics of the fused loop: enable = 1 zeroes the SDF work counters and switches them on for every launch recorded or issued
// afterwards (a stage graph captured while they are on keeps counting); enable = 0 synchronises, copies the sixteen counter slots out
// (ray tests, exact distances, inside voxels, needed voxels, bounding-sphere tests, voxels answered from candidate lists, voxels
// of such hands handed to the full search, voxels whose lists were rebuilt, plane + circle tests; the rest unused) and switches them off.
extern "C" int ihmr_opt_sdf_counters(const ihmr_opt_io* io, int B, unsigned long long* out16, int enable) {
    if (!io || B <= 0) return -1;
    SdfWorkspace ws = sdf_carve(opt_carve(io->workspace, B).sdf_ws, 2 * B, true);
    HIP_TRY(hipDeviceSynchronize());
    if (enable) {
        HIP_TRY(hipMemset(ws.stats, 0, 128));
        g_collect_stats = 1;
        return 0;
    }
    g_collect_stats = 0;
    if (!out16) return -1;
    HIP_TRY(hipMemcpy(out16, ws.stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ihmr_opt_sdf_inside_bits(const ihmr_opt_io* io, int B, unsigned* out, float* box) {
    if (!io || B <= 0 || 2 * B > SDF_MAX_HANDS || (!out && !box)) return -1;
    SdfWorkspace ws = sdf_carve(opt_carve(io->workspace, B).sdf_ws, 2 * B, true);
    HIP_TRY(hipDeviceSynchronize());
    if (out) HIP_TRY(hipMemcpy(out, ws.inside_bits, (size_t)2 * B * SDF_NCOL * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (box) HIP_TRY(hipMemcpy(box, ws.box, (size_t)2 * B * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------ encoder
extern "C" int ihmr_conv_igemm(const float* x, const float* w, const float* bias, const float* residual, float* y, int N, int H,
                               int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int ldx, int ldw,
                               int ldy, int ldr, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !w || !y || N <= 0 || Cout <= 0) return -1;
    ConvArgs a{x, w, bias, residual, y, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, ldx, ldw, ldy, ldr, act, (float*)workspace, 1};
    const int M = N * Ho * Wo, nk = (kh * kw * Cin + CONV_BK - 1) / CONV_BK;
    hipStream_t st = (hipStream_t)stream;
    // Tile and K split, from per-layer measurements on MI355X (scripts/prof_encoder.py with IHMR_CONV_FORCE):
    // the 128 x 128 tile wins on every ResNet-50 layer, even when it leaves CUs without a workgroup -- smaller
    // tiles move twice the operands through LDS per MFMA.  Occupancy is repaired with split-K instead: layers with
    // fewer than 1.5 workgroups per CU and a long K loop run their K halves in separate workgroups (two resident
    // workgroups per CU also hide each other's barriers); partial sums go to the caller's workspace and
    // conv_splitk_reduce_kernel adds them in fixed order.  Single-image-row layers (the Linear layers at batch 64)
    // take the 64-row tile and the deepest split the K loop allows.
    const bool wide_ok = Cout > 64 && ldw % 128 == 0;
    if (!wide_ok && ldw % 64 != 0) return -1;
    const int tiles[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
    auto blocks = [&](int t) { return (long)((M + tiles[t][0] - 1) / tiles[t][0]) * ((Cout + tiles[t][1] - 1) / tiles[t][1]); };
    auto usable = [&](int t) { return tiles[t][1] == 64 || wide_ok; };
    const long cap = workspace ? (long)(workspace_bytes / ((size_t)M * Cout * sizeof(float))) : 1;
    int pick = wide_ok ? 0 : 2, ksplit = 1;
    // Linear layers (a handful of workgroups): every K step costs a global-load round trip (~1 us) that nothing hides at this
    // occupancy, so the K loop is cut as deep as 4 steps per workgroup allow (measured on the IHMR-MLP training step:
    // 8-way 0.47 ms, 16-way 0.41 ms, 32-way 0.39 ms per step)
    if (M <= 64) {
        pick = wide_ok ? 1 : 3;
        ksplit = (int)std::max<long>(1, std::min<long>(std::min<long>(32, cap), nk / 4));
    } else if (blocks(pick) < 64) {
        ksplit = (int)std::max<long>(1, std::min<long>(std::min<long>(32, cap), nk / 4));
    } else if (blocks(pick) < 384 && nk >= 64 && cap >= 2) {
        ksplit = 2;
    } else if (wide_ok && blocks(0) > 768 && blocks(0) < 896) {
        // 784 tiles (the 14 x 14 layers with >= 1024 output channels): three resident workgroups per CU take 768, the last 16 run alone at ~2 us
        // per K step (8-33 us: scripts/experiments/conv_tail_generation.py); as 1568 half-height tiles (six resident per CU) the stragglers
        // are half as long.  Measured per layer (scripts/experiments/tile_sweep.sh): 145 -> 132 us (32 K steps), 82 -> 77.5 us (16 K steps)
        pick = 1;
    }
#ifdef IHMR_TUNING_BUILD   // per-layer tile / split measurements (scripts/prof_encoder.py builds its own library with this macro)
    if (const char* force = getenv("IHMR_CONV_FORCE")) {   // "<tile 0-3> <ksplit>"
        int ft = -1, fk = 1;
        if (sscanf(force, "%d %d", &ft, &fk) >= 1 && ft >= 0 && ft < 4 && usable(ft)) {
            pick = ft;
            ksplit = (int)std::max<long>(1, std::min<long>(std::min<long>(fk, cap), std::max(1, nk / 4)));
        }
    }
#endif
    a.ksplit = ksplit;
    const bool fast = (Cin % CONV_BK) == 0 && (ldx % 4) == 0 && Cin <= 2048;   // (2048: the zero page the padding pixels are read from, csrc/encoder.h)
    // Stream-K (csrc/encoder.h): layers with a long K loop and at most three 128 x 128 tiles per CU -- at batch 64 every 3 x 3 layer and the
    // first 1 x 1 of every bottleneck from 28 x 28 down (100, 196 or 392 tiles: 0.4-1.5 per CU) -- are shared evenly by two workers per CU.
    // Measured per layer (scripts/prof_encoder_layers.sh, round 4): 3 x 3 layers 175-187 -> 144-158 us, 1 x 1 layers with K >= 1024
    // 90-155 -> 77-141 us; with 32 K steps the fix-up's traffic eats the gain (85 -> 88 us), so those keep one workgroup per tile.
    // Workers: two per CU of THIS device, a multiple of 8 (conv_streamk_kernel numbers them by XCD), at most 512 (the workspace
    // contract of include/ihmr_hip.h: two 64 KB tile slots per worker = 64 MiB)
    int sk_cus = 0;
    if (int rc = device_cu_count(&sk_cus)) return rc;
    int sk_workers = std::max(8, std::min(512, 2 * sk_cus / 8 * 8)), sk_max_tiles = 768, sk_min_nk = 64;
#ifdef IHMR_TUNING_BUILD
    if (const char* f = getenv("IHMR_CONV_SK")) {   // "<max tiles> <min K steps> <workers>"
        sscanf(f, "%d %d %d", &sk_max_tiles, &sk_min_nk, &sk_workers);
        if (sk_workers < 8 || sk_workers % 8 != 0) return -1;
    }
#endif
    const long sk_tiles = blocks(0);
    if (fast && pick == 0 && M > 64 && Cout % 128 == 0 && ldy % 4 == 0 && ((uintptr_t)y % 16) == 0 && sk_tiles >= 64 && sk_tiles <= sk_max_tiles && nk >= sk_min_nk &&
        sk_tiles * nk >= 4L * sk_workers && workspace && workspace_bytes >= (size_t)sk_workers * 2 * 128 * 128 * sizeof(float)) {
        const int tiles_m = (M + 127) / 128, total = (int)(sk_tiles * nk);
        hipLaunchKernelGGL(conv_streamk_kernel, dim3(sk_workers), dim3(512), 0, st, a, tiles_m, nk, total);
        hipLaunchKernelGGL(conv_streamk_fixup_kernel, dim3((unsigned)sk_tiles, 8), dim3(256), 0, st, a, tiles_m, nk, total, sk_workers);
        return (int)hipGetLastError();
    }
    // (a persistent 1-D grid walking the tiles with a stride -- the cure for sdf_dist_kernel's slow slot refill -- was measured here too:
    // 6.95 -> 7.17 ms per 64-image pass at 6, 5 and 4 waves per SIMD alike; one workgroup per tile stays)
    const dim3 grid((M + tiles[pick][0] - 1) / tiles[pick][0], (Cout + tiles[pick][1] - 1) / tiles[pick][1], ksplit);
    if (fast) {
        switch (pick) {
            case 0: hipLaunchKernelGGL((conv_igemm_kernel<128, 128, CONV_FAST>), grid, dim3(512), 0, st, a); break;
            case 1: hipLaunchKernelGGL((conv_igemm_kernel<64, 128, CONV_FAST>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((conv_igemm_kernel<128, 64, CONV_FAST>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((conv_igemm_kernel<64, 64, CONV_FAST>), grid, dim3(128), 0, st, a); break;
        }
    } else if (Cin == 4 && (ldx % 4) == 0 && kw >= 4 && !wide_ok) {      // the stem on the image padded to 4 channels
        if (pick == 2) hipLaunchKernelGGL((conv_igemm_kernel<128, 64, CONV_C4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<64, 64, CONV_C4>), grid, dim3(128), 0, st, a);
    } else {
        switch (pick) {
            case 0: hipLaunchKernelGGL((conv_igemm_kernel<128, 128, CONV_GENERIC>), grid, dim3(512), 0, st, a); break;
            case 1: hipLaunchKernelGGL((conv_igemm_kernel<64, 128, CONV_GENERIC>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((conv_igemm_kernel<128, 64, CONV_GENERIC>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((conv_igemm_kernel<64, 64, CONV_GENERIC>), grid, dim3(128), 0, st, a); break;
        }
    }
    if (ksplit > 1) {
        if (Cout % 4 == 0) {
            const long total = (long)M * (Cout / 4);
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
        } else {
            const long total = (long)M * Cout;
            hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
        }
    }
    return (int)hipGetLastError();
}

extern "C" int ihmr_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (C % 4) return -1;
    const long total = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
    return (int)hipGetLastError();
}

extern "C" int ihmr_avgpool_relu(const float* x, float* y, int N, int HW, int C, int ldy, void* stream) {
    hipLaunchKernelGGL(avgpool_relu_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, N, HW, C, ldy);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ misc
// ------------------------------------------------------------------------------------------ evaluator
extern "C" int ihmr_eval_metrics(const float* pred_joints_3d, const float* gt_joints_3d, const float* coll_origin_scale,
                                 const float* sample_scale, const unsigned char* interacting, int B, double* out6, void* stream) {
    if (!pred_joints_3d || !gt_joints_3d || !coll_origin_scale || !out6 || B <= 0) return -1;
    hipLaunchKernelGGL(eval_metrics_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pred_joints_3d, gt_joints_3d, coll_origin_scale,
                       sample_scale, interacting, B, out6);
    return (int)hipGetLastError();
}

extern "C" int ihmr_eval_mpvpe(const float* pred_right, const float* pred_left, const float* gt_right, const float* gt_left,
                               const float* root_weights, const float* mano_params_weight, const float* sample_scale, int B,
                               double* out4, void* stream) {
    if (!pred_right || !pred_left || !gt_right || !gt_left || !root_weights || !mano_params_weight || !out4 || B <= 0) return -1;
    hipLaunchKernelGGL(eval_mpvpe_kernel, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, pred_right, pred_left, gt_right, gt_left,
                       root_weights, mano_params_weight, sample_scale, B, out4);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ IHMR-MLP training step
extern "C" int ihmr_mlp_train_grad(const ihmr_mano* m, const ihmr_mano* m_left, const ihmr_opt_io* io, int B,
                                   const ihmr_opt_weights* w, const ihmr_train_weights* tw, const float* gt_pose,
                                   const float* gt_shape, const float* params_weight, const float* init_shape,
                                   const float* trans_weight_mean, float* grad122, float* terms5, const int32_t* out_cols,
                                   int n_out, float* d_out, int ld_out, void* stream) {
    if (!m || !io || !w || !tw || !gt_pose || !gt_shape || !params_weight || !init_shape || !grad122 || !terms5 || B <= 0 ||
        (d_out && (!out_cols || n_out <= 0 || ld_out < n_out)))
        return -1;
    hipStream_t st = (hipStream_t)stream;
    OptWork wk = opt_carve(io->workspace, B);
    int rc = opt_forward(m, m_left, io, wk, B, *w, kNoStep, st);
    if (rc) return rc;
    lbs_backward_launch(m, true, 2 * B, B, wk.g_verts, wk.g_joints, wk.g_orient, wk.g_pose, wk.g_shape, wk.g_trans, 15, wk.lbs, st);
    hipLaunchKernelGGL(mlp_train_grad_kernel, dim3(B), dim3(128), 0, st, *io, wk, B, *tw, gt_pose, gt_shape, params_weight, init_shape,
                       trans_weight_mean, grad122, terms5, out_cols, n_out, d_out, ld_out);
    return (int)hipGetLastError();
}

extern "C" int ihmr_transpose(const float* x, float* y, int rows, int cols, int ldx, int ldy, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < rows) return -1;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, ldx,
                       ldy);
    return (int)hipGetLastError();
}

extern "C" int ihmr_relu_backward(float* dx, const float* y, int rows, int cols, int ld_dx, int ld_y, void* stream) {
    if (!dx || !y || rows <= 0 || cols <= 0) return -1;
    const long total = (long)rows * cols;
    if (ld_dx == cols && ld_y == cols && total % 4 == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)y % 16) == 0)
        hipLaunchKernelGGL(relu_backward4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)dx,
                           (const float4*)y, total / 4);
    else
        hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, y, rows, cols, ld_dx,
                           ld_y);
    return (int)hipGetLastError();
}

extern "C" int ihmr_colsum(const float* x, float* out, int rows, int cols, int ldx, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return -1;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, out, rows, cols, ldx);
    return (int)hipGetLastError();
}

extern "C" int ihmr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float grad_scale,
                              float lr, float beta1, float beta2, float eps, int step, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n == 0 || step <= 0) return -1;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, grad_scale, beta1, beta2, eps, (float)((double)lr / bc1), (float)sqrt(bc2));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ encoder training kernels
#define BN_MAX_CHUNKS 1024
static int bn_chunks(long M, int* rows_per) {
    long rp = std::max<long>(16, (M + BN_MAX_CHUNKS - 1) / BN_MAX_CHUNKS);
    *rows_per = (int)rp;
    return (int)((M + rp - 1) / rp);
}
static dim3 bn_grid(int C, int S) { return dim3((unsigned)((C / 4 + 255) / 256), (unsigned)S); }

extern "C" size_t ihmr_bn_workspace_bytes(int C) { return (size_t)BN_MAX_CHUNKS * 2 * C * sizeof(float) + (size_t)2 * C * sizeof(float); }

extern "C" int ihmr_bn_train_forward(const float* z, long M, int C, const float* gamma, const float* beta, const float* residual,
                                     int relu, float eps, float* y, float* mean, float* var, float* invstd, float* running_mean,
                                     float* running_var, float momentum, void* workspace, void* stream) {
    if (!z || !gamma || !beta || !y || !mean || !var || !invstd || !workspace || M <= 0 || C <= 0 || C % 4) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rows_per;
    const int S = bn_chunks(M, &rows_per);
    float* part = (float*)workspace;
    const dim3 grid = bn_grid(C, S);
    // one pass over z: sum z and sum (z - z0)^2 with row 0 as the pivot, then mean / variance / invstd
    hipLaunchKernelGGL(bn_partial_kernel<3>, grid, dim3(256), 0, st, z, (const float*)nullptr, (int)M, C, C, C, rows_per,
                       (const float*)nullptr, (const float*)nullptr, part, (const float*)nullptr);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)part, z, S, C, M, mean, var, invstd, eps, running_mean,
                       running_var, momentum);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((M * (C / 4) + 255) / 256)), dim3(256), 0, st, z, (const float*)mean,
                       (const float*)invstd, gamma, beta, residual, y, M, C, relu);
    return (int)hipGetLastError();
}

extern "C" int ihmr_bn_train_backward(const float* z, const float* g, long M, int C, const float* mean, const float* invstd,
                                      const float* gamma, const float* relu_y, float* dz, float* dgamma, float* dbeta, void* workspace,
                                      void* stream) {
    if (!z || !g || !mean || !invstd || !gamma || !dz || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || C % 4) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rows_per;
    const int S = bn_chunks(M, &rows_per);
    float* part = (float*)workspace;
    // sum g -> dbeta, sum g * xhat -> dgamma (written by the finish kernel itself, read back by the apply kernel)
    hipLaunchKernelGGL(bn_partial_kernel<2>, bn_grid(C, S), dim3(256), 0, st, z, g, (int)M, C, C, C, rows_per, mean, invstd, part, relu_y);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, st, (const float*)part, S, 2, C, 1.0, dbeta,
                       (float*)nullptr, 0.f, dgamma);
    hipLaunchKernelGGL(bn_backward_apply_kernel, dim3((unsigned)((M * (C / 4) + 255) / 256)), dim3(256), 0, st, z, g, mean, invstd, gamma,
                       (const float*)dbeta, (const float*)dgamma, dz, M, C, relu_y);
    return (int)hipGetLastError();
}

extern "C" int ihmr_conv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                               int kh, int kw, int stride, int pad, int ldx, int lddy, int ldw, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace || N <= 0 || Cin % 4 || lddy % 4 || ldx % 4 || ldw < Cout) return -1;
    hipStream_t st = (hipStream_t)stream;
    // (conv_wgrad_kernel splits a pixel index by a float-reciprocal product with a +-1 correction: exact below 2^23 pixels)
    if ((long)N * Ho * Wo >= (1L << 23)) return -1;
    const int M = N * Ho * Wo, K = kh * kw * Cin;
    const bool wide_n = Cout > 64, wide_m = K > 64;
    const int BMv = wide_m ? 128 : 64, BNv = wide_n ? 128 : 64;
    const long tiles = (long)((K + BMv - 1) / BMv) * ((Cout + BNv - 1) / BNv);
    const int nchunks = (M + CONV_BK - 1) / CONV_BK;
    const long cap = (long)(workspace_bytes / ((size_t)K * Cout * sizeof(float)));
    if (cap < 1) return -1;
    long msplit = std::max<long>(1, std::min<long>(std::min<long>(cap, 256), std::min<long>((1024 + tiles - 1) / tiles, std::max(1, nchunks / 8))));
    const int chunks_per = (int)((nchunks + msplit - 1) / msplit);
    msplit = (nchunks + chunks_per - 1) / chunks_per;
    WgradArgs a{x, dy, (float*)workspace, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, ldx, lddy, chunks_per};
    const dim3 grid((K + BMv - 1) / BMv, (Cout + BNv - 1) / BNv, (unsigned)msplit);
    if (wide_m && wide_n) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, dim3(512), 0, st, a);
    else if (wide_m) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, dim3(256), 0, st, a);
    else if (wide_n) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, dim3(128), 0, st, a);
    // fixed-order sum of the pixel-range partials into dw [K][ldw]
    ConvArgs r{nullptr, nullptr, nullptr, nullptr, dw, K, 1, 1, 0, 1, 1, Cout, 1, 1, 1, 0, 0, 0, ldw, 0, 0, (float*)workspace, (int)msplit};
    if (Cout % 4 == 0 && ldw % 4 == 0 && msplit >= 32)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((long)K * (Cout / 4) + 15) / 16)), dim3(256), 0, st, (const float*)workspace, dw, K,
                           Cout, ldw, (int)msplit);
    else if (Cout % 4 == 0) hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3((unsigned)(((long)K * (Cout / 4) + 255) / 256)), dim3(256), 0, st, r);
    else hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3((unsigned)(((long)K * Cout + 255) / 256)), dim3(256), 0, st, r);
    return (int)hipGetLastError();
}

extern "C" int ihmr_pack_dgrad_weight(const float* w, float* out, int kh, int kw, int Cin, int Cout, int ldw, int ldo, void* stream) {
    if (!w || !out || ldw < Cout || ldo < Cin) return -1;
    const long total = (long)kh * kw * Cout * Cin;
    hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, kh, kw, Cin,
                       Cout, ldw, ldo);
    return (int)hipGetLastError();
}

extern "C" int ihmr_interleave2(const float* p00, const float* p01, const float* p10, const float* p11, float* dx, int N, int Ho, int Wo,
                                int C, void* stream) {
    if (!p00 || !p01 || !p10 || !p11 || !dx || C % 4) return -1;
    const long total = (long)N * 2 * Ho * 2 * Wo * (C / 4);
    hipLaunchKernelGGL(interleave2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p00, p01, p10, p11, dx, N,
                       Ho, Wo, C);
    return (int)hipGetLastError();
}

extern "C" int ihmr_dilate2(const float* dy, float* out, int N, int Ho, int Wo, int C, void* stream) {
    if (!dy || !out || C % 4) return -1;
    const long total = (long)N * 2 * Ho * 2 * Wo * (C / 4);
    hipLaunchKernelGGL(dilate2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, out, N, Ho, Wo, C);
    return (int)hipGetLastError();
}

extern "C" int ihmr_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                                          void* stream) {
    if (!x || !dy || !dx || C % 4) return -1;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H,
                       W, C, Ho, Wo);
    return (int)hipGetLastError();
}

extern "C" int ihmr_avgpool_relu_backward(const float* y, const float* dy, float* dx, int N, int HW, int C, int ldy, void* stream) {
    if (!y || !dy || !dx) return -1;
    const long total = (long)N * HW * C;
    hipLaunchKernelGGL(avgpool_relu_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, N, HW,
                       C, ldy);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ image preprocessing
extern "C" int ihmr_preprocess_images(const uint8_t* pixels, const int64_t* offsets, const int32_t* sizes, const uint8_t* do_flip,
                                      int B, int final_size, float* img_out, uint8_t* img_u8, const float* joints_in,
                                      float* joints_out, void* stream) {
    if (!pixels || !offsets || !sizes || !img_out || B <= 0 || final_size <= 0 || final_size % PRE_PPT || (joints_in && !joints_out))
        return -1;
    hipLaunchKernelGGL(preprocess_kernel, dim3((final_size * final_size / PRE_PPT + PRE_THREADS - 1) / PRE_THREADS, B), dim3(PRE_THREADS), 0,
                       (hipStream_t)stream, pixels, offsets, sizes, do_flip, final_size, img_out, img_u8, joints_in, joints_out);
    return (int)hipGetLastError();
}

extern "C" int ihmr_copy_segments(const ihmr_copy_seg* segs, int n, void* stream) {
    if (!segs || n <= 0 || n > IHMR_COPY_MAX_SEGS) return -1;
    CopySegTable t;
    memset(&t, 0, sizeof(t));
    long biggest = 0;
    for (int i = 0; i < n; ++i) {
        const ihmr_copy_seg& g = segs[i];
        if (!g.src || !g.dst || g.rows <= 0 || g.width <= 0 || g.src_ld < g.width || g.dst_ld < g.width) return -1;
        if (((uintptr_t)g.src | (uintptr_t)g.dst) & 3u) return -1;
        t.s[i] = g;
        biggest = std::max(biggest, (long)g.rows * g.width);
    }
    const unsigned bx = (unsigned)std::max<long>(1, std::min<long>(64, (biggest + 1023) / 1024));
    hipLaunchKernelGGL(copy_segments_kernel, dim3(bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

extern "C" int ihmr_root_align_joints(const float* joints4, float* out4, int B, void* stream) {
    if (!joints4 || !out4 || B <= 0) return -1;
    hipLaunchKernelGGL(root_align_joints_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, joints4, out4, B);
    return (int)hipGetLastError();
}

extern "C" int ihmr_set_kernel_timer(ihmr_kernel_timer* t) {
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    g_timer = t;
    return 0;
}

extern "C" int ihmr_flush_kernel_timer(void) {
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    for (auto& p : g_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(p.b));
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        if (g_timer) {
            if (p.k < 0) { g_timer->ms_event_pair += ms; g_timer->n_event_pair += 1; }
            else if (p.k < IHMR_TIMED_KERNELS) { g_timer->ms[p.k] += ms; g_timer->n[p.k] += 1; }
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_pending.clear();
    return 0;
}


#ifdef SDF_STAMPS
// experiment builds only (scripts/sdf_stamps.py): zero = 1 clears the per-wave phase sums, zero = 0 copies them to the host (4096 * 4 * 8 int64)
extern "C" int ihmr_debug_stamps(long long* host, int zero) {
    HIP_TRY(hipDeviceSynchronize());
    if (zero) {
        void* p; HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_sdf_stamps))); HIP_TRY(hipMemset(p, 0, sizeof(long long) * 4096 * 4 * 8));
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_sdf_prep))); HIP_TRY(hipMemset(p, 0, sizeof(long long) * 4096 * 8));
        return 0;
    }
    HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sdf_stamps), sizeof(long long) * 4096 * 4 * 8));
    HIP_TRY(hipMemcpyFromSymbol(host + 4096 * 4 * 8, HIP_SYMBOL(g_sdf_span), sizeof(long long) * 4096 * 4 * 4));
    HIP_TRY(hipMemcpyFromSymbol(host + 4096 * 4 * 12, HIP_SYMBOL(g_sdf_prep), sizeof(long long) * 4096 * 8));
    return 0;
}
#endif

#ifdef SDF_QMASK_CHECK
// device pointers of the collision workspace of a fused-loop io: qcell, inside_bits, box, phi
extern "C" int ihmr_debug_sdf_ptrs(const ihmr_opt_io* io, int B, void** out4) {
    SdfWorkspace ws = sdf_carve(opt_carve(io->workspace, B).sdf_ws, 2 * B, true);
    out4[0] = ws.qcell; out4[1] = ws.inside_bits; out4[2] = ws.box; out4[3] = ws.phi;
    return 0;
}
extern "C" int ihmr_debug_qmask(unsigned* host8) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_qmask_bad), 32);
}
#endif

#ifdef SDF_HANDLOG
// experiment builds only: cap > 0 -- allocate a log of `cap` records and start; cap == 0 -- copy the records out ([n][4] uint32: hand,
// voxels with a candidate list, voxels for the full search, bit 0 lists reused | bit 1 static hand), n = return value
extern "C" long ihmr_debug_handlog(unsigned* host, long cap) {
    static uint4* buf = nullptr;
    static unsigned cap_now = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (cap > 0) {
        if (buf) (void)hipFree(buf);
        if (hipMalloc(&buf, (size_t)cap * 16) != hipSuccess) return -1;
        cap_now = (unsigned)cap;
        const unsigned zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_handlog_n), &zero, 4) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_handlog_cap), &cap_now, 4) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_handlog), &buf, 8) != hipSuccess) return -1;
        return 0;
    }
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_handlog_n), 4) != hipSuccess) return -1;
    if (n > cap_now) n = cap_now;
    if (n && hipMemcpy(host, buf, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    uint4* null = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_handlog), &null, 8);
    return (long)n;
}
#endif

#ifdef CONV_STAMPS
// experiment builds only (scripts/experiments/conv_stamps.py): zero = 1 clears, zero = 0 copies the 1024 x 10 phase sums of conv_streamk_kernel out
extern "C" int ihmr_debug_conv_stamps(long long* host, int zero) {
    HIP_TRY(hipDeviceSynchronize());
    void* p;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_stamps)));
    if (zero) { HIP_TRY(hipMemset(p, 0, sizeof(long long) * 1024 * 10)); return 0; }
    HIP_TRY(hipMemcpy(host, p, sizeof(long long) * 1024 * 10, hipMemcpyDeviceToHost));
    return 0;
}
#endif

#ifdef TAIL_STAMPS
// experiment builds only (scripts/tail_stamps.py): zero = 1 clears, zero = 0 copies the 3 x 4096 x 8 phase sums of opt_tail_kernel out
extern "C" int ihmr_debug_tail_stamps(long long* host, int zero) {
    HIP_TRY(hipDeviceSynchronize());
    void *p, *q;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_tail_stamps)));
    HIP_TRY(hipGetSymbolAddress(&q, HIP_SYMBOL(g_samp_stamps)));
    if (zero) { HIP_TRY(hipMemset(p, 0, sizeof(long long) * 3 * 4096 * 8)); HIP_TRY(hipMemset(q, 0, sizeof(long long) * 4096 * 8)); return 0; }
    HIP_TRY(hipMemcpy(host, p, sizeof(long long) * 3 * 4096 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(host + 3 * 4096 * 8, q, sizeof(long long) * 4096 * 8, hipMemcpyDeviceToHost));      // the sampler's own phases
    return 0;
}
#endif

#ifdef IHMR_TIMELINE
// experiment builds only (scripts/timeline_wg.py): cap > 0 -- allocate a ring of `cap` workgroup records (256 segments) and start
// recording; cap == 0 -- copy the records out (host: [n][3] uint64, n = return value) and stop
extern "C" long ihmr_debug_timeline(unsigned long long* host, long cap) {
    static unsigned long long* ring = nullptr;
    static unsigned seg_cap = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    void* cur = nullptr;
    if (hipGetSymbolAddress(&cur, HIP_SYMBOL(g_tl_cursor)) != hipSuccess) return -1;
    if (cap > 0) {
        if (ring) (void)hipFree(ring);
        seg_cap = (unsigned)(cap / 256);
        if (hipMalloc(&ring, (size_t)seg_cap * 256 * 24) != hipSuccess) return -1;
        if (hipMemset(cur, 0, 256 * 32 * 4) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_cap), &seg_cap, 4) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_ring), &ring, 8) != hipSuccess) return -1;
        return 0;
    }
    static unsigned counts[256 * 32];
    if (hipMemcpy(counts, cur, sizeof(counts), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long n = 0;
    for (int s = 0; s < 256; ++s) {
        const unsigned c = counts[s * 32] < seg_cap ? counts[s * 32] : seg_cap;
        if (c && hipMemcpy(host + 3 * n, ring + 3 * (size_t)s * seg_cap, (size_t)c * 24, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        n += c;
    }
    unsigned long long* null = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_ring), &null, 8);
    return n;
}
#endif

extern "C" const char* ihmr_version(void) { return "ihmr_hip 0.1 (gfx950)"; }
