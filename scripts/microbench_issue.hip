// Issue-rate calibration for bench.py's `roofline.*.issue_slots` (VALU wave-instructions x cycles per instruction / SIMD-cycles):
// how many shader cycles does ONE wave64 vector instruction of each class occupy its SIMD's issue port when the SIMD has enough
// independent work (4 waves per SIMD, 8 independent chains per wave)?  Classes = what the IHMR-OPT kernels are made of (counted in
// their ISA): fp32 fma / mul / add, packed fp32 fma (sdf_dist_kernel's sphere passes), integer / logic, compare + select, fp32
// division sequence pieces (v_rcp_f32, v_sqrt_f32: quarter rate), LDS reads.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_issue.hip -o scripts/microbench_issue && scripts/microbench_issue
// Prints cycles per instruction per SIMD for every class at 1 and 4 waves per SIMD (s_memtime runs at the shader clock, clock_ratio.hip)
// and the same from wall time at the measured clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
enum { K_FMA, K_PKFMA, K_INT, K_CMPSEL, K_RCP, K_SQRT, K_LDS, K_MUL, K_ADDU, K_LSHL, K_DIV, K_N };
static const char* kNames[K_N] = {"v_fma_f32", "v_pk_fma_f32", "v_and_or_b32 / v_add_u32 (int)", "v_cmp_gt_f32 + v_cndmask_b32 (2 instr)",
                                   "v_rcp_f32", "v_sqrt_f32", "ds_read_b32 (conflict-free)", "v_mul_f32", "v_add_u32 (2-operand int)",
                                   "v_lshlrev_b32", "a / b (IEEE fp32 division, per division)"};

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, long long* cyc, int iters) {
    __shared__ float lds[4096];
    float a[8], b = 1.0001f, c = 0.37f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[8];
    unsigned u[8];
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 1e-3f + k; p[k] = v2{a[k], a[k] + 1.f}; u[k] = threadIdx.x * 7u + k; }
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = k;
    __syncthreads();
    const unsigned la = (threadIdx.x % 64) * 4;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == K_FMA) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == K_MUL) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == K_PKFMA) {
                const v2 bb = {b, b}, cc = {c, c};
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(bb), "v"(cc));
                REP8(X)
#undef X
            } else if (KIND == K_INT) {
#define X(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(0x7fffffu), "v"(r + 1u));
                REP8(X)
#undef X
            } else if (KIND == K_CMPSEL) {
#define X(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(c), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == K_RCP) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (KIND == K_SQRT) {
#define X(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
                REP8(X)
#undef X
            } else if (KIND == K_ADDU) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(r + 3u));
                REP8(X)
#undef X
            } else if (KIND == K_LSHL) {
#define X(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[k]));
                REP8(X)
#undef X
            } else if (KIND == K_DIV) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = c / a[k];
            } else if (KIND == K_LDS) {
#define X(k) asm volatile("ds_read_b32 %0, %1 offset:" #k "*256" : "=v"(a[k]) : "v"(la));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y + (float)u[k];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x % 64 == 0) { cyc[(blockIdx.x * 16 + threadIdx.x / 64) * 2] = t0; cyc[(blockIdx.x * 16 + threadIdx.x / 64) * 2 + 1] = t1; }
}

template <int KIND>
static int run(float* out, long long* cyc, int cus) {
    const int iters = 20000;
    for (int wps : {1, 4}) {            // waves per SIMD: one block per CU of 4 or 16 waves
        const int threads = 256 * wps;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(cus), dim3(threads), 0, 0, out, cyc, 50);      // warm-up (clock ramp)
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(cus), dim3(threads), 0, 0, out, cyc, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        long long h[32]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));        // block 0: (start, end) of its 4 * wps waves
        const double instr_per_wave = 64.0 * iters * (KIND == K_CMPSEL ? 2 : 1);
        // the block's waves start together and share the CU's four SIMDs evenly: from the first start to the last end every SIMD has
        // issued wps waves' instructions (the arbiter favours the oldest wave: one wave's own lifetime says nothing under contention)
        long long lo = h[0], hi = h[1], own = 0;
        for (int w = 0; w < 4 * wps; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; own += h[2 * w + 1] - h[2 * w]; }
        printf("  %-42s %d wave(s)/SIMD: %6.2f shader cycles per instruction per SIMD (mean wave lifetime / its instructions: %6.2f); wall %.3f ms "
               "= %.2f cycles at 2.4 GHz\n", kNames[KIND], wps, (double)(hi - lo) / (instr_per_wave * wps), (double)own / (4 * wps) / instr_per_wave, ms,
               ms * 1e-3 * 2.4e9 / (instr_per_wave * wps));
    }
    return 0;
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1024)); CK(hipMalloc(&cyc, (size_t)cus * 16 * 16));
    printf("issue-rate calibration on %d CUs (cycles of the shader clock, s_memtime)\n", cus);
    if (run<K_FMA>(out, cyc, cus) || run<K_MUL>(out, cyc, cus) || run<K_PKFMA>(out, cyc, cus) || run<K_INT>(out, cyc, cus) || run<K_CMPSEL>(out, cyc, cus) ||
        run<K_RCP>(out, cyc, cus) || run<K_SQRT>(out, cyc, cus) || run<K_LDS>(out, cyc, cus) || run<K_ADDU>(out, cyc, cus) || run<K_LSHL>(out, cyc, cus) ||
        run<K_DIV>(out, cyc, cus)) return 1;
    return 0;
}
