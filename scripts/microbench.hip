// Calibration micro-benchmarks: effective clock under short bursts, dependent-load latency, launch gap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fma_kernel(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c);
        a = __builtin_fmaf(a, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, a);
        a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c);
    }
    if (a + d + e == 12345.f) out[0] = a;
}
__global__ void cycles_kernel(long long* out, int iters) {
    long long t0 = clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, c);
    long long t1 = clock64();
    long long w0 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)a; out[2] = w0; }
}
__global__ void chase_kernel(const int* next, int* out, int hops) {
    int p = threadIdx.x;
    for (int i = 0; i < hops; ++i) p = next[p];
    out[threadIdx.x] = p;
}
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) p[0] = 1; }

int main() {
    float* out; CK(hipMalloc(&out, 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    // 1. FMA throughput: grid 1024 blocks x 256 thr, 8 fma per iter
    for (int rep = 0; rep < 3; ++rep) {
        for (int iters : {1000, 20000, 400000}) {
            CK(hipEventRecord(a)); hipLaunchKernelGGL(fma_kernel, dim3(2048), dim3(256), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            double fl = 2.0 * 8 * iters * 2048.0 * 256;
            printf("fma iters=%d  %.3f ms  %.1f TFLOP/s\n", iters, ms, fl / ms / 1e9);
        }
    }
    // single wave dependent fma chain: cycles per fma, and shader clock estimate
    long long* cyc; CK(hipMalloc(&cyc, 64));
    for (int iters : {10000, 1000000}) {
        CK(hipEventRecord(a)); hipLaunchKernelGGL(cycles_kernel, dim3(1), dim3(64), 0, 0, cyc, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        long long h[3]; CK(hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost));
        printf("dep-chain iters=%d: clock64 delta=%lld (%.2f per fma), wall %.3f ms => clock64 rate %.1f MHz\n", iters, h[0], (double)h[0] / iters, ms, h[0] / ms / 1e3);
    }
    // 2. pointer chase at several footprints (stride 64 ints = 256 B to defeat line reuse)
    for (size_t n : {(size_t)1 << 12, (size_t)1 << 18, (size_t)1 << 22, (size_t)1 << 26}) {
        std::vector<int> h(n);
        const size_t stride = 4099 * 64;  // co-prime walk
        for (size_t i = 0; i < n; ++i) h[i] = (int)((i + stride) % n);
        int* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        int* o; CK(hipMalloc(&o, 256));
        const int hops = 2000;
        hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, 0, d, o, hops); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, 0, d, o, hops); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("chase footprint %8.1f KB: %.1f ns/hop\n", n * 4 / 1024.0, ms * 1e6 / hops);
        CK(hipFree(d)); CK(hipFree(o));
    }
    // 3. launch gap: 200 empty kernels back to back
    CK(hipEventRecord(a));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, 0, (int*)nullptr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("200 empty kernels: %.3f ms => %.2f us per launch\n", ms, ms * 1e3 / 200);
    // graph of 200 empty kernels
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("graph of 200 empty kernels: %.3f ms => %.2f us per kernel\n", ms, ms * 1e3 / 200);
    return 0;
}
