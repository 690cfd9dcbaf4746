// Sustained fp32 MFMA rate and shader clock: the bare v_mfma_f32_32x32x2_f32 loop of scripts/mfma_peak.hip (4 waves per SIMD, 2 independent
// accumulators per wave) run for ~1 ms, ~10 ms and ~100 ms; the clock is the shader-clock counter (s_memtime) against the 100 MHz wall clock
// (s_memrealtime), read by one thread per workgroup at the start and the end of the loop.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const long long c0 = (long long)__builtin_readcyclecounter(), w0 = (long long)wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = (long long)__builtin_readcyclecounter(), w1 = (long long)wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int blocks = 1024, threads = 256;
    float* out; long long* clk; hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&clk, (size_t)blocks * 16);
    long long* h = (long long*)malloc((size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, clk, 10); hipDeviceSynchronize();
    for (int iters : {2000, 20000, 200000, 2000}) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
        const double flops = (double)blocks * (threads / 64) * iters * 8.0 * 2 * 4096.0;
        printf("iters %6d: %8.3f ms  %6.1f TFLOP/s  shader clock %.3f GHz  (%.1f TFLOP/s per GHz)\n", iters, ms, flops / ms / 1e9, c / (w * 10.0), flops / ms / 1e9 / (c / (w * 10.0)));
    }
    return 0;
}
