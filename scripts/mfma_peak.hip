// f32 MFMA issue-rate microbenchmark: NACC independent accumulators per wave, W waves per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int threads, const char* name) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    printf("%s blocks=%d threads=%d nacc=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, NACC, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 256, "1 wave/SIMD"); run<2>(256, 256, "1 wave/SIMD"); run<4>(256, 256, "1 wave/SIMD");
    run<2>(512, 256, "2 waves/SIMD"); run<2>(1024, 256, "4 waves/SIMD"); run<2>(256, 512, "2 waves/SIMD (512 thr)");
    run<4>(1024, 256, "4 waves/SIMD");
    return 0;
}
