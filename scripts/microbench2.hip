#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ void stream_kernel(const float4* __restrict__ src, float* out, int rows, int rowlen) {
    const int v = (blockIdx.x % 4) * 195 + threadIdx.x;
    float a0 = 0, a1 = 0, a2 = 0;
    if (threadIdx.x < 195) {
#pragma unroll U
        for (int e = 0; e < rows; ++e) {
            const float4 p = src[e * rowlen + v];
            a0 += p.x; a1 += p.y; a2 += p.z;
        }
    }
    if (a0 + a1 + a2 == 1234.5f) out[0] = a0;
}
__global__ void producer(float* buf, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = (float)((i * 7919) % n);
}
__global__ void consumer_chain(const float* buf, float* out, int n, int hops) {
    int p = (blockIdx.x * 977 + threadIdx.x * 131) % n;
    for (int h = 0; h < hops; ++h) p = (int)buf[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)p;
}
int main() {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    const int rows = 135, rowlen = 832;
    float4* src; CK(hipMalloc(&src, rows * rowlen * 16)); CK(hipMemset(src, 0, rows * rowlen * 16));
    float* out; CK(hipMalloc(&out, 1 << 20));
    float* junk; CK(hipMalloc(&junk, 64 << 20));
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {64, 256}) {
#define RUN(U) { hipLaunchKernelGGL(producer, dim3(1024), dim3(256), 0, 0, junk, 16 << 20); /* thrash L2 */ \
            CK(hipEventRecord(a)); hipLaunchKernelGGL(stream_kernel<U>, dim3(blocks), dim3(256), 0, 0, src, out, rows, rowlen); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
            CK(hipEventElapsedTime(&ms, a, b)); printf("stream blocks=%d unroll=%d after-thrash: %.1f us\n", blocks, U, ms * 1e3); \
            CK(hipEventRecord(a)); hipLaunchKernelGGL(stream_kernel<U>, dim3(blocks), dim3(256), 0, 0, src, out, rows, rowlen); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
            CK(hipEventElapsedTime(&ms, a, b)); printf("stream blocks=%d unroll=%d warm:         %.1f us\n", blocks, U, ms * 1e3); }
            RUN(1) RUN(9) RUN(27)
        }
    }
    // producer -> consumer dependent chain
    const int n = 100000;
    float* buf; CK(hipMalloc(&buf, n * 4));
    for (int hops : {1, 10, 40}) {
        hipLaunchKernelGGL(producer, dim3(128), dim3(256), 0, 0, buf, n);
        CK(hipEventRecord(a)); hipLaunchKernelGGL(consumer_chain, dim3(128), dim3(256), 0, 0, buf, out, n, hops); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b)); printf("consumer after producer hops=%d: %.1f us\n", hops, ms * 1e3);
        CK(hipEventRecord(a)); hipLaunchKernelGGL(consumer_chain, dim3(128), dim3(256), 0, 0, buf, out, n, hops); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b)); printf("consumer again           hops=%d: %.1f us\n", hops, ms * 1e3);
    }
    return 0;
}
