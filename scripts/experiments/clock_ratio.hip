// calibrates s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz): hipcc --offload-arch=gfx950 -O2 clock_ratio.hip -o clock_ratio
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long* out, int spin) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
    long long* d; long long h[3];
    hipMalloc(&d, 24);
    for (int spin : {100000, 1000000}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, spin);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("spin %d: clock64 ticks %lld, wall_clock64 ticks %lld (100 MHz => %.1f us), ticks per us %.1f\n", spin, h[0], h[1], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
    return 0;
}
