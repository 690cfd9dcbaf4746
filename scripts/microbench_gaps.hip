// What does a dependent launch boundary cost for the grid shapes of ONE refinement iteration at batch 64?
//
// The single-batch latency (ms per refinement iteration, one batch of 64 in flight) is a chain of three one-generation
// kernels -- sdf_prep_kernel 128 x 1024 threads, sdf_dist_kernel 1024 x 256 (persistent grid), opt_tail_kernel 64 x 512 with ~80 KB of
// LDS -- replayed from a hipGraph; round 5 found 12.4 us per iteration in NO kernel (three boundaries of 4.1 us, against 1.45-1.9 us
// for a dependent boundary in MI355X_MICROARCH.md).  This program replays the same chain with TRIVIAL kernels of the same grid shapes,
// and adds the suspects one at a time:
//   v0  three 256 x 256 kernels, 16 bytes of arguments                      (the guide's "boundary" row)
//   v1  the real grid shapes (128 x 1024, 1024 x 256, 64 x 512)              -> dispatch ramp of the shapes
//   v2  + the real LDS footprints (40 KB / 40 KB / 80 KB)                     -> LDS allocation in the dispatcher
//   v3  + the real kernel-argument sizes (~0.4 / 0.4 / 1.0 KB by value)        -> kernarg fetch
//   v4  + every kernel leaves the bytes dirty its real counterpart does (5.6 / 0.6 / 1.2 MB) and the next one reads them
//                                                                              -> end-of-kernel write-back + cold first loads
//   v5  v4 with the persistent grid cut to 256 x 256
// per variant: wall time per iteration of a 200-iteration graph (median of 9 replays) minus the same graph with ONE kernel per
// iteration ... / 3 = microseconds per boundary.  Under `rocprofv3 --kernel-trace` the same run gives the trace's view of the gaps
// (scripts/gap_table.py), which calibrates what the trace adds.
//
// build: hipcc -O2 --offload-arch=gfx950 scripts/microbench_gaps.hip -o scripts/microbench_gaps
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int N> struct Blob { int v[N]; };

// reads `rd` floats per thread from `in` (coalesced, strided over the grid), writes `wr` floats per thread to `out`
template <int ARGW>
__global__ void link_kernel(const float* __restrict__ in, float* __restrict__ out, int rd, int wr, Blob<ARGW> blob) {
    extern __shared__ float lds[];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    float acc = (float)blob.v[0];
    for (int i = 0; i < rd; ++i) acc += in[(size_t)i * n + t];
    if (acc == 12345.678f) lds[threadIdx.x] = acc;      // (keeps the LDS allocation and the loads alive)
    for (int i = 0; i < wr; ++i) out[(size_t)i * n + t] = acc + (float)i;
}

struct Shape { int grid, block, lds; int rd, wr; };     // rd / wr: floats per thread

template <int A0, int A1, int A2>
static int run_variant(const char* name, const Shape* sh, float* buf[4], hipStream_t st, int iters, int reps) {
    auto build = [&](int kernels_per_iter, hipGraphExec_t* exec) -> int {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int it = 0; it < iters; ++it) {
            for (int k = 0; k < kernels_per_iter; ++k) {
                const Shape& s = sh[k];
                if (k == 0) hipLaunchKernelGGL(link_kernel<A0>, dim3(s.grid), dim3(s.block), s.lds, st, buf[2], buf[0], s.rd, s.wr, Blob<A0>{});
                if (k == 1) hipLaunchKernelGGL(link_kernel<A1>, dim3(s.grid), dim3(s.block), s.lds, st, buf[k - 1], buf[k], s.rd, s.wr, Blob<A1>{});
                if (k == 2) hipLaunchKernelGGL(link_kernel<A2>, dim3(s.grid), dim3(s.block), s.lds, st, buf[k - 1], buf[k], s.rd, s.wr, Blob<A2>{});
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        return 0;
    };
    auto time_graph = [&](hipGraphExec_t exec, double* us_per_iter) -> int {
        std::vector<double> t;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(exec, st));
            CK(hipStreamSynchronize(st));
            const auto t1 = std::chrono::steady_clock::now();
            if (r >= 2) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
        }
        std::sort(t.begin(), t.end());
        *us_per_iter = t[t.size() / 2];
        return 0;
    };
    hipGraphExec_t e3, e1;
    if (build(3, &e3) || build(1, &e1)) return 1;
    double us3 = 0, us1 = 0;
    if (time_graph(e3, &us3) || time_graph(e1, &us1)) return 1;
    // a chain of the FIRST kernel alone has one boundary per iteration too: us1 = body0 + boundary; us3 = body0 + body1 + body2 + 3 boundaries
    printf("  %-3s %8.2f us per 3-launch iteration, %6.2f us per 1-launch iteration (first kernel alone)\n", name, us3, us1);
    CK(hipGraphExecDestroy(e3)); CK(hipGraphExecDestroy(e1));
    return 0;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf[4];
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&buf[i], 64 << 20)); CK(hipMemsetAsync(buf[i], 0, 64 << 20, st)); }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&link_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&link_kernel<100>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&link_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10));
    const int iters = 200, reps = 9;
    printf("dependent-boundary cost by grid shape (graph replay of %d iterations, median of %d; wall time per iteration)\n", iters, reps);
    {   const Shape s[3] = {{256, 256, 0, 0, 0}, {256, 256, 0, 0, 0}, {256, 256, 0, 0, 0}};
        printf("v0: three 256 x 256 kernels, 16-byte arguments\n");
        if (run_variant<4, 4, 4>("v0", s, buf, st, iters, reps)) return 1; }
    {   const Shape s[3] = {{128, 1024, 0, 0, 0}, {1024, 256, 0, 0, 0}, {64, 512, 0, 0, 0}};
        printf("v1: the iteration's grid shapes (128 x 1024, 1024 x 256, 64 x 512)\n");
        if (run_variant<4, 4, 4>("v1", s, buf, st, iters, reps)) return 1; }
    {   const Shape s[3] = {{128, 1024, 40 << 10, 0, 0}, {1024, 256, 40 << 10, 0, 0}, {64, 512, 80 << 10, 0, 0}};
        printf("v2: + LDS footprints 40 / 40 / 80 KB\n");
        if (run_variant<4, 4, 4>("v2", s, buf, st, iters, reps)) return 1; }
    {   const Shape s[3] = {{128, 1024, 40 << 10, 0, 0}, {1024, 256, 40 << 10, 0, 0}, {64, 512, 80 << 10, 0, 0}};
        printf("v3: + kernel arguments 0.4 / 0.4 / 1.0 KB by value\n");
        if (run_variant<100, 100, 256>("v3", s, buf, st, iters, reps)) return 1; }
    {   // prep: 128 x 1024 threads write 5.6 MB = 10.7 floats per thread; dist reads ~4 MB of it, writes 0.6 MB; tail reads 2 MB, writes 1.2 MB
        const Shape s[3] = {{128, 1024, 40 << 10, 2, 11}, {1024, 256, 40 << 10, 4, 1}, {64, 512, 80 << 10, 16, 9}};
        printf("v4: + dirty bytes 5.6 / 1.0 / 1.2 MB per kernel, read by the next one\n");
        if (run_variant<100, 100, 256>("v4", s, buf, st, iters, reps)) return 1; }
    {   const Shape s[3] = {{128, 1024, 40 << 10, 2, 11}, {256, 256, 40 << 10, 16, 4}, {64, 512, 80 << 10, 16, 9}};
        printf("v5: v4 with the persistent grid cut to 256 x 256\n");
        if (run_variant<100, 100, 256>("v5", s, buf, st, iters, reps)) return 1; }
    {   const Shape s[3] = {{128, 512, 20 << 10, 4, 22}, {256, 256, 40 << 10, 16, 4}, {64, 512, 80 << 10, 16, 9}};
        printf("v6: v5 with 128 x 512 for the first kernel\n");
        if (run_variant<100, 100, 256>("v6", s, buf, st, iters, reps)) return 1; }
    return 0;
}
