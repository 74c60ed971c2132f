// Developer probe (not part of the product): what one dependent kernel launch costs on this box as a function of the launch
// shape -- workgroup size, dynamic LDS per workgroup, grid size, kernel-argument bytes, and whether the kernel leaves dirty
// lines behind.  Back-to-back launches of one kernel on one stream, HIP events around N of them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_probe.bin tools/launch_probe.hip && tools/launch_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Args { float* buf; int words; int pad[60]; };          // 256 B of kernel arguments, like the layer kernels' parameter structs
struct Small { float* buf; int words; };

template <int THREADS, typename A>
__global__ __launch_bounds__(THREADS) void probe(const A a) {
    extern __shared__ float sm[];
    if (a.words > 0) {                                         // every thread stores `words` floats: bytes left dirty for the boundary
        float* p = a.buf + ((size_t)blockIdx.x * THREADS + threadIdx.x) * a.words;
        for (int i = 0; i < a.words; ++i) p[i] = (float)i;
    }
    if (a.words < 0) sm[threadIdx.x] = 1.0f;                   // (never) keeps the LDS allocation alive
}

template <int THREADS, typename A>
static float run(int grid, size_t lds, int words, float* buf, int n) {
    A a{};
    a.buf = buf; a.words = words;
    auto k = probe<THREADS, A>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}

int main() {
    float* buf;
    hipMalloc(&buf, (size_t)1 << 28);
    const int n = 3000;
    printf("us per back-to-back launch (eager, one stream)\n");
    printf("%-8s %-6s %-8s %-7s %-6s %s\n", "threads", "grid", "lds", "args", "dirty", "us");
    for (int grid : {157, 256, 1024}) {
        for (size_t lds : {(size_t)0, (size_t)64 * 1024, (size_t)150 * 1024}) {
            printf("%-8d %-6d %-8zu %-7s %-6d %.2f\n", 256, grid, lds, "small", 0, run<256, Small>(grid, lds, 0, buf, n));
            printf("%-8d %-6d %-8zu %-7s %-6d %.2f\n", 512, grid, lds, "small", 0, run<512, Small>(grid, lds, 0, buf, n));
            printf("%-8d %-6d %-8zu %-7s %-6d %.2f\n", 512, grid, lds, "256B", 0, run<512, Args>(grid, lds, 0, buf, n));
        }
    }
    for (int words : {4, 16, 64}) {          // 157 x 512 threads x words x 4 B dirty per launch: 1.3 / 5.1 / 20.6 MB
        printf("%-8d %-6d %-8d %-7s %-6d %.2f\n", 512, 157, 150 * 1024, "256B", words * 4 * 512 * 157, run<512, Args>(157, 150 * 1024, words, buf, n));
        printf("%-8d %-6d %-8d %-7s %-6d %.2f\n", 512, 157, 0, "256B", words * 4 * 512 * 157, run<512, Args>(157, 0, words, buf, n));
    }
    printf("%-8d %-6d %-8d %-7s %-6d %.2f\n", 1024, 157, 150 * 1024, "small", 0, run<1024, Small>(157, 150 * 1024, 0, buf, n));
    printf("%-8d %-6d %-8d %-7s %-6d %.2f\n", 64, 157, 0, "small", 0, run<64, Small>(157, 0, 0, buf, n));
    return 0;
}
