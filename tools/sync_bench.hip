// sync_bench.hip — completion latency of a short kernel: hipStreamSynchronize vs spinning on a flag the kernel writes
// into pinned host memory (system-scope store after its last result).   hipcc --offload-arch=gfx950 -O3 sync_bench.hip -o sync_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void work_kernel(volatile unsigned* host_flag, unsigned seq, int spin, unsigned long long* sink)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) {
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *sink = t0;
        __threadfence_system();
        *host_flag = seq;
    }
}

int main(int argc, char** argv)
{
    if (argc > 1) { // "spin": ask the runtime to busy-wait in its synchronisation calls
        const hipError_t e = hipSetDeviceFlags(hipDeviceScheduleSpin);
        std::printf("hipSetDeviceFlags(hipDeviceScheduleSpin) -> %d\n", (int)e);
    }
    hipStream_t st;
    hipStreamCreate(&st);
    unsigned* flag;
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    *flag = 0;
    unsigned long long* sink;
    hipMalloc(&sink, 8);
    const int iters = 2000;
    for (int spin : {2000, 100000}) { // ~1 us and ~50 us of device work (100 MHz memtime ticks: adjust below)
        for (int mode = 0; mode < 2; ++mode) {
            // warm up
            for (int i = 0; i < 50; ++i) {
                hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, flag, 0u, spin / 100, sink);
                hipStreamSynchronize(st);
            }
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= iters; ++i) {
                hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, flag, (unsigned)i + 1000000u * mode + 77u * spin, spin / 100, sink);
                if (mode == 0) {
                    hipStreamSynchronize(st);
                } else {
                    const unsigned want = (unsigned)i + 1000000u * mode + 77u * spin;
                    while (*(volatile unsigned*)flag != want) {
                    }
                }
            }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            std::printf("device work ~%d memtime ticks, %s: %.2f us per launch+completion\n", spin / 100,
                        mode == 0 ? "hipStreamSynchronize" : "spin on pinned flag  ", us);
        }
    }
    return 0;
}
