// Shader clock seen by a ONE-workgroup kernel (the shape of the persistent half-sweep of configs[1] and of every panel kernel): s_memtime ticks per
// microsecond of the constant 100 MHz wall clock, for kernels of 20 us .. 2 ms, cold (after 200 ms of idling) and right behind a chip-wide kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/clk_bench tools/clk_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void spin_one(int iters, unsigned long long* out, double* sink)
{
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    double x = threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (x == 12345.678) *sink = x;
}
__global__ void burn_all(int iters, double* sink)
{
    double x = threadIdx.x * 1e-3 + blockIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    if (x == 12345.678) *sink = x;
}
int main()
{
    unsigned long long* d_out;
    double* d_sink;
    (void)hipMalloc(&d_out, 16);
    (void)hipMalloc(&d_sink, 8);
    unsigned long long h[2];
    auto run = [&](int iters, const char* label) {
        hipLaunchKernelGGL(spin_one, dim3(1), dim3(64), 0, 0, iters, d_out, d_sink);
        (void)hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        std::printf("%-44s %8d dependent FMAs: %9.1f us, %7.1f s_memtime ticks per us, %5.2f ticks per FMA\n", label, iters, h[1] / 100.0,
                    (double)h[0] / (h[1] / 100.0), (double)h[0] / iters);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int iters : {2000, 20000, 200000}) {
            usleep(200000);
            run(iters, "cold (200 ms idle), one wave");
        }
        for (int iters : {2000, 20000, 200000}) {
            hipLaunchKernelGGL(burn_all, dim3(2048), dim3(256), 0, 0, 400000, d_sink);
            run(iters, "behind a chip-wide kernel, one wave");
        }
        for (int iters : {2000, 20000}) {
            for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(spin_one, dim3(1), dim3(64), 0, 0, iters, d_out, d_sink);
            run(iters, "behind 50 one-wave kernels, one wave");
        }
    }
    return 0;
}
