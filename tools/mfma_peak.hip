// Sustained v_mfma_f64_16x16x4 rate of the whole chip with nothing else in the loop (registers only): the practical ceiling
// for gemm_kernel on this box (power / clock limits included).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main()
{
    double* d;
    hipMalloc(&d, 4096 * 256 * 8);
    for (int wgs : {256, 512, 1024}) {
        for (int iters : {2000, 20000}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            k<<<wgs, 256>>>(d, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<wgs, 256>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2048.0 * 4.0 * iters * wgs * 4.0;
            printf("%d workgroups x %d iterations: %.3f ms  %.1f TF/s\n", wgs, iters, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
