// xchg2_bench.hip — all-gather of one key per workgroup through a shared table, cycles per round, for the key sizes
// that matter for the rrLU hop 1: GR granules of 8 bytes per key, packed (stride = GR words), W workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ void st8(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld8(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int GR>
__global__ void allgather(u64* keys, int W, int rounds, u64* out, int delay)
{
    __shared__ int bad;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) bad = 0;
    __syncthreads();
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int r = 1; r <= rounds; ++r) {
        const int par = r & 1;
        const u64 tag = (u64)r << 32;
        if (wave == 0 && lane == 0) {
            u64* k = keys + ((size_t)par * W + w) * GR;
#pragma unroll
            for (int g = 0; g < GR; ++g) st8(k + g, tag | (unsigned)(w + g));
        }
        if (wave == 1) {
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
            const u64* kb = keys + (size_t)par * W * GR;
            for (unsigned sp = 0;; ++sp) {
                bool ok = true;
                u64 g[4][GR];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = lane + 64 * j;
                    if (q < W)
#pragma unroll
                        for (int x = 0; x < GR; ++x) g[j][x] = ld8(kb + (size_t)q * GR + x);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = lane + 64 * j;
                    if (q < W)
#pragma unroll
                        for (int x = 0; x < GR; ++x) ok &= (g[j][x] >> 32) == (u64)r;
                }
                if (__all(ok)) break;
                if (sp > 1000000u) { bad = 1; break; }
            }
        }
        __syncthreads();
        if (bad) break;
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[2 * w] = t1 - t0; out[2 * w + 1] = bad; }
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    u64 *d, *o; hipMalloc(&d, 16 << 20); hipMalloc(&o, 4096 * 16);
    const int rounds = 500;
    std::vector<u64> h(2 * 256);
#define RUN(GR, W, DLY) do { hipMemset(d, 0, 16 << 20); \
    hipLaunchKernelGGL(allgather<GR>, dim3(W), dim3(384), 0, 0, d, W, rounds, o, DLY); hipMemcpy(h.data(), o, 16 * W, hipMemcpyDeviceToHost); \
    double s = 0; int bad = 0; for (int i = 0; i < W; ++i) { s += h[2 * i]; bad |= (int)h[2 * i + 1]; } printf("granules=%d W=%3d delay=%2d: %8.1f cycles / round%s\n", GR, W, DLY, s / W / rounds, bad ? "  TIMEOUT" : ""); } while (0)
    for (int W : {115, 172, 230}) for (int dly : {0, 8, 14}) { RUN(1, W, dly); RUN(2, W, dly); }
    return 0;
}
