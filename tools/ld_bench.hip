// ld_bench.hip — cost of agent-scope (sc1) loads issued by one wave: latency vs number of outstanding requests,
// 8-byte vs 16-byte, with 1 or many workgroups doing the same concurrently.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long ld8(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int K, int STRIDE_U64>
__global__ void k8(const unsigned long long* base, unsigned long long* out, int iters)
{
    extern __shared__ char pad[];
    if (threadIdx.x >= 64) return;
    const unsigned long long* p = base + (size_t)blockIdx.x * 64 * K * STRIDE_U64 + threadIdx.x * STRIDE_U64;
    unsigned long long acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        unsigned long long v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ld8(p + (size_t)k * 64 * STRIDE_U64);
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k];
        if (acc == 0x1234567ull) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}
template <int K>
__global__ void k16(const unsigned long long* base, unsigned long long* out, int iters)
{
    extern __shared__ char pad[];
    if (threadIdx.x >= 64) return;
    const unsigned long long* p = base + (size_t)blockIdx.x * 64 * K * 2 + threadIdx.x * 2;
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[K];
        if (K == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[K-1]) : "v"(p), "v"(p + 128) : "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k].x + v[k].w;
        if (acc == 0x1234567u) break;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *d, *o; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20); hipMalloc(&o, 4096 * 16);
    const int iters = 500, LDS = 84 * 1024;
    unsigned long long h[2 * 256];
#define RUN(name, kern, G) do { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    hipLaunchKernelGGL(kern, dim3(G), dim3(64), LDS, 0, d, o, iters); hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); \
    double s = 0; for (int i = 0; i < G; ++i) s += h[2 * i]; printf("%-46s G=%3d : %8.1f cycles / sweep\n", name, G, s / G / iters); } while (0)
    for (int G : {1, 86, 240}) {
        RUN("8B sc1 x1 per lane, 8B stride (512B/wave)", (k8<1, 1>), G);
        RUN("8B sc1 x1 per lane, 32B stride (2KB/wave)", (k8<1, 4>), G);
        RUN("8B sc1 x3 per lane, 32B stride", (k8<3, 4>), G);
        RUN("8B sc1 x6 per lane, 32B stride", (k8<6, 4>), G);
        RUN("8B sc1 x4 per lane, 16B stride", (k8<4, 2>), G);
        RUN("16B sc1 x1 per lane (1KB/wave)", (k16<1>), G);
        RUN("16B sc1 x2 per lane (2KB/wave)", (k16<2>), G);
    }
    return 0;
}
