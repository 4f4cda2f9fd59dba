// hop_bench.hip — ping-pong latency between two workgroups (one per CU) on gfx950:
//   variant 0: sc1 (write-through) stores + sc1 loads           (placement-independent form)
//   variant 1: plain stores + sc1 loads                         (valid only when both CUs share an XCD/L2)
// Reports round-trip cycles (s_memtime) for a same-XCD pair and a cross-XCD pair, and a 1->N broadcast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

__global__ void census(unsigned* out)
{
    extern __shared__ char pad[];
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

template <int VARIANT>
__global__ void pingpong(unsigned long long* flag_a, unsigned long long* flag_b, int wa, int wb, int iters,
                         unsigned long long* cycles)
{
    extern __shared__ char pad[];
    const int w = blockIdx.x;
    if (w != wa && w != wb) return;
    if (threadIdx.x != 0) return;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= iters; ++i) {
        if (w == wa) {
            if (VARIANT == 0) __hip_atomic_store(flag_a, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (VARIANT == 2) (void)__hip_atomic_exchange(flag_a, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { *((volatile unsigned long long*)flag_a) = (unsigned long long)i; }
            { unsigned sp = 0; while (__hip_atomic_load(flag_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) { if (++sp > 2000000u) { cycles[1] = 1; return; } } }
        } else {
            { unsigned sp = 0; while (__hip_atomic_load(flag_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) { if (++sp > 2000000u) { cycles[1] = 1; return; } } }
            if (VARIANT == 0) __hip_atomic_store(flag_b, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (VARIANT == 2) (void)__hip_atomic_exchange(flag_b, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { *((volatile unsigned long long*)flag_b) = (unsigned long long)i; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (w == wa) cycles[0] = t1 - t0;
}

// 1 -> N broadcast + N -> 1 gather (all members on `members` list), each round: leader writes epoch,
// members poll and ack into their own slot; leader polls all slots.
template <int VARIANT>
__global__ void bcast_gather(unsigned long long* flag, unsigned long long* acks, const int* member_of, int leader,
                             int nmembers, int iters, unsigned long long* cycles)
{
    extern __shared__ char pad[];
    const int w = blockIdx.x;
    const int me = member_of[w];
    if (me < 0) return;
    const int lane = threadIdx.x;
    if (lane >= 64) return;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= iters; ++i) {
        if (w == leader) {
            if (lane == 0) {
                if (VARIANT == 0) __hip_atomic_store(flag, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *((volatile unsigned long long*)flag) = (unsigned long long)i;
            }
            for (unsigned sp = 0;;) {
                bool ok = true;
                for (int m = lane; m < nmembers; m += 64)
                    ok &= __hip_atomic_load(acks + 16 * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)i;
                if (__all(ok)) break;
                if (++sp > 2000000u) { cycles[1] = 1; return; }
            }
        } else {
            if (lane == 0) {
                { unsigned sp = 0; bool dead = false; while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) { if (++sp > 2000000u) { dead = true; break; } } if (dead) { cycles[1] = 1; return; } }
                if (VARIANT == 0) __hip_atomic_store(acks + 16 * me, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *((volatile unsigned long long*)(acks + 16 * me)) = (unsigned long long)i;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (w == leader && lane == 0) cycles[0] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int G = 256, LDS = 84 * 1024;
    unsigned* d_x; CK(hipMalloc(&d_x, G * 4));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipLaunchKernelGGL(census, dim3(G), dim3(64), LDS, 0, d_x);
    std::vector<unsigned> x(G); CK(hipMemcpy(x.data(), d_x, G * 4, hipMemcpyDeviceToHost));
    int cnt[16] = {0}; for (int i = 0; i < G; ++i) cnt[x[i] & 15]++;
    printf("census (blocks per XCC id):"); for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]); printf("\n");
    printf("first 16 block->xcc:"); for (int i = 0; i < 16; ++i) printf(" %u", x[i]); printf("\n");
    int same = -1, cross = -1;
    for (int i = 1; i < G; ++i) { if (same < 0 && x[i] == x[0]) same = i; if (cross < 0 && x[i] != x[0]) cross = i; }
    unsigned long long *d_f, *d_c; CK(hipMalloc(&d_f, 1 << 20)); CK(hipMalloc(&d_c, 64));
    const int iters = 2000;
    auto run_pp = [&](int variant, int wb, const char* name) {
        hipMemset(d_f, 0, 1 << 20); hipMemset(d_c, 0, 64);
        CK(hipFuncSetAttribute((const void*)pingpong<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        CK(hipFuncSetAttribute((const void*)pingpong<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        CK(hipFuncSetAttribute((const void*)pingpong<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        if (variant == 0) hipLaunchKernelGGL(pingpong<0>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 1024, 0, wb, iters, d_c);
        else if (variant == 2) hipLaunchKernelGGL(pingpong<2>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 1024, 0, wb, iters, d_c);
        else hipLaunchKernelGGL(pingpong<1>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 1024, 0, wb, iters, d_c);
        unsigned long long c2[2] = {0,0}; CK(hipMemcpy(c2, d_c, 16, hipMemcpyDeviceToHost)); unsigned long long c = c2[0]; if (c2[1]) printf("  [TIMEOUT in kernel] ");
        printf("%-44s round trip %8.1f cycles  (one hop ~%.1f)\n", name, (double)c / iters, (double)c / iters / 2);
        return 0;
    };
    run_pp(0, same, "pingpong same-XCD  sc1 store / sc1 load:");
    run_pp(1, same, "pingpong same-XCD  plain store / sc1 load:");
    run_pp(2, same, "pingpong same-XCD  atomic xchg / sc1 load:");
    run_pp(0, cross, "pingpong cross-XCD sc1 store / sc1 load:");
    run_pp(2, cross, "pingpong cross-XCD atomic xchg / sc1 load:");
    // broadcast+gather among the blocks of block 0's XCD (variant 1) and among 64 blocks spread over all XCDs (variant 0)
    std::vector<int> member(G, -1); int n = 0;
    for (int i = 0; i < G; ++i) if (x[i] == x[0]) member[i] = n++;
    int* d_m; CK(hipMalloc(&d_m, G * 4)); CK(hipMemcpy(d_m, member.data(), G * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)bcast_gather<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)bcast_gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    for (int variant = 1; variant >= 0; --variant) {
        hipMemset(d_f, 0, 1 << 20); hipMemset(d_c, 0, 64);
        if (variant) hipLaunchKernelGGL(bcast_gather<1>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 4096, d_m, 0, n, iters, d_c);
        else hipLaunchKernelGGL(bcast_gather<0>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 4096, d_m, 0, n, iters, d_c);
        unsigned long long c = 0; CK(hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost));
        printf("bcast+gather same-XCD team of %d, %s stores: %8.1f cycles per round\n", n, variant ? "plain" : "sc1", (double)c / iters);
    }
    for (int nm : {16, 64, 128}) {
        std::fill(member.begin(), member.end(), -1);
        for (int i = 0; i < nm; ++i) member[i * (G / nm)] = i;
        CK(hipMemcpy(d_m, member.data(), G * 4, hipMemcpyHostToDevice));
        hipMemset(d_f, 0, 1 << 20); hipMemset(d_c, 0, 64);
        hipLaunchKernelGGL(bcast_gather<0>, dim3(G), dim3(64), LDS, 0, d_f, d_f + 4096, d_m, 0, nm, iters, d_c);
        unsigned long long c = 0; CK(hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost));
        printf("bcast+gather cross-XCD team of %d, sc1 stores:   %8.1f cycles per round\n", nm, (double)c / iters);
    }
    return 0;
}
