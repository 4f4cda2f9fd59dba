// xcd_bench.hip — what one pivot step's exchange costs when every participating workgroup sits on ONE XCD (shared L2)
// compared with workgroups spread over the chip.  Emulates the rrLU hand-offs: 16-byte tagged key per workgroup
// (all-gather), a pivot column of M tagged 16-byte rows published by the winner (+ a few speculating workgroups),
// read by every workgroup.  Store flavours: plain (line stays in the XCD's L2: valid only inside one XCD) vs sc1
// (write-through: valid anywhere).  Loads are always sc1 (L1 bypass).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_bench.hip -o tools/xcd_bench && tools/xcd_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
__device__ __forceinline__ unsigned hw_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
template <bool SC1> __device__ __forceinline__ void st16(void* p, u32x4 v)
{
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld16(const void* p)
{
    u32x4 o;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(p) : "memory");
    return o;
}
__device__ __forceinline__ void ld16x2_plain(const void* p0, const void* p1, u32x4& a, u32x4& b)
{
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p0), "v"(p1)
                 : "memory");
}
__device__ __forceinline__ void ld16x2(const void* p0, const void* p1, u32x4& a, u32x4& b)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p0), "v"(p1)
                 : "memory");
}

__global__ void census(unsigned* out, u64* when)
{
    extern __shared__ char pad[];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc_id();
        out[2 * blockIdx.x + 1] = hw_id();
        when[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

struct XArgs {
    int want_xcc;     // >= 0: only workgroups on this XCC take part; -1: the first W tickets whatever their XCC
    int W, M, rounds;
    int spec_every;   // besides the winner, workgroups with (w + round) % spec_every == 0 publish their column too
    int poll_sleep;
    int flags;        // 1 key gather, 2 column publish, 4 column read, 8 divide + LDS, 16 phase stamps, 32 readers sleep first
    int work;         // dependent f64 ops between rounds (emulates the update pass)
    unsigned* ticket; // [0] tickets [1] arrivals
    u32x4* keys;      // [2][W]
    u32x4* cols;      // [2][W][M]
    u64* out;         // per rank: [0] cycles [1] bad [2..9] stamps
    unsigned salt;
};

template <bool SC1>
__global__ void __launch_bounds__(512) xchg(XArgs p)
{
    extern __shared__ char smem[];
    int* sh = (int*)smem;           // [0] rank [1] bad [2] winner
    double* lcol = (double*)(smem + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int rank = -1;
        if (p.want_xcc < 0 || (int)xcc_id() == p.want_xcc) {
            rank = (int)atomicAdd(&p.ticket[0], 1u);
            if (rank >= p.W) rank = -1;
        }
        sh[0] = rank;
        sh[1] = 0;
    }
    __syncthreads();
    const int w = sh[0];
    if (w < 0) return;
    if (tid == 0) { // wait until everybody is resident
        atomicAdd(&p.ticket[1], 1u);
        unsigned sp = 0;
        while (__hip_atomic_load(&p.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.W)
            if (++sp > 20000000u) { sh[1] = 1; break; }
    }
    __syncthreads();
    if (sh[1]) { if (tid == 0) p.out[16 * w + 1] = 2; return; }
    const bool stat = (p.flags & 64) != 0;
    if (stat) {
        for (int par = 0; par < 2; ++par)
            for (int i = tid; i < p.M; i += 512) {
                const double v = (double)(w * 1000 + i);
                const u64 vb = (u64)__double_as_longlong(v);
                u32x4 g;
                g.x = (unsigned)vb; g.y = p.salt * 65536u; g.z = (unsigned)(vb >> 32); g.w = p.salt * 65536u;
                st16<SC1>(p.cols + ((size_t)par * p.W + w) * p.M + i, g);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&p.ticket[2], 1u);
            unsigned sp = 0;
            while (__hip_atomic_load(&p.ticket[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.W)
                if (++sp > 20000000u) { sh[1] = 1; break; }
        }
        __syncthreads();
    }
    u64 st[4] = {0, 0, 0, 0};
    u64 t0 = __builtin_amdgcn_s_memtime(), tl = t0;
    for (int r = 1; r <= p.rounds; ++r) {
        const int par = r & 1;
        const unsigned tag = p.salt * 65536u + (unsigned)r;
        const unsigned ctag = stat ? p.salt * 65536u : tag;
        const double radd = stat ? 0.0 : 0.5 * r;
        const int winner = (int)(((unsigned)r * 2654435761u) >> 8) % p.W;
        if (p.work > 0) {
            double acc = (double)tid;
            for (int i = 0; i < p.work; ++i) acc = acc * 1.0000001 + 0.5;
            if (acc == 12345.678) lcol[0] = acc;
        }
        if ((p.flags & 1) && wave == 0 && lane == 0) {
            u32x4 k;
            k.x = (unsigned)w; k.y = tag; k.z = (unsigned)r; k.w = tag;
            st16<SC1>(p.keys + (size_t)par * p.W + w, k);
        }
        const bool pub = (w == winner) || (p.spec_every > 0 && ((w + r) % p.spec_every) == 0);
        if (pub && (p.flags & 2) && !stat) {
            for (int i = tid; i < p.M; i += 512) {
                const double v = (double)(w * 1000 + i) + 0.5 * r;
                const u64 vb = (u64)__double_as_longlong(v);
                u32x4 g;
                g.x = (unsigned)vb; g.y = tag; g.z = (unsigned)(vb >> 32); g.w = tag;
                st16<SC1>(p.cols + ((size_t)par * p.W + w) * p.M + i, g);
            }
        }
        if ((p.flags & 16) && tid == 0) { u64 n = __builtin_amdgcn_s_memtime(); st[0] += n - tl; tl = n; }
        if (!(p.flags & 1)) { if (tid == 0) sh[2] = winner; }
        else if (wave == 1) {
            for (int d = 0; d < p.poll_sleep; ++d) __builtin_amdgcn_s_sleep(1);
            const u32x4* kb = p.keys + (size_t)par * p.W;
            unsigned sp = 0;
            for (;;) {
                bool ok = true;
                u32x4 g = ld16(kb + (lane < p.W ? lane : 0));
                if (lane < p.W) ok = (g.y == tag) && (g.w == tag) && (g.x == (unsigned)lane);
                if (__all(ok)) break;
                if (++sp > 2000000u) { sh[1] = 1; break; }
            }
            if (lane == 0) sh[2] = winner;
        }
        __syncthreads();
        if ((p.flags & 16) && tid == 0) { u64 n = __builtin_amdgcn_s_memtime(); st[1] += n - tl; tl = n; }
        if (sh[1]) break;
        const int ww = sh[2];
        const u32x4* src = p.cols + ((size_t)par * p.W + ww) * p.M;
        const int i0 = tid, i1 = tid + 512;
        if ((p.flags & 32)) __builtin_amdgcn_s_sleep(4);
        if ((p.flags & 4) && i0 < p.M) {
            u32x4 a, b;
            unsigned sp = 0;
            for (;;) {
                if (p.flags & 128) ld16x2_plain(src + i0, src + (i1 < p.M ? i1 : i0), a, b);
                else ld16x2(src + i0, src + (i1 < p.M ? i1 : i0), a, b);
                if (a.y == ctag && a.w == ctag && b.y == ctag && b.w == ctag) break;
                if (++sp > 2000000u) { sh[1] = 1; break; }
            }
            if (tid == 0) st[3] += sp;
            const double v0 = __longlong_as_double((long long)(((u64)a.z << 32) | a.x));
            const double v1 = __longlong_as_double((long long)(((u64)b.z << 32) | b.x));
            if (v0 != (double)(ww * 1000 + i0) + radd) sh[1] = 3;
            if (i1 < p.M && v1 != (double)(ww * 1000 + i1) + radd) sh[1] = 3;
            if (p.flags & 8) {
                const double dv = (double)(r | 3);
                lcol[i0] = v0 / dv;
                if (i1 < p.M) lcol[i1] = v1 / dv;
            }
        }
        __syncthreads();
        if ((p.flags & 16) && tid == 0) { u64 n = __builtin_amdgcn_s_memtime(); st[2] += n - tl; tl = n; }
        if (sh[1]) break;
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        p.out[16 * w] = t1 - t0;
        p.out[16 * w + 1] = (u64)sh[1];
        for (int i = 0; i < 4; ++i) p.out[16 * w + 2 + i] = st[i];
        p.out[16 * w + 8] = xcc_id();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int LDS = 84 * 1024;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    unsigned* d_x; u64* d_when; CK(hipMalloc(&d_x, 4096 * 8)); CK(hipMalloc(&d_when, 4096 * 8));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    auto do_census = [&](hipStream_t s, int G, const char* name) {
        hipLaunchKernelGGL(census, dim3(G), dim3(64), LDS, s, d_x, d_when);
        hipStreamSynchronize(s);
        std::vector<unsigned> x(2 * G); hipMemcpy(x.data(), d_x, G * 8, hipMemcpyDeviceToHost);
        int cnt[16] = {0}; int rr = 0;
        for (int i = 0; i < G; ++i) { cnt[x[2 * i] & 15]++; if (i >= 8 && x[2 * i] == x[2 * (i - 8)]) rr++; }
        printf("%-28s G=%4d blocks per XCC:", name, G); for (int i = 0; i < 8; ++i) printf(" %3d", cnt[i]);
        printf("   b~b+8 same-XCC %d/%d  first:", rr, G - 8); for (int i = 0; i < 10 && i < G; ++i) printf(" %u", x[2 * i]); printf("\n");
    };
    // `xcd_bench floor`: only the exchange of one pivot step (key all-gather + fresh 700-row column + two barriers, no arithmetic) —
    // what bench.py quotes as the latency floor of the single-XCD rrLU kernel, re-measured on the box it runs on
    const bool floor_only = argc > 1 && std::string(argv[1]) == "floor";
    if (!floor_only) {
    do_census(0, 256, "census default stream");
    do_census(0, 64, "census default stream");
    do_census(0, 1024, "census default stream");
    }
    // CU-masked streams: which XCCs do the workgroups land on?
    if (!floor_only) {
        struct MaskCase { const char* name; unsigned m[8]; };
        MaskCase cases[] = {
            {"mask bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
            {"mask bits 32..63", {0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}},
            {"mask every 8th bit (i%8==0)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
            {"mask every 8th bit (i%8==3)", {0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u}},
            {"mask bits 0..127", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0}},
        };
        for (auto& c : cases) {
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, c.m);
            if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask failed: %s\n", c.name, hipGetErrorString(e)); continue; }
            do_census(s, 256, c.name);
            hipStreamDestroy(s);
        }
    }
    // exchange benchmark
    unsigned* d_t; u32x4 *d_k, *d_c; u64* d_o;
    const int MAXW = 256, MAXM = 1024;
    CK(hipMalloc(&d_t, 64)); CK(hipMalloc(&d_k, sizeof(u32x4) * 2 * MAXW)); CK(hipMalloc(&d_c, sizeof(u32x4) * 2 * (size_t)MAXW * MAXM)); CK(hipMalloc(&d_o, 16 * 8 * MAXW));
    CK(hipFuncSetAttribute((const void*)xchg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)xchg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    unsigned salt = 1;
    auto run = [&](bool sc1, int want_xcc, int W, int M, int spec_every, int sleep, int G, int flags, int work) {
        hipMemset(d_t, 0, 64); hipMemset(d_o, 0, 16 * 8 * MAXW);
        hipMemset(d_k, 0, sizeof(u32x4) * 2 * MAXW);
        XArgs a; a.want_xcc = want_xcc; a.W = W; a.M = M; a.rounds = 2000; a.spec_every = spec_every; a.poll_sleep = sleep; a.flags = flags; a.work = work;
        a.ticket = d_t; a.keys = d_k; a.cols = d_c; a.out = d_o; a.salt = salt++;
        if (sc1) hipLaunchKernelGGL(xchg<true>, dim3(G), dim3(512), LDS, 0, a);
        else hipLaunchKernelGGL(xchg<false>, dim3(G), dim3(512), LDS, 0, a);
        hipError_t e = hipDeviceSynchronize();
        std::vector<u64> o(16 * W); hipMemcpy(o.data(), d_o, 16 * 8 * W, hipMemcpyDeviceToHost);
        double s = 0, s0 = 0, s1 = 0, s2 = 0; u64 bad = 0; int xs[16] = {0};
        for (int i = 0; i < W; ++i) { s += o[16 * i]; bad |= o[16 * i + 1]; xs[o[16 * i + 8] & 15]++; }
        s0 = (double)o[2]; s1 = (double)o[3]; s2 = (double)o[4];
        printf("%s xcc=%2d W=%3d M=%4d spec=%d sleep=%2d flags=%2d work=%4d: %8.1f cyc/round (rank0: publish %6.1f gather %6.1f column %6.1f colretry %.2f)%s%s  xcc-hist:",
               sc1 ? "sc1  " : "plain", want_xcc, W, M, spec_every, sleep, flags, work, s / W / a.rounds, s0 / a.rounds, s1 / a.rounds, s2 / a.rounds, (double)o[5] / a.rounds,
               bad ? "  BAD=" : "", bad ? (bad == 3 ? "value" : (bad == 2 ? "arrival" : "timeout")) : "");
        for (int i = 0; i < 8; ++i) printf(" %d", xs[i]); printf(" %s\n", e == hipSuccess ? "" : hipGetErrorString(e));
    };
    const int M = 700;
    if (floor_only) {
        double best_ns = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            run(false, 0, 32, M, 4, 0, 256, 1 + 2 + 4, 0); // key gather + fresh column + barriers
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms * 1e6 / 2000 < best_ns) best_ns = ms * 1e6 / 2000; // (the launch: 2000 rounds; its fixed cost is < 1 %)
        }
        printf("floor_ns_per_round=%.1f\n", best_ns);
        return 0;
    }
    { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0); run(false, 0, 32, M, 4, 0, 256, 0, 800); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("  (that launch: %.3f ms wall for 2000 rounds => %.1f ns/round)\n", ms, ms * 1e6 / 2000); }
    run(false, 0, 32, M, 4, 0, 256, 0, 0);              // two barriers only
    run(false, 0, 32, M, 4, 0, 256, 4 + 64, 0);         // static column, sc1 loads
    run(false, 0, 32, M, 4, 0, 256, 4 + 64 + 128, 0);   // static column, plain loads (L1)
    run(false, 0, 32, 512, 4, 0, 256, 4 + 64, 0);       // one load per thread
    run(false, 0, 32, 64, 4, 0, 256, 4 + 64, 0);        // one wave only
    run(false, 0, 1, M, 4, 0, 8, 4 + 64, 0);            // a single workgroup
    run(false, 0, 32, M, 4, 0, 256, 2 + 4, 0);          // fresh column, winner known
    run(false, 0, 32, 64, 4, 0, 256, 2 + 4, 0);         // fresh column of 64 rows
    run(false, 0, 32, M, 4, 0, 256, 1, 0);              // key gather only
    run(false, 0, 32, M, 4, 0, 256, 1 + 2 + 4, 0);
    run(true, 0, 32, M, 4, 0, 256, 4 + 64, 0);          // static column stored sc1
    run(true, 0, 32, M, 4, 0, 256, 2 + 4, 0);
    return 0;
}
