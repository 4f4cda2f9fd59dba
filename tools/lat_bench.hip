// lat_bench.hip — cycles per DEPENDENT operation of the instruction classes the pivot step of rrlu_xcd2_kernel is made of (gfx950), with
// one wave per SIMD (256 threads) and two (512 threads, the kernel's arrangement: every wave runs the same chain).  s_memtime around N
// repetitions of a chain in which every operation consumes the previous one's result; lane 0 of wave 0 reports.
//   hipcc --offload-arch=gfx950 -O3 lat_bench.hip -o lat_bench && ./lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 512;
enum { T_ADD64, T_DPPMAX, T_RDLANE, T_BALLOT, T_LDS_RW, T_LDS_CHASE, T_BARRIER, T_L2_CHASE, T_DIV64, T_FMA_INDEP, T_COUNT };
static const char* NAMES[T_COUNT] = {"v_add_f64 (dependent)", "DPP reduction step (2 v_mov_dpp + v_max_f64)", "v_readlane -> s_add -> v_mov -> v_readlane",
                                     "v_cmp -> ballot -> s_ff1 -> v_readlane", "ds_write_b64 -> ds_read_b64 (same address)", "ds_read_b32 pointer chase",
                                     "s_barrier (all waves arrive together)", "buffer/global load sc1 pointer chase (L2 hit)", "f64 division (IEEE sequence, dependent)",
                                     "v_fma_f64 x 8 independent chains (issue rate)"};

__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(b & 0xFFFFFFFFll), (int)(b & 0xFFFFFFFFll), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0xB1, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ void __launch_bounds__(512) lat_kernel(int which, const unsigned* chase, double seed, unsigned long long* out, double* sink)
{
    __shared__ double lds_d[1024];
    __shared__ unsigned lds_i[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += blockDim.x) {
        lds_d[i] = seed + i;
        lds_i[i] = (unsigned)((i * 37 + 11) & 1023);
    }
    __syncthreads();
    double x = seed + lane, y = seed * 0.5 + 1.0;
    unsigned u = (unsigned)lane;
    unsigned long long t0 = 0, t1 = 0;
    auto begin = [&]() { __syncthreads(); t0 = __builtin_amdgcn_s_memtime(); };
    auto end = [&]() { t1 = __builtin_amdgcn_s_memtime(); };
    switch (which) {
    case T_ADD64:
        begin();
        for (int i = 0; i < N; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
        end();
        break;
    case T_DPPMAX:
        begin();
        for (int i = 0; i < N; ++i) {
            x = fmax(x, dpp_f64(x)) + 0.0;
            asm volatile("" : "+v"(x));
        }
        end();
        break;
    case T_RDLANE:
        begin();
        for (int i = 0; i < N; ++i) {
            int s = __builtin_amdgcn_readlane((int)u, 3);
            s = (s + 1) & 63;
            asm volatile("" : "+s"(s));
            u = (unsigned)s + (unsigned)lane;
            asm volatile("" : "+v"(u));
        }
        end();
        break;
    case T_BALLOT:
        begin();
        for (int i = 0; i < N; ++i) {
            const unsigned long long b = __ballot(u == (unsigned)(i & 63));
            const int l = b ? (int)__builtin_ctzll(b) : 0;
            u = (unsigned)__builtin_amdgcn_readlane((int)u, l) + (unsigned)lane;
            asm volatile("" : "+v"(u));
        }
        end();
        break;
    case T_LDS_RW:
        begin();
        for (int i = 0; i < N; ++i) {
            lds_d[tid] = x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            x = ((volatile double*)lds_d)[tid] + 1.0;
        }
        end();
        break;
    case T_LDS_CHASE:
        begin();
        for (int i = 0; i < N; ++i) u = ((volatile unsigned*)lds_i)[u & 1023];
        end();
        break;
    case T_BARRIER:
        begin();
        for (int i = 0; i < N; ++i) __syncthreads();
        end();
        break;
    case T_L2_CHASE:
        for (int i = 0; i < 64; ++i) u = __builtin_nontemporal_load(chase + (u & 4095)); // (warm: the table sits in the L2)
        begin();
        for (int i = 0; i < N; ++i) {
            unsigned v;
            const unsigned* p = chase + (u & 4095);
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            u = v;
        }
        end();
        break;
    case T_DIV64:
        begin();
        for (int i = 0; i < N; ++i) {
            x = y / x + 1.5;
            asm volatile("" : "+v"(x));
        }
        end();
        break;
    case T_FMA_INDEP: {
        double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        begin();
        for (int i = 0; i < N / 8; ++i) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                         "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(y), "v"(seed));
        }
        end();
        x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        break;
    }
    default: break;
    }
    if (tid == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + tid] = x + (double)u;
}

int main()
{
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned)((i * 1237 + 331) & 4095);
    unsigned* d_chase;
    unsigned long long* d_out;
    double* d_sink;
    hipMalloc(&d_chase, 4096 * 4);
    hipMalloc(&d_out, 8);
    hipMalloc(&d_sink, 1024 * 8);
    hipMemcpy(d_chase, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::printf("# cycles (s_memtime) per dependent operation, %d repetitions; one workgroup on one compute unit\n", N);
    std::printf("%-62s %12s %12s\n", "operation", "1 wave/SIMD", "2 waves/SIMD");
    for (int w = 0; w < T_COUNT; ++w) {
        double r[2];
        for (int k = 0; k < 2; ++k) {
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(k == 0 ? 256 : 512), 0, 0, w, d_chase, 1.25, d_out, d_sink);
                unsigned long long c = 0;
                hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
                if (c < best) best = c;
            }
            r[k] = (double)best / N;
        }
        std::printf("%-62s %12.1f %12.1f\n", NAMES[w], r[0], r[1]);
    }
    return 0;
}
