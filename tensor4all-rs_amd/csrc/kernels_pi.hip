// kernels_pi.hip — K1: candidate-matrix (Π) build on the device.
//
// Replaces the Π evaluation loop of update_pivots / sweep1site / fill_site_tensors
// (tensor4all-tensorci/src/tensorci2.rs:1859-1893, :946-957, :1101-1145) for the built-in function family
// of include/t4a_testfunctions.h: every row / column multi-index has been folded (on the host, from the
// I/J index tables) into n_acc integer accumulators, so one entry costs n_acc integer adds plus the scalar
// stage.  HBM-bound on the 8*M*N byte store: consecutive lanes write consecutive rows of a column.
// max_sample_value bookkeeping (tensorci2.rs:2009-2014) is fused: max of sqrt(v*v) via an integer
// atomicMax on the (non-negative) bit pattern.
#include "kernels.hpp"

namespace t4a {

namespace {

__device__ __forceinline__ void wave_absmax_commit(double av, unsigned long long* max_abs_bits)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(av, off);
        if (o > av) av = o;
    }
    if ((threadIdx.x & 63) == 0 && av > 0.0 && max_abs_bits)
        atomicMax(max_abs_bits, (unsigned long long)__double_as_longlong(av));
}

__global__ void __launch_bounds__(256) pi_eval_kernel(FnDevice fn, const uint64_t* __restrict__ rowacc, int M,
                                                      const uint64_t* __restrict__ colacc, int N,
                                                      double* __restrict__ out, int ld, int transpose_out,
                                                      unsigned long long* max_abs_bits)
{
    // blockIdx.y walks columns (grid-stride), threads walk rows: coalesced column-major stores
    double av = 0.0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t racc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
    if (i < M)
        for (int k = 0; k < fn.n_acc; ++k) racc[k] = rowacc[(size_t)i * fn.n_acc + k];
    for (int j = blockIdx.y; j < N; j += gridDim.y) {
        if (i < M) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < fn.n_acc; ++k) acc[k] = racc[k] + colacc[(size_t)j * fn.n_acc + k];
            const double v = t4a_fn_value(fn.fid, acc, fn.params);
            if (transpose_out)
                out[(size_t)i * ld + j] = v;
            else
                out[(size_t)j * ld + i] = v;
            const double a = sqrt(v * v);
            if (a > av) av = a;
        }
    }
    wave_absmax_commit(av, max_abs_bits);
}

__global__ void __launch_bounds__(256) pi_eval_batched_kernel(FnDevice fn, const PiJob* __restrict__ jobs)
{
    const PiJob jb = jobs[blockIdx.z];
    double av = 0.0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * (int)blockDim.x >= jb.M) return; // whole block outside this (smaller) job
    uint64_t racc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
    if (i < jb.M)
        for (int k = 0; k < fn.n_acc; ++k) racc[k] = jb.rowacc[(size_t)i * fn.n_acc + k];
    for (int j = blockIdx.y; j < jb.N; j += gridDim.y) {
        if (i < jb.M) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < fn.n_acc; ++k) acc[k] = racc[k] + jb.colacc[(size_t)j * fn.n_acc + k];
            const double v = t4a_fn_value(fn.fid, acc, fn.params);
            jb.out[(size_t)j * jb.ld + i] = v;
            const double a = sqrt(v * v);
            if (a > av) av = a;
        }
    }
    wave_absmax_commit(av, jb.max_abs_bits);
}

__global__ void __launch_bounds__(256) absmax_kernel(const double* __restrict__ data, size_t count,
                                                     unsigned long long* max_abs_bits)
{
    double av = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const double v = data[i];
        const double a = sqrt(v * v);
        if (a > av) av = a;
    }
    wave_absmax_commit(av, max_abs_bits);
}

// accumulators: pinned host arena -> device buffer.  A kernel instead of a DMA copy: the candidate-matrix kernel that follows on
// the same stream starts a few microseconds sooner behind a kernel than behind an SDMA transfer, and a bond update is a chain
// of such hand-offs.
__global__ void __launch_bounds__(256) stage_copy_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

} // namespace

void stage_copy_launch(const uint64_t* pinned_src, uint64_t* dst, size_t count, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(stage_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pinned_src, dst, count);
}

void pi_eval_launch(const FnDevice& fn, const uint64_t* rowacc, int M, const uint64_t* colacc, int N, double* out,
                    int ld, bool transpose_out, unsigned long long* max_abs_bits, hipStream_t stream)
{
    if (M <= 0 || N <= 0) return;
    dim3 block(256);
    int gy = N < 2048 ? N : 2048;
    dim3 grid((M + 255) / 256, gy);
    hipLaunchKernelGGL(pi_eval_kernel, grid, block, 0, stream, fn, rowacc, M, colacc, N, out, ld,
                       transpose_out ? 1 : 0, max_abs_bits);
}

void pi_eval_batched_launch(const FnDevice& fn, const PiJob* d_jobs, int n_jobs, int max_M, int max_N, hipStream_t stream)
{
    if (n_jobs <= 0 || max_M <= 0 || max_N <= 0) return;
    const int gy = max_N < 512 ? max_N : 512;
    dim3 grid((max_M + 255) / 256, gy, n_jobs);
    hipLaunchKernelGGL(pi_eval_batched_kernel, grid, dim3(256), 0, stream, fn, d_jobs);
}

namespace {

struct FillSmallArgs {
    FnDevice fn;
    const PiJob* pis;
    const LuProblem* lups;
    const PackJob* packs;
    int n_jobs, last_site;
};

__global__ void __launch_bounds__(256) fill_small_kernel(FillSmallArgs a)
{
    __shared__ double As[FILL_SMALL_MAX_N * FILL_SMALL_MAX_N];   // P^T, n x n, column-major (ld = n)
    __shared__ double Bs[FILL_SMALL_MAX_N * FILL_SMALL_MAX_RHS]; // Pi1^T, n x nrhs (ld = n); last site: Pi1, ni x nj (ld = ni)
    __shared__ double red_v[4];
    __shared__ int red_i[4];
    __shared__ int piv_s, info_s;
    __shared__ double pivval_s;
    const int k = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    if (k >= a.n_jobs) return;
    const bool last = a.last_site && k == a.n_jobs - 1;
    const int K = a.fn.n_acc;
    // ---- evaluations (pi_eval_batched_kernel: value = f(row accumulators + column accumulators)) ----
    double pmax = 0.0;
    for (int q = 0; q < (last ? 1 : 2); ++q) {
        const PiJob jb = a.pis[2 * k + q];
        double* const dst = q == 0 ? Bs : As;
        double av = 0.0;
        for (int e = tid; e < jb.M * jb.N; e += T) {
            const int i = e % jb.M, j = e / jb.M;
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int c = 0; c < K; ++c) acc[c] = jb.rowacc[(size_t)i * K + c] + jb.colacc[(size_t)j * K + c];
            const double v = t4a_fn_value(a.fn.fid, acc, a.fn.params);
            dst[(size_t)j * jb.ld + i] = v;
            const double av1 = sqrt(v * v);
            if (av1 > av) av = av1;
        }
        if (jb.max_abs_bits) { // max sqrt(v * v) over the pivot matrix (zero-pivot-matrix guard, tensorci2.rs:1154-1157)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(av, off);
                if (o > av) av = o;
            }
            if (lane == 0) red_v[wave] = av;
            __syncthreads();
            pmax = fmax(fmax(red_v[0], red_v[1]), fmax(red_v[2], red_v[3]));
            if (tid == 0 && pmax > 0.0) atomicMax(jb.max_abs_bits, (unsigned long long)__double_as_longlong(pmax));
            __syncthreads();
        }
    }
    __syncthreads();
    const PackJob pk = a.packs[k];
    const size_t total = (size_t)pk.L * pk.S * pk.R;
    if (last) { // the last site stores Pi1 itself (tensorci2.rs:1109-1128), R == 1
        for (size_t e = tid; e < total; e += T) {
            const int l = (int)(e % pk.L), s_ = (int)((e / pk.L) % pk.S), r = (int)(e / ((size_t)pk.L * pk.S));
            pk.core[e] = Bs[(size_t)r * pk.ld + (l * pk.S + s_)];
        }
        return;
    }
    // ---- partial-pivot LU of P^T with the row swaps applied to the right-hand sides (lu_kernel, operation for operation) ----
    const LuProblem pr = a.lups[k];
    const int n = pr.n, nrhs = pr.nrhs;
    if (tid == 0) info_s = (pmax < 2.220446049250313e-16) ? -1 : 0; // every |p| < EPS: zero core
    __syncthreads();
    if (info_s == 0) {
        for (int c = 0; c < n; ++c) {
            double bv = -1.0;
            int bi = 0x7fffffff;
            for (int i = c + tid; i < n; i += T) {
                const double v = fabs(As[(size_t)c * n + i]);
                if (v > bv || (v == bv && i < bi)) {
                    bv = v;
                    bi = i;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (lane == 0) {
                red_v[wave] = bv;
                red_i[wave] = bi;
            }
            __syncthreads();
            if (tid == 0) {
                double v = red_v[0];
                int idx = red_i[0];
                for (int q = 1; q < (T >> 6); ++q)
                    if (red_v[q] > v || (red_v[q] == v && red_i[q] < idx)) {
                        v = red_v[q];
                        idx = red_i[q];
                    }
                piv_s = idx;
                pr.piv[c] = idx;
                if (!(v > 0.0) && info_s == 0) info_s = c + 1;
            }
            __syncthreads();
            const int p = piv_s;
            if (p != c && p < n) {
                for (int q = tid; q < n; q += T) {
                    const double t = As[(size_t)q * n + c];
                    As[(size_t)q * n + c] = As[(size_t)q * n + p];
                    As[(size_t)q * n + p] = t;
                }
                for (int q = tid; q < nrhs; q += T) {
                    const double t = Bs[(size_t)q * n + c];
                    Bs[(size_t)q * n + c] = Bs[(size_t)q * n + p];
                    Bs[(size_t)q * n + p] = t;
                }
            }
            __syncthreads();
            if (tid == 0) pivval_s = As[(size_t)c * n + c];
            __syncthreads();
            const double piv = pivval_s;
            if (piv == 0.0 || piv != piv) continue; // singular column: left as it is (info already set)
            for (int i = c + 1 + tid; i < n; i += T) As[(size_t)c * n + i] = As[(size_t)c * n + i] / piv;
            __syncthreads();
            const int rem = n - c - 1;
            for (int e = tid; e < rem * rem; e += T) {
                const int i = c + 1 + e % rem, q = c + 1 + e / rem;
                const double prod = As[(size_t)c * n + i] * As[(size_t)q * n + c];
                As[(size_t)q * n + i] = As[(size_t)q * n + i] - prod;
            }
            __syncthreads();
        }
    }
    if (tid == 0) pr.info[0] = info_s;
    __syncthreads();
    if (info_s == 0) { // (the substitutions skip flagged problems: trsm_left_kernel's skip_flag)
        // unit-lower, then upper: column-oriented substitution, every element sees its updates in the order of trsm_left_kernel
        for (int pass = 0; pass < 2; ++pass) {
            const bool lower = pass == 0;
            for (int step = 0; step < n; ++step) {
                const int kk = lower ? step : (n - 1 - step);
                const double* tk = As + (size_t)kk * n;
                if (!lower) {
                    const double dkk = tk[kk];
                    for (int c = tid; c < nrhs; c += T) Bs[(size_t)c * n + kk] = Bs[(size_t)c * n + kk] / dkk;
                    __syncthreads();
                }
                const int lo = lower ? kk + 1 : 0;
                const int cnt = lower ? (n - 1 - kk) : kk;
                for (int e = tid; e < cnt * nrhs; e += T) {
                    const int i = lo + e % cnt, c = e / cnt;
                    const double prod = tk[i] * Bs[(size_t)c * n + kk];
                    Bs[(size_t)c * n + i] = Bs[(size_t)c * n + i] - prod;
                }
                __syncthreads();
            }
        }
    }
    // ---- core[l, s, r] = X^T[r + ld (l S + s)] (tensorci2.rs:1167-1181); a numerically zero pivot matrix gives a zero core ----
    const bool zero = info_s == -1;
    for (size_t e = tid; e < total; e += T) {
        const int l = (int)(e % pk.L), s_ = (int)((e / pk.L) % pk.S), r = (int)(e / ((size_t)pk.L * pk.S));
        pk.core[e] = zero ? 0.0 : Bs[(size_t)(l * pk.S + s_) * pk.ld + r];
    }
}

} // namespace

void fill_small_launch(const FnDevice& fn, const PiJob* d_pis, const LuProblem* d_lups, const PackJob* d_packs, int n_jobs, int last_site,
                       hipStream_t stream)
{
    if (n_jobs <= 0) return;
    FillSmallArgs a;
    a.fn = fn;
    a.pis = d_pis;
    a.lups = d_lups;
    a.packs = d_packs;
    a.n_jobs = n_jobs;
    a.last_site = last_site;
    hipLaunchKernelGGL(fill_small_kernel, dim3(n_jobs), dim3(256), 0, stream, a);
}

void absmax_launch(const double* data, size_t count, unsigned long long* max_abs_bits, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, data, count, max_abs_bits);
}

} // namespace t4a
