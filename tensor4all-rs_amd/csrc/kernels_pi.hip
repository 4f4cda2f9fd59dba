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

void absmax_launch(const double* data, size_t count, unsigned long long* max_abs_bits, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, data, count, max_abs_bits);
}

} // namespace t4a
