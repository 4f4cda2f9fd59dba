// kernels_rrlu_global.hip — full-pivot rank-revealing LU for matrices that fit neither the register-resident kernel
// (kernels_rrlu_reg.hip) nor the LDS-resident one (kernels_rrlu.hip): very tall / very large candidate matrices, e.g.
// the d*chi*chi x d*chi matrices at a branching vertex of a tree (treetci/src/proposer.rs:57-88).
//
// The matrix stays in HBM and is never physically permuted: rowpos/colpos map a physical row / column to its position
// in the reference's swapped buffer, so "first strict maximum in column-major order" (matrixlu.rs:480-519) becomes
// "largest v*v, ties to the smallest (colpos, rowpos)" (64-bit position keys: no 65535 limit here).  Three launches per pivot step:
//   argmax  — every workgroup scans a slice of the trailing submatrix; the last one to finish reduces the partial
//             winners, applies the stop rules (matrixlu.rs:757-791) and swaps the positions;
//   scale   — the pivot column (left-orthogonal, :562-577) or pivot row (:579-591) divided by the pivot;
//   update  — t - x*y on the trailing submatrix with separately rounded multiply and subtract (:593-612).
// Every launch returns immediately once the stop flag is set, so the host enqueues max_steps steps without reading back.
// HBM-bound: 16 bytes per trailing element and step, exactly the algorithmic traffic of the reference loop.
#include "kernels.hpp"

#include <cstdlib>

namespace t4a {

namespace {

struct GCand {
    double sc, val;
    unsigned long long pos; // colpos << 32 | rowpos
    int pi, pj;             // physical row / column
};
constexpr unsigned long long G_NOPOS = ~0ull;

__device__ __forceinline__ bool g_beats(double sa, unsigned long long pa, double sb, unsigned long long pb)
{
    return (sa > sb) || (sa == sb && pa < pb);
}

// a NaN square only wins when it sits on the first scanned element (the reference seeds its maximum with it)
__device__ __forceinline__ double g_score(double a, unsigned long long pos, int k)
{
    double sc = a * a;
    if (sc != sc) {
        const unsigned long long diag = ((unsigned long long)k << 32) | (unsigned long long)k;
        sc = (pos == diag) ? __builtin_huge_val() : -1.0;
    }
    return sc;
}

__device__ __forceinline__ GCand g_block_reduce(GCand c, GCand* red)
{
    for (int off = 32; off >= 1; off >>= 1) {
        GCand o;
        o.sc = __shfl_xor(c.sc, off);
        o.val = __shfl_xor(c.val, off);
        o.pos = (unsigned long long)__shfl_xor((long long)c.pos, off);
        o.pi = __shfl_xor(c.pi, off);
        o.pj = __shfl_xor(c.pj, off);
        if (g_beats(o.sc, o.pos, c.sc, c.pos)) c = o;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) red[wave] = c;
    __syncthreads();
    GCand b = red[0];
    for (int q = 1; q < nw; ++q)
        if (g_beats(red[q].sc, red[q].pos, b.sc, b.pos)) b = red[q];
    __syncthreads();
    return b;
}

__global__ void __launch_bounds__(256) rg_init_kernel(RrluGlobalArgs p)
{
    const size_t total = (size_t)p.M * (size_t)p.N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double amax = 0.0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const double v = p.A[e];
        p.W[e] = v;
        const double av = sqrt(v * v);
        if (av > amax) amax = av;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)p.M; i += stride) {
        p.rowpos[i] = (int)i;
        p.posrow[i] = (int)i;
    }
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < (size_t)p.N; j += stride) {
        p.colpos[j] = (int)j;
        p.poscol[j] = (int)j;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(amax, off);
        if (o > amax) amax = o;
    }
    if ((threadIdx.x & 63) == 0 && amax > 0.0)
        atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(amax));
    if (blockIdx.x == 0 && threadIdx.x == 0) { // dstate: [0] pivot [1] max_error [2] lu.error (NaN until the first search)
        p.dstate[0] = 0.0;
        p.dstate[1] = 0.0;
        p.dstate[2] = __builtin_nan("");
    }
}

__global__ void __launch_bounds__(256) rg_argmax_kernel(RrluGlobalArgs p, int k)
{
    __shared__ GCand red[4];
    __shared__ int s_last;
    if (p.istate[1] != 0) return; // stopped at an earlier step
    const int tid = threadIdx.x;
    const size_t total = (size_t)p.M * (size_t)p.N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    GCand best;
    best.sc = -2.0;
    best.val = 0.0;
    best.pos = G_NOPOS;
    best.pi = 0;
    best.pj = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + tid; e < total; e += stride) {
        const int j = (int)(e / (size_t)p.M);
        const int i = (int)(e - (size_t)j * (size_t)p.M);
        const int cp = p.colpos[j];
        const int rp = p.rowpos[i];
        if (cp < k || rp < k) continue;
        const double a = p.W[e];
        const unsigned long long pos = ((unsigned long long)cp << 32) | (unsigned long long)rp;
        const double sc = g_score(a, pos, k);
        if (g_beats(sc, pos, best.sc, best.pos)) {
            best.sc = sc;
            best.val = a;
            best.pos = pos;
            best.pi = i;
            best.pj = j;
        }
    }
    best = g_block_reduce(best, red);
    if (tid == 0) {
        p.partials_sc[blockIdx.x] = best.sc;
        p.partials_val[blockIdx.x] = best.val;
        p.partials_pos[blockIdx.x] = best.pos;
        p.partials_ij[2 * blockIdx.x] = best.pi;
        p.partials_ij[2 * blockIdx.x + 1] = best.pj;
        __threadfence();
        const int ticket = atomicAdd(&p.istate[0], 1);
        s_last = (ticket == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // ---- last workgroup: global winner, stop rules, position swap, pivot column / row scaling ----
    GCand g;
    g.sc = -2.0;
    g.val = 0.0;
    g.pos = G_NOPOS;
    g.pi = 0;
    g.pj = 0;
    for (int b = tid; b < (int)gridDim.x; b += blockDim.x) {
        const double sc = p.partials_sc[b];
        const unsigned long long pos = p.partials_pos[b];
        if (g_beats(sc, pos, g.sc, g.pos)) {
            g.sc = sc;
            g.val = p.partials_val[b];
            g.pos = pos;
            g.pi = p.partials_ij[2 * b];
            g.pj = p.partials_ij[2 * b + 1];
        }
    }
    g = g_block_reduce(g, red);
    if (tid == 0) {
        const double pivot_abs = sqrt(g.val * g.val);
        const double max_error = p.dstate[1];
        p.dstate[2] = pivot_abs; // lu.error (:758)
        int stop = 0;
        if (k > 0 && (pivot_abs < p.rel_tol * max_error || pivot_abs < p.abs_tol)) stop = 1;
        const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
        if (!stop && pivot_abs <= min_pivot_abs) stop = 1;
        if (!stop) {
            p.dstate[1] = fmax(max_error, pivot_abs);
            p.dstate[0] = g.val;
            const int prp = (int)(g.pos & 0xFFFFFFFFull), pcp = (int)(g.pos >> 32);
            const int pr = g.pi, pc = g.pj;
            const int rk = p.posrow[k];
            p.posrow[k] = pr;
            p.posrow[prp] = rk;
            p.rowpos[rk] = prp;
            p.rowpos[pr] = k;
            const int ck = p.poscol[k];
            p.poscol[k] = pc;
            p.poscol[pcp] = ck;
            p.colpos[ck] = pcp;
            p.colpos[pc] = k;
            p.istate[2] = pr;
            p.istate[3] = pc;
            p.istate[4] = k + 1; // npivots
            p.pivot_vals[k] = g.val;
        } else {
            p.istate[1] = 1;
        }
        p.istate[0] = 0; // ticket counter for the next step
        __threadfence();
    }
}

// scale_column_tail (left-orthogonal, matrixlu.rs:562-577) or scale_row_tail (:579-591) of the step decided by the search
__global__ void __launch_bounds__(256) rg_scale_kernel(RrluGlobalArgs p, int k)
{
    if (p.istate[1] != 0) return;
    const int pr = p.istate[2], pc = p.istate[3];
    const double pivot = p.dstate[0];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (p.left_orth) {
        if (gid < p.M && p.rowpos[gid] > k) {
            double* q = p.W + (size_t)pc * (size_t)p.M + gid;
            *q = *q / pivot;
        }
    } else {
        if (gid < p.N && p.colpos[gid] > k) {
            double* q = p.W + (size_t)gid * (size_t)p.M + pr;
            *q = *q / pivot;
        }
    }
}

__global__ void __launch_bounds__(256) rg_update_kernel(RrluGlobalArgs p, int k)
{
    if (p.istate[1] != 0) return;
    const int pr = p.istate[2], pc = p.istate[3];
    const size_t total = (size_t)p.M * (size_t)p.N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double* x = p.W + (size_t)pc * (size_t)p.M;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e / (size_t)p.M);
        const int i = (int)(e - (size_t)j * (size_t)p.M);
        if (p.colpos[j] <= k || p.rowpos[i] <= k) continue;
        const double y = p.W[(size_t)j * (size_t)p.M + pr];
        const double prod = x[i] * y;
        p.W[e] = p.W[e] - prod;
    }
}

__global__ void __launch_bounds__(256) rg_final_kernel(RrluGlobalArgs p)
{
    const int npiv = p.istate[4];
    const size_t total = (size_t)p.M * (size_t)p.N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0) {
        double error = p.dstate[2];
        if (npiv >= (p.M < p.N ? p.M : p.N)) error = 0.0; // matrixlu.rs:811-813
        p.iresult[0] = npiv;
        p.dresult[0] = error;
    }
    for (size_t i = gid; i < (size_t)p.M; i += stride) p.row_perm[i] = p.posrow[i];
    for (size_t j = gid; j < (size_t)p.N; j += stride) p.col_perm[j] = p.poscol[j];
    int nan_seen = 0;
    for (size_t e = gid; e < total; e += stride) {
        const int cp = (int)(e / (size_t)p.M);
        const int rp = (int)(e - (size_t)cp * (size_t)p.M);
        const double v = p.W[(size_t)p.poscol[cp] * (size_t)p.M + p.posrow[rp]];
        const bool in_l = (cp < npiv) && (rp >= cp);
        const bool in_u = (rp < npiv) && (cp >= rp);
        if ((in_l || in_u) && v != v) nan_seen = 1;
        if (p.Aout) p.Aout[e] = v;
    }
    if (nan_seen) atomicExch(&p.iresult[2], 1);
}

} // namespace

size_t rrlu_global_int_words(int M, int N, int blocks) { return 2 * (size_t)M + 2 * (size_t)N + 2 * (size_t)blocks + 16; }
size_t rrlu_global_double_words(int M, int N, int blocks) { return (size_t)M * (size_t)N + 3 * (size_t)blocks + 8; }

int rrlu_global_blocks(int M, int N)
{
    const size_t total = (size_t)M * (size_t)N;
    size_t b = (total + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

void rrlu_global_launch(RrluGlobalArgs a, int* iwork, double* dwork, hipStream_t stream)
{
    const int blocks = rrlu_global_blocks(a.M, a.N);
    a.rowpos = iwork;
    a.colpos = a.rowpos + a.M;
    a.posrow = a.colpos + a.N;
    a.poscol = a.posrow + a.M;
    a.partials_ij = a.poscol + a.N;
    a.istate = a.partials_ij + 2 * (size_t)blocks;
    a.W = dwork;
    a.partials_sc = a.W + (size_t)a.M * (size_t)a.N;
    a.partials_val = a.partials_sc + blocks;
    a.partials_pos = reinterpret_cast<unsigned long long*>(a.partials_val + blocks);
    a.dstate = a.partials_val + 2 * (size_t)blocks;
    (void)hipMemsetAsync(a.istate, 0, 16 * sizeof(int), stream);
    hipLaunchKernelGGL(rg_init_kernel, dim3(blocks), dim3(256), 0, stream, a);
    // the search ends with one same-address atomic per workgroup (they serialise in L2, ~40 ns each): a moderate grid (512) is the measured optimum
    static const int acap = diag_env("T4A_RG_ABLOCKS") ? std::atoi(diag_env("T4A_RG_ABLOCKS")) : 512;
    const int ablocks = blocks < acap ? blocks : acap;
    for (int k = 0; k < a.max_steps; ++k) {
        hipLaunchKernelGGL(rg_argmax_kernel, dim3(ablocks), dim3(256), 0, stream, a, k);
        hipLaunchKernelGGL(rg_scale_kernel, dim3(((a.left_orth ? a.M : a.N) + 255) / 256), dim3(256), 0, stream, a, k);
        if (k + 1 < a.M && k + 1 < a.N) hipLaunchKernelGGL(rg_update_kernel, dim3(blocks), dim3(256), 0, stream, a, k);
    }
    hipLaunchKernelGGL(rg_final_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

} // namespace t4a
