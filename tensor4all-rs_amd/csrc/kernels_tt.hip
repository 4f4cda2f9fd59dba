// kernels_tt.hip — tensor-train kernels next to the TCI2 sweep (SURVEY.md §8 rows a15–a18):
//   core <-> left/right matrix reshapes   <- tensor3_to_left_matrix / tensor3_to_right_matrix / split_indices
//                                            (simplett/src/compression.rs:127-161, tensorci/src/conversion.rs:273-351)
//   sum, norm2                             <- AbstractTensorTrain::sum / norm2 (simplett/src/traits.rs:231-354)
//   left / right environments + dots       <- TTCache::evaluate_left / evaluate_right / evaluate_many
//                                            (simplett/src/cache.rs:430-688, einsum_helper.rs:192-268)
// The chain contractions keep the reference's summation order (index ascending, separately rounded multiply and
// add; built with -ffp-contract=off), so their results are bit-identical to the CPU oracle's.
#include "kernels.hpp"

namespace t4a {

namespace {

// mode 0: core -> left matrix  (row l*S+s, col r)      1: left matrix -> core
// mode 2: core -> right matrix (row l, col s*R+r)      3: right matrix -> core
__global__ void __launch_bounds__(256) core_reshape_kernel(const double* __restrict__ in, int L, int S, int R, int mode,
                                                           double* __restrict__ out)
{
    const size_t total = (size_t)L * S * R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        // e enumerates the OUTPUT linearly
        size_t src;
        if (mode == 0) { // out = left matrix: e = (l*S+s) + L*S*r
            const size_t r = e / ((size_t)L * S), row = e % ((size_t)L * S);
            const size_t l = row / S, s = row % S;
            src = l + (size_t)L * (s + (size_t)S * r);
        } else if (mode == 1) { // out = core: e = l + L*(s + S*r)
            const size_t l = e % L, sr = e / L;
            const size_t s = sr % S, r = sr / S;
            src = (l * S + s) + (size_t)L * S * r;
        } else if (mode == 2) { // out = right matrix: e = l + L*(s*R + r)
            const size_t l = e % L, c = e / L;
            const size_t s = c / R, r = c % R;
            src = l + (size_t)L * (s + (size_t)S * r);
        } else { // out = core from right matrix
            const size_t l = e % L, sr = e / L;
            const size_t s = sr % S, r = sr / S;
            src = l + (size_t)L * (s * (size_t)R + r);
        }
        out[e] = in[src];
    }
}

// sum over all indices (traits.rs:231-275): cur[r] = sum_l cur[l] * (sum_s T[l,s,r]), single workgroup.
__global__ void __launch_bounds__(256) tt_sum_kernel(const TtCoreDesc* cores, int n_sites, int max_bond, double* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cur = (double*)smem_raw;
    double* nxt = cur + max_bond;
    const int tid = threadIdx.x, T = blockDim.x;
    {
        const TtCoreDesc c0 = cores[0];
        for (int r = tid; r < c0.r; r += T) {
            double acc = 0.0;
            for (int s = 0; s < c0.d; ++s) acc = acc + c0.data[(size_t)c0.l * ((size_t)s + (size_t)c0.d * r)];
            cur[r] = acc;
        }
    }
    __syncthreads();
    for (int site = 1; site < n_sites; ++site) {
        const TtCoreDesc c = cores[site];
        for (int r = tid; r < c.r; r += T) {
            double sum = 0.0;
            for (int l = 0; l < c.l; ++l) {
                double ss = 0.0;
                for (int s = 0; s < c.d; ++s) ss = ss + c.data[(size_t)l + (size_t)c.l * ((size_t)s + (size_t)c.d * r)];
                const double prod = cur[l] * ss;
                sum = sum + prod;
            }
            nxt[r] = sum;
        }
        __syncthreads();
        double* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (tid == 0) out[0] = cur[0];
}

// norm2 transfer step (traits.rs:314-347): nxt[ra*R+rc] = sum_{la,lc,s} (cur[la*L+lc] * T[la,s,ra]) * T[lc,s,rc]
// in the reference's loop order.  first != 0: cur is implicitly the 1x1 identity... the first site has its own
// form nxt[ra*R+rc] = sum_s T[0,s,ra]*T[0,s,rc] (traits.rs:298-309).
__global__ void __launch_bounds__(256) tt_norm2_step_kernel(TtCoreDesc c, const double* __restrict__ cur, int first,
                                                            double* __restrict__ nxt)
{
    const int R = c.r, L = c.l, S = c.d;
    const size_t total = (size_t)R * R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ra = (int)(e / R), rc = (int)(e % R);
        const double* ta = c.data + (size_t)L * S * ra;
        const double* tc = c.data + (size_t)L * S * rc;
        double acc = 0.0;
        if (first) {
            for (int s = 0; s < S; ++s) {
                const double prod = ta[(size_t)L * s] * tc[(size_t)L * s];
                acc = acc + prod;
            }
        } else {
            for (int la = 0; la < L; ++la)
                for (int lc = 0; lc < L; ++lc) {
                    const double cv = cur[(size_t)la * L + lc];
                    for (int s = 0; s < S; ++s) {
                        const double p1 = cv * ta[la + (size_t)L * s];
                        const double p2 = p1 * tc[lc + (size_t)L * s];
                        acc = acc + p2;
                    }
                }
        }
        nxt[e] = acc;
    }
}

// Left environments (cache.rs:430-467): one workgroup per prefix; env <- env * T_k[:, i_k, :] for k < split.
// idx: n_items x split (uint32, item-major); out: n_items x ld.
__global__ void __launch_bounds__(256) tt_env_left_kernel(const TtCoreDesc* cores, int split, int max_bond,
                                                          const uint32_t* __restrict__ idx, int n_items, double* out,
                                                          int ld)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cur = (double*)smem_raw;
    double* nxt = cur + max_bond;
    const int tid = threadIdx.x, T = blockDim.x;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t* my = idx + (size_t)it * split;
        {
            const TtCoreDesc c0 = cores[0];
            for (int r = tid; r < c0.r; r += T) cur[r] = c0.data[(size_t)c0.l * ((size_t)my[0] + (size_t)c0.d * r)];
        }
        __syncthreads();
        for (int s = 1; s < split; ++s) {
            const TtCoreDesc c = cores[s];
            for (int r = tid; r < c.r; r += T) {
                const double* col = c.data + (size_t)c.l * ((size_t)my[s] + (size_t)c.d * r);
                double sum = 0.0;
                for (int l = 0; l < c.l; ++l) {
                    const double prod = cur[l] * col[l];
                    sum = sum + prod;
                }
                nxt[r] = sum;
            }
            __syncthreads();
            double* t = cur;
            cur = nxt;
            nxt = t;
        }
        const int rl = cores[split - 1].r;
        for (int r = tid; r < rl; r += T) out[(size_t)it * ld + r] = cur[r];
        __syncthreads();
    }
}

// Right environments (cache.rs:469-518): env <- T_k[:, i_k, :] * env from the last site down to `split`.
// idx: n_items x (n_sites - split); out: n_items x ld.
__global__ void __launch_bounds__(256) tt_env_right_kernel(const TtCoreDesc* cores, int n_sites, int split, int max_bond,
                                                           const uint32_t* __restrict__ idx, int n_items, double* out,
                                                           int ld)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cur = (double*)smem_raw;
    double* nxt = cur + max_bond;
    const int tid = threadIdx.x, T = blockDim.x;
    const int w = n_sites - split;
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t* my = idx + (size_t)it * w;
        {
            const TtCoreDesc cl = cores[n_sites - 1];
            // slice of the last site: (l, r = 0..cl.r), the chain requires cl.r == 1
            for (int l = tid; l < cl.l; l += T) cur[l] = cl.data[(size_t)l + (size_t)cl.l * (size_t)my[w - 1]];
        }
        __syncthreads();
        for (int s = n_sites - 2; s >= split; --s) {
            const TtCoreDesc c = cores[s];
            const uint32_t is = my[s - split];
            for (int l = tid; l < c.l; l += T) {
                double sum = 0.0;
                for (int r = 0; r < c.r; ++r) {
                    const double prod = c.data[(size_t)l + (size_t)c.l * ((size_t)is + (size_t)c.d * r)] * cur[r];
                    sum = sum + prod;
                }
                nxt[l] = sum;
            }
            __syncthreads();
            double* t = cur;
            cur = nxt;
            nxt = t;
        }
        const int ll = cores[split].l;
        for (int l = tid; l < ll; l += T) out[(size_t)it * ld + l] = cur[l];
        __syncthreads();
    }
}

// out[p] = sum_i left[il[p]][i] * right[ir[p]][i]  (cache.rs:668-685)
__global__ void __launch_bounds__(256) tt_env_dot_kernel(const double* __restrict__ left, const double* __restrict__ right,
                                                         int len, int ld, const uint32_t* __restrict__ il,
                                                         const uint32_t* __restrict__ ir, size_t n_pts, double* out)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pts; p += (size_t)gridDim.x * blockDim.x) {
        const double* a = left + (size_t)il[p] * ld;
        const double* b = right + (size_t)ir[p] * ld;
        double acc = 0.0;
        for (int i = 0; i < len; ++i) {
            const double prod = a[i] * b[i];
            acc = acc + prod;
        }
        out[p] = acc;
    }
}

} // namespace

void core_reshape_launch(const double* in, int L, int S, int R, int mode, double* out, hipStream_t stream)
{
    const size_t total = (size_t)L * S * R;
    if (total == 0) return;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(core_reshape_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, L, S, R, mode, out);
}

void tt_sum_launch(const TtCoreDesc* d_cores, int n_sites, int max_bond, double* d_out, hipStream_t stream)
{
    const size_t lds = (size_t)2 * (max_bond > 0 ? max_bond : 1) * 8;
    hipLaunchKernelGGL(tt_sum_kernel, dim3(1), dim3(256), lds, stream, d_cores, n_sites, max_bond, d_out);
}

void tt_norm2_step_launch(const TtCoreDesc& core, const double* d_cur, bool first, double* d_nxt, hipStream_t stream)
{
    const size_t total = (size_t)core.r * core.r;
    if (total == 0) return;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(tt_norm2_step_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, core, d_cur, first ? 1 : 0,
                       d_nxt);
}

void tt_env_left_launch(const TtCoreDesc* d_cores, int split, int max_bond, const uint32_t* d_idx, int n_items,
                        double* d_out, int ld, hipStream_t stream)
{
    if (n_items <= 0) return;
    const int blocks = n_items < 8192 ? n_items : 8192;
    const size_t lds = (size_t)2 * (max_bond > 0 ? max_bond : 1) * 8;
    hipLaunchKernelGGL(tt_env_left_kernel, dim3(blocks), dim3(256), lds, stream, d_cores, split, max_bond, d_idx,
                       n_items, d_out, ld);
}

void tt_env_right_launch(const TtCoreDesc* d_cores, int n_sites, int split, int max_bond, const uint32_t* d_idx,
                         int n_items, double* d_out, int ld, hipStream_t stream)
{
    if (n_items <= 0) return;
    const int blocks = n_items < 8192 ? n_items : 8192;
    const size_t lds = (size_t)2 * (max_bond > 0 ? max_bond : 1) * 8;
    hipLaunchKernelGGL(tt_env_right_kernel, dim3(blocks), dim3(256), lds, stream, d_cores, n_sites, split, max_bond,
                       d_idx, n_items, d_out, ld);
}

void tt_env_dot_launch(const double* d_left, const double* d_right, int len, int ld, const uint32_t* d_il,
                       const uint32_t* d_ir, size_t n_pts, double* d_out, hipStream_t stream)
{
    if (n_pts == 0) return;
    size_t blocks = (n_pts + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tt_env_dot_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_left, d_right, len, ld, d_il,
                       d_ir, n_pts, d_out);
}

} // namespace t4a
