// rook.hip — see rook.hpp.  The search is driven from the host (one small device-to-host read per visited
// column / row, exactly the places where the reference's rook_pivot looks at an argmax); all arithmetic — the
// residuals A[r,c] - A[r,J] (A[I,J]^{-1} A[I,c]), the partial-pivot LU of the pivot block, the triangular solves —
// runs on the device in the reference's operation order (solve_matrix then mat_mul, block_rook.rs:33-42).
#include "rook.hpp"

#include <algorithm>
#include <cmath>

namespace t4a {

namespace {

__global__ void __launch_bounds__(256) rook_gather_vec_kernel(const double* __restrict__ v, const int* __restrict__ idx,
                                                              int k, double* __restrict__ out)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) out[j] = v[idx[j]];
}

// b <- P b with the LAPACK-style pivot sequence of the LU (row j swapped with piv[j], j ascending)
__global__ void rook_apply_swaps_kernel(const int* __restrict__ piv, int k, double* b)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int j = 0; j < k; ++j) {
            const int p = piv[j];
            if (p != j) {
                const double t = b[j];
                b[j] = b[p];
                b[p] = t;
            }
        }
}

// block-wide "first strict maximum in ascending index order" (block_rook.rs:46-61); out: [0] index (as double,
// -1 when nothing beat the initial -1), [1] max |residual|
__device__ inline void block_argmax(double bv, int bi, double* out)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = sv[0] < 0.0 ? -1.0 : (double)si[0];
        out[1] = sv[0] < 0.0 ? 0.0 : sv[0];
    }
}

// residual of column `a_c` over the not-selected rows: a_c[i] - sum_j A[i, J[j]] * x[j]   (mat_mul order: j ascending)
__global__ void __launch_bounds__(256) rook_col_residual_kernel(const double* __restrict__ a_c,
                                                                const double* __restrict__ A, int M,
                                                                const int* __restrict__ J, int k,
                                                                const double* __restrict__ x,
                                                                const int* __restrict__ row_selected, double* out)
{
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        if (row_selected[i]) continue;
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            const double prod = A[(size_t)i + (size_t)M * J[j]] * x[j];
            acc = acc + prod;
        }
        const double r = k > 0 ? a_c[i] - acc : a_c[i];
        const double v = fabs(r);
        if (v > bv) {
            bv = v;
            bi = i;
        }
    }
    block_argmax(bv, bi, out);
}

// residual of row `a_r` over the not-selected columns: a_r[c] - sum_j y[j] * X[j, c]
__global__ void __launch_bounds__(256) rook_row_residual_kernel(const double* __restrict__ a_r,
                                                                const double* __restrict__ X, int N, int k,
                                                                const double* __restrict__ y,
                                                                const int* __restrict__ col_selected, double* out)
{
    double bv = -1.0;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        if (col_selected[c]) continue;
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            const double prod = y[j] * X[(size_t)j + (size_t)k * c];
            acc = acc + prod;
        }
        const double r = k > 0 ? a_r[c] - acc : a_r[c];
        const double v = fabs(r);
        if (v > bv) {
            bv = v;
            bi = c;
        }
    }
    block_argmax(bv, bi, out);
}

__global__ void rook_set_flag_kernel(int* flags, int idx) { flags[idx] = 1; }

// ------------------------------------------------------------------------------------------------
// Device-resident rook search (round 5; block_rook.rs:71-118 rook_pivot + :120-190 factorize_lazy): ONE persistent workgroup runs the
// whole pivot loop on a materialised matrix — per pivot the LU of the k x k pivot block and X = P^-1 A[I, :] (the reference recomputes
// both from scratch for every pivot), then the alternating column / row arg-max of the residual — and hands back the selection.  The
// arithmetic is that of the launch-per-visit path below, operation for operation: lu_kernel's right-looking LU with partial pivoting
// (first maximum of |a_ik|), trsm_left_kernel's column-oriented substitutions (k ascending / descending, separately rounded multiply
// and subtract), rook_col/row_residual_kernel's j-ascending sums, block_argmax's first strict maximum in ascending index order.
// ------------------------------------------------------------------------------------------------
struct RookDenseArgs {
    const double* A; // M x N, column-major
    int M, N, max_bond;
    double rel_tol, abs_tol;
    double* P;       // kcap x kcap
    double* X;       // kcap x N
    double* b;       // kcap
    double* y;       // kcap
    int* I;          // kcap selected rows
    int* J;          // kcap selected columns
    int* rowsel;     // M
    int* colsel;     // N
    int* seen;       // M + N visited flags
    int* piv;        // kcap
    double* dres;    // [0] last error [1] sampled max [2] evaluated entries [3 + k] accepted pivot errors
    int* ires;       // [0] rank [1] info [2] visits
    double* packed;  // everything the host reads, in one block: [0 .. kcap + 2] dres, then as ints [0..3] ires, [4 ..] I, [4 + kcap ..] J
    int kcap;
};

__device__ inline void rook_block_argmax(double bv, int bi, double* s_v, int* s_i, double* out_v, int* out_i)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    __syncthreads(); // (the scratch of the previous reduction has been read)
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    double v = s_v[0];
    int idx = s_i[0];
    for (int q = 1; q < nw; ++q)
        if (s_v[q] > v || (s_v[q] == v && s_i[q] < idx)) {
            v = s_v[q];
            idx = s_i[q];
        }
    *out_v = v;
    *out_i = idx;
}

// T x = b in place for ONE right-hand side, column-oriented like trsm_left_kernel (barrier per step)
__device__ inline void rook_trsm_vec(const double* T, int ldt, int n, bool lower, bool unit, double* b)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    for (int step = 0; step < n; ++step) {
        const int k = lower ? step : (n - 1 - step);
        const double* tk = T + (size_t)k * ldt;
        if (!unit) {
            if (tid == 0) b[k] = b[k] / tk[k];
            __syncthreads();
        }
        const int lo = lower ? k + 1 : 0;
        const int cnt = lower ? (n - 1 - k) : k;
        const double bk = b[k];
        for (int e = tid; e < cnt; e += NT) {
            const int i = lo + e;
            const double prod = tk[i] * bk;
            b[i] = b[i] - prod;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) rook_dense_kernel(RookDenseArgs a)
{
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int s_int[8];
    __shared__ double s_dbl[2];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int M = a.M, N = a.N;
    const double* A = a.A;
    for (int i = tid; i < M; i += NT) a.rowsel[i] = 0;
    for (int j = tid; j < N; j += NT) a.colsel[j] = 0;
    for (int e = tid; e < M + N; e += NT) a.seen[e] = 0;
    if (tid == 0) a.ires[1] = 0;
    __syncthreads();
    int k = 0, n_rows_seen = 0, n_cols_seen = 0, visits = 0;
    double max_error = 0.0, last_error = __builtin_nan(""), evals = 0.0;
    bool singular = false;
    while (k < a.max_bond) {
        // remaining_indices (:63-69): first unselected row / column and how many there are
        int fr = 0x7fffffff, fc = 0x7fffffff, nr = 0, nc = 0;
        for (int i = tid; i < M; i += NT)
            if (!a.rowsel[i]) {
                fr = i < fr ? i : fr;
                ++nr;
            }
        for (int j = tid; j < N; j += NT)
            if (!a.colsel[j]) {
                fc = j < fc ? j : fc;
                ++nc;
            }
        if (tid < 4) s_int[tid] = tid < 2 ? 0x7fffffff : 0;
        __syncthreads();
        atomicMin(&s_int[0], fr);
        atomicMin(&s_int[1], fc);
        atomicAdd(&s_int[2], nr);
        atomicAdd(&s_int[3], nc);
        __syncthreads();
        const int first_row = s_int[0], first_col = s_int[1], n_rem_rows = s_int[2], n_rem_cols = s_int[3];
        __syncthreads();
        if (n_rem_rows == 0 || n_rem_cols == 0) break;
        if (k > 0) {
            // factor_step: P = A[I, J] -> LU in place (row swaps applied to X as well); X = P^-1 A[I, :]
            for (int e = tid; e < k * k; e += NT) a.P[e] = A[(size_t)a.I[e % k] + (size_t)M * a.J[e / k]];
            for (long long e = tid; e < (long long)k * N; e += NT) a.X[e] = A[(size_t)a.I[e % k] + (size_t)M * (e / k)];
            __syncthreads();
            for (int c = 0; c < k; ++c) { // lu_kernel, one column per pass
                double bv = -1.0;
                int bi = 0x7fffffff;
                for (int i = c + tid; i < k; i += NT) {
                    const double v = fabs(a.P[(size_t)c * k + i]);
                    if (v > bv || (v == bv && i < bi)) {
                        bv = v;
                        bi = i;
                    }
                }
                double gv;
                int gi;
                rook_block_argmax(bv, bi, s_v, s_i, &gv, &gi);
                if (tid == 0) {
                    a.piv[c] = gi;
                    if (!(gv > 0.0) && a.ires[1] == 0) a.ires[1] = c + 1;
                }
                const int p = gi;
                if (p != c && p < k) {
                    for (int q = tid; q < k; q += NT) {
                        const double t = a.P[(size_t)q * k + c];
                        a.P[(size_t)q * k + c] = a.P[(size_t)q * k + p];
                        a.P[(size_t)q * k + p] = t;
                    }
                    for (int q = tid; q < N; q += NT) {
                        const double t = a.X[(size_t)q * k + c];
                        a.X[(size_t)q * k + c] = a.X[(size_t)q * k + p];
                        a.X[(size_t)q * k + p] = t;
                    }
                }
                __syncthreads();
                const double pv = a.P[(size_t)c * k + c];
                if (pv == 0.0 || pv != pv) {
                    __syncthreads();
                    continue;
                }
                __syncthreads();
                for (int i = c + 1 + tid; i < k; i += NT) a.P[(size_t)c * k + i] = a.P[(size_t)c * k + i] / pv;
                __syncthreads();
                const int rem = k - c - 1;
                for (int e = tid; e < rem * rem; e += NT) {
                    const int i = c + 1 + e % rem, q = c + 1 + e / rem;
                    const double prod = a.P[(size_t)c * k + i] * a.P[(size_t)q * k + c];
                    a.P[(size_t)q * k + i] = a.P[(size_t)q * k + i] - prod;
                }
                __syncthreads();
            }
            singular = a.ires[1] != 0;
            // L X' = P_swap A[I, :], then U X = X' (trsm_left_kernel over all N right-hand sides)
            for (int pass = 0; pass < 2; ++pass) {
                const bool lower = pass == 0;
                for (int step = 0; step < k; ++step) {
                    const int kk = lower ? step : (k - 1 - step);
                    const double* tk = a.P + (size_t)kk * k;
                    if (!lower) {
                        const double dkk = tk[kk];
                        for (int c = tid; c < N; c += NT) a.X[(size_t)c * k + kk] = a.X[(size_t)c * k + kk] / dkk;
                        __syncthreads();
                    }
                    const int lo = lower ? kk + 1 : 0;
                    const int cnt = lower ? (k - 1 - kk) : kk;
                    for (long long e = tid; e < (long long)cnt * N; e += NT) {
                        const int i = lo + (int)(e % cnt), c = (int)(e / cnt);
                        const double prod = tk[i] * a.X[(size_t)c * k + kk];
                        a.X[(size_t)c * k + i] = a.X[(size_t)c * k + i] - prod;
                    }
                    __syncthreads();
                }
            }
        }
        // rook_pivot (:71-118)
        int cur_col = first_col, cur_row = first_row;
        double pivot_abs = 0.0;
        const int max_steps = n_rem_rows + n_rem_cols + 1;
        for (int it = 0; it < max_steps; ++it) {
            if (singular) break;
            ++visits;
            if (!a.seen[M + cur_col]) { // (uniform: every thread reads the same flag)
                evals += (double)(M - n_rows_seen);
                ++n_cols_seen;
            }
            __syncthreads();
            if (tid == 0) a.seen[M + cur_col] = 1;
            const double* a_c = A + (size_t)M * cur_col;
            if (k > 0) {
                for (int j = tid; j < k; j += NT) a.b[j] = a_c[a.I[j]];
                __syncthreads();
                if (tid == 0)
                    for (int j = 0; j < k; ++j) {
                        const int p = a.piv[j];
                        if (p != j) {
                            const double t = a.b[j];
                            a.b[j] = a.b[p];
                            a.b[p] = t;
                        }
                    }
                __syncthreads();
                rook_trsm_vec(a.P, k, k, true, true, a.b);
                rook_trsm_vec(a.P, k, k, false, false, a.b);
            }
            {
                double bv = -1.0;
                int bi = 0x7fffffff;
                for (int i = tid; i < M; i += NT) {
                    if (a.rowsel[i]) continue;
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j) {
                        const double prod = A[(size_t)i + (size_t)M * a.J[j]] * a.b[j];
                        acc = acc + prod;
                    }
                    const double r = k > 0 ? a_c[i] - acc : a_c[i];
                    const double v = fabs(r);
                    if (v > bv) {
                        bv = v;
                        bi = i;
                    }
                }
                double gv;
                int gi;
                rook_block_argmax(bv, bi, s_v, s_i, &gv, &gi);
                cur_row = gv < 0.0 ? first_row : gi;
            }
            if (!a.seen[cur_row]) {
                evals += (double)(N - n_cols_seen);
                ++n_rows_seen;
            }
            __syncthreads();
            if (tid == 0) a.seen[cur_row] = 1;
            if (k > 0) {
                for (int j = tid; j < k; j += NT) a.y[j] = A[(size_t)cur_row + (size_t)M * a.J[j]];
                __syncthreads();
            }
            int next_col;
            {
                double bv = -1.0;
                int bi = 0x7fffffff;
                for (int c = tid; c < N; c += NT) {
                    if (a.colsel[c]) continue;
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j) {
                        const double prod = a.y[j] * a.X[(size_t)j + (size_t)k * c];
                        acc = acc + prod;
                    }
                    const double arc = A[(size_t)cur_row + (size_t)M * c];
                    const double r = k > 0 ? arc - acc : arc;
                    const double v = fabs(r);
                    if (v > bv) {
                        bv = v;
                        bi = c;
                    }
                }
                double gv;
                int gi;
                rook_block_argmax(bv, bi, s_v, s_i, &gv, &gi);
                next_col = gv < 0.0 ? first_col : gi;
                pivot_abs = gv < 0.0 ? 0.0 : gv;
            }
            if (next_col == cur_col) break;
            cur_col = next_col;
        }
        if (singular) break;
        // factorize_lazy stop rules (:158-176)
        last_error = pivot_abs;
        if (k > 0 && (pivot_abs < a.rel_tol * max_error || pivot_abs < a.abs_tol)) break;
        if (pivot_abs < 2.220446049250313e-16) break;
        max_error = fmax(max_error, pivot_abs);
        __syncthreads();
        if (tid == 0) {
            a.I[k] = cur_row;
            a.J[k] = cur_col;
            a.dres[3 + k] = pivot_abs;
            a.rowsel[cur_row] = 1;
            a.colsel[cur_col] = 1;
        }
        ++k;
        __syncthreads();
    }
    // what the lazy evaluator would have looked at: the visited rows and columns
    double mx = 0.0;
    for (long long e = tid; e < (long long)M * N; e += NT) {
        const int i = (int)(e % M), c = (int)(e / M);
        if (a.seen[i] || a.seen[M + c]) {
            const double v = A[e];
            const double av = sqrt(v * v);
            if (av > mx) mx = av;
        }
    }
    {
        double gv;
        int gi;
        rook_block_argmax(mx, tid, s_v, s_i, &gv, &gi);
        mx = gv;
    }
    if (tid == 0) {
        a.ires[0] = k;
        a.ires[2] = visits;
        a.dres[0] = last_error;
        a.dres[1] = mx;
        a.dres[2] = evals;
    }
    __syncthreads();
    if (a.packed) { // one block for the host (one copy into pinned memory instead of four into pageable vectors)
        int* const pi_ = reinterpret_cast<int*>(a.packed + a.kcap + 3);
        for (int e = tid; e < a.kcap + 3; e += NT) a.packed[e] = (e < 3 + k) ? a.dres[e] : 0.0;
        if (tid < 3) pi_[tid] = a.ires[tid];
        for (int e = tid; e < a.kcap; e += NT) {
            pi_[4 + e] = e < k ? a.I[e] : 0;
            pi_[4 + a.kcap + e] = e < k ? a.J[e] : 0;
        }
    }
    (void)s_dbl;
}

} // namespace

LuciResult rook_luci(Engine& eng, RookWork& w, const RookSource& src, const RrLUOptions& opts, double* sampled_max,
                     double* n_evaluated)
{
    const int M = src.M, N = src.N;
    hipStream_t st = eng.stream();
    LuciResult out;
    out.M = M;
    out.N = N;
    out.row_perm.resize(M);
    out.col_perm.resize(N);
    const int full_rank = std::min(M, N);
    if (full_rank == 0) { // block_rook.rs:127-134
        for (int i = 0; i < M; ++i) out.row_perm[i] = i;
        for (int j = 0; j < N; ++j) out.col_perm[j] = j;
        out.rank = 0;
        out.pivot_errors = {0.0};
        out.last_error = 0.0;
        out.has_factors = true;
        eng.reserve_factors(1, 1);
        return out;
    }
    const int max_bond = (int)std::min<size_t>(opts.max_bond_dim, (size_t)full_rank);
    const int kcap = std::max(max_bond, 1);

    w.A.reserve((size_t)M * N);
    w.At.reserve((size_t)M * N);
    w.P.reserve((size_t)kcap * kcap);
    w.X.reserve((size_t)kcap * std::max(M, N) + (size_t)std::max(M, N) * kcap);
    w.vec.reserve(2 * (size_t)kcap);
    w.res.reserve(4);
    w.I.reserve(kcap);
    w.J.reserve(kcap);
    w.rowsel.reserve(M);
    w.colsel.reserve(N);
    w.piv.reserve(kcap);
    w.info.reserve(1);
    w.maxbits.reserve(1);
    w.lup.reserve(1);
    w.trp.reserve(4);
    const bool device_search_planned = (bool)src.full && kcap <= 256 && (long long)M * N <= (1ll << 24);
    // the status word is ALWAYS cleared: a search that ends with rank 0 runs no LU, yet the final block reads the word — out of
    // recycled pool memory or a failed earlier call it was undefined
    T4A_HIP(hipMemsetAsync(w.info.get(), 0, sizeof(int), st));
    if (!device_search_planned) { // (the device-resident search clears its flags itself)
        T4A_HIP(hipMemsetAsync(w.rowsel.get(), 0, sizeof(int) * M, st));
        T4A_HIP(hipMemsetAsync(w.colsel.get(), 0, sizeof(int) * N, st));
        T4A_HIP(hipMemsetAsync(w.maxbits.get(), 0, sizeof(unsigned long long), st));
    }

    std::vector<char> col_seen(N, 0), row_seen(M, 0), row_sel(M, 0), col_sel(N, 0);
    size_t n_rows_seen = 0, n_cols_seen = 0;
    double evals = 0.0;
    auto visit_col = [&](int c) {
        if (col_seen[c]) return;
        src.column(c, w.A.get() + (size_t)M * c);
        absmax_launch(w.A.get() + (size_t)M * c, (size_t)M, w.maxbits.get(), st);
        col_seen[c] = 1;
        ++n_cols_seen;
        evals += (double)(M - (long)n_rows_seen); // entries not already cached through a visited row
    };
    auto visit_row = [&](int r) {
        if (row_seen[r]) return;
        src.row(r, w.At.get() + (size_t)N * r);
        absmax_launch(w.At.get() + (size_t)N * r, (size_t)N, w.maxbits.get(), st);
        row_seen[r] = 1;
        ++n_rows_seen;
        evals += (double)(N - (long)n_cols_seen);
    };

    std::vector<int> sel_rows, sel_cols;
    std::vector<double> accepted;
    double max_error = 0.0;
    double last_error = std::numeric_limits<double>::quiet_NaN();
    double hres[2];
    double* d_b = w.vec.get();
    double* d_y = w.vec.get() + kcap;
    TrsmProblem tp[4];
    LuProblem lp;

    auto upload_sel = [&]() {
        const int k = (int)sel_rows.size();
        if (k == 0) return;
        T4A_HIP(hipMemcpyAsync(w.I.get(), sel_rows.data(), sizeof(int) * k, hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(w.J.get(), sel_cols.data(), sizeof(int) * k, hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
    };
    // gather Rw = A[I, :] (k x N, ld k) out of the row cache At (N x M, row r contiguous)
    auto gather_rows = [&](double* dst, int k) {
        double* tmp = w.X.get() + (size_t)kcap * std::max(M, N); // N x k
        gather_launch(w.At.get(), N, nullptr, N, w.I.get(), k, tmp, N, st);
        transpose_launch(tmp, N, k, N, dst, k, st);
    };
    auto factor_step = [&](int k) { // P = A[I,J] -> LU in place; X = P^{-1} A[I,:]
        gather_launch(w.A.get(), M, w.I.get(), k, w.J.get(), k, w.P.get(), k, st);
        gather_rows(w.X.get(), k);
        lp.A = w.P.get();
        lp.lda = k;
        lp.n = k;
        lp.piv = w.piv.get();
        lp.info = w.info.get();
        lp.B = w.X.get();
        lp.ldb = k;
        lp.nrhs = N;
        lp.pmax_bits = nullptr;
        for (int q = 0; q < 4; ++q) {
            tp[q].T = w.P.get();
            tp[q].ldt = k;
            tp[q].n = k;
            tp[q].ldb = k;
            tp[q].skip_flag = nullptr;
        }
        tp[0].B = w.X.get(); // L X' = P_swap Rw
        tp[0].nrhs = N;
        tp[0].lower = 1;
        tp[0].unit_diag = 1;
        tp[1].B = w.X.get(); // U X = X'
        tp[1].nrhs = N;
        tp[1].lower = 0;
        tp[1].unit_diag = 0;
        tp[2].B = d_b;       // single right-hand side (column visits)
        tp[2].nrhs = 1;
        tp[2].lower = 1;
        tp[2].unit_diag = 1;
        tp[3].B = d_b;
        tp[3].nrhs = 1;
        tp[3].lower = 0;
        tp[3].unit_diag = 0;
        T4A_HIP(hipMemcpyAsync(w.lup.get(), &lp, sizeof(lp), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(w.trp.get(), tp, sizeof(tp), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st)); // lp / tp are pageable
        lu_batched_launch(w.lup.get(), 1, k, st);
        trsm_left_batched_launch(w.trp.get() + 0, 1, k, N, st);
        trsm_left_batched_launch(w.trp.get() + 1, 1, k, N, st);
    };
    auto read_result = [&](int fallback) -> std::pair<int, double> {
        int hinfo = 0;
        T4A_HIP(hipMemcpyAsync(hres, w.res.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipMemcpyAsync(&hinfo, w.info.get(), sizeof(int), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
        ++w.n_host_syncs;
        if (hinfo != 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "residual pivot solve failed: singular pivot matrix in the rook search");
        const int idx = hres[0] < 0.0 ? fallback : (int)hres[0];
        return {idx, hres[1]};
    };

    // ---- device-resident search (round 5): sources that can put the whole matrix into device memory run the pivot loop as ONE
    // persistent launch; the host reads the selection back with a single synchronisation (the launch-per-visit loop below costs two
    // per visited column / row pair: 54 ms for BASELINE configs[1] against 0.4 ms on one CPU core, tools/bench_components.py)
    bool device_search = device_search_planned;
    double device_mx = 0.0;
    if (device_search) {
        src.full(w.A.get());
        transpose_launch(w.A.get(), M, N, M, w.At.get(), N, st); // (the row cache of the factor builders below)
        w.dres.reserve((size_t)kcap + 4);
        w.ires.reserve(4);
        w.seen.reserve((size_t)M + N);
        RookDenseArgs ka;
        ka.A = w.A.get();
        ka.M = M;
        ka.N = N;
        ka.max_bond = max_bond;
        ka.rel_tol = opts.rel_tol;
        ka.abs_tol = opts.abs_tol;
        ka.P = w.P.get();
        ka.X = w.X.get();
        ka.b = w.vec.get();
        ka.y = w.vec.get() + kcap;
        ka.I = w.I.get();
        ka.J = w.J.get();
        ka.rowsel = w.rowsel.get();
        ka.colsel = w.colsel.get();
        ka.seen = w.seen.get();
        ka.piv = w.piv.get();
        ka.dres = w.dres.get();
        ka.ires = w.ires.get();
        const size_t packed_doubles = (size_t)kcap + 3 + ((size_t)4 + 2 * (size_t)kcap + 1) / 2;
        w.packed.reserve(packed_doubles);
        w.hpacked.reserve(packed_doubles);
        ka.packed = w.packed.get();
        ka.kcap = kcap;
        hipLaunchKernelGGL(rook_dense_kernel, dim3(1), dim3(1024), 0, st, ka);
        T4A_HIP(hipMemcpyAsync(w.hpacked.get(), w.packed.get(), packed_doubles * sizeof(double), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
        const double* const hd = w.hpacked.get();
        const int* const hpi = reinterpret_cast<const int*>(hd + kcap + 3);
        const int hi[3] = {hpi[0], hpi[1], hpi[2]};
        const int* const hI = hpi + 4;
        const int* const hJ = hpi + 4 + kcap;
        T4A_HIP(hipGetLastError());
        ++w.n_device_searches;
        w.n_device_visits += (size_t)hi[2];
        ++w.n_host_syncs;
        if (hi[1] != 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "residual pivot solve failed: singular pivot matrix in the rook search");
        const int rank_d = hi[0];
        for (int q = 0; q < rank_d; ++q) {
            sel_rows.push_back(hI[q]);
            sel_cols.push_back(hJ[q]);
            accepted.push_back(hd[3 + (size_t)q]);
            row_sel[hI[q]] = 1;
            col_sel[hJ[q]] = 1;
        }
        last_error = hd[0];
        device_mx = hd[1];
        evals = hd[2];
    } else {
        ++w.n_host_searches;
    }
    while (!device_search && (int)sel_rows.size() < max_bond) {
        const int k = (int)sel_rows.size();
        // remaining_indices (:63-69): ascending, selected ones skipped
        int first_row = -1, first_col = -1, n_rem_rows = 0, n_rem_cols = 0;
        for (int i = 0; i < M; ++i)
            if (!row_sel[i]) {
                if (first_row < 0) first_row = i;
                ++n_rem_rows;
            }
        for (int j = 0; j < N; ++j)
            if (!col_sel[j]) {
                if (first_col < 0) first_col = j;
                ++n_rem_cols;
            }
        if (n_rem_rows == 0 || n_rem_cols == 0) break;
        if (k > 0) factor_step(k);

        // rook_pivot (:71-118)
        int cur_col = first_col, cur_row = first_row;
        double pivot_abs = 0.0;
        const int max_steps = n_rem_rows + n_rem_cols + 1;
        for (int it = 0; it < max_steps; ++it) {
            visit_col(cur_col);
            const double* a_c = w.A.get() + (size_t)M * cur_col;
            if (k > 0) {
                hipLaunchKernelGGL(rook_gather_vec_kernel, dim3(1), dim3(256), 0, st, a_c, w.I.get(), k, d_b);
                hipLaunchKernelGGL(rook_apply_swaps_kernel, dim3(1), dim3(64), 0, st, w.piv.get(), k, d_b);
                trsm_left_batched_launch(w.trp.get() + 2, 1, k, 1, st);
                trsm_left_batched_launch(w.trp.get() + 3, 1, k, 1, st);
            }
            hipLaunchKernelGGL(rook_col_residual_kernel, dim3(1), dim3(256), 0, st, a_c, w.A.get(), M, w.J.get(), k, d_b,
                               w.rowsel.get(), w.res.get());
            cur_row = read_result(first_row).first;

            visit_row(cur_row);
            const double* a_r = w.At.get() + (size_t)N * cur_row;
            if (k > 0) hipLaunchKernelGGL(rook_gather_vec_kernel, dim3(1), dim3(256), 0, st, a_r, w.J.get(), k, d_y);
            hipLaunchKernelGGL(rook_row_residual_kernel, dim3(1), dim3(256), 0, st, a_r, w.X.get(), N, k, d_y,
                               w.colsel.get(), w.res.get());
            const auto rr = read_result(first_col);
            pivot_abs = rr.second;
            const int next_col = rr.first;
            if (next_col == cur_col) break;
            cur_col = next_col; // also the answer of the fall-through branch (:108-117): same row residual again
        }
        T4A_HIP(hipGetLastError());

        // factorize_lazy stop rules (:158-176)
        last_error = pivot_abs;
        if (k > 0 && (pivot_abs < opts.rel_tol * max_error || pivot_abs < opts.abs_tol)) break;
        if (pivot_abs < 2.220446049250313e-16) break;
        max_error = std::fmax(max_error, pivot_abs);
        sel_rows.push_back(cur_row);
        sel_cols.push_back(cur_col);
        accepted.push_back(pivot_abs);
        row_sel[cur_row] = 1;
        col_sel[cur_col] = 1;
        hipLaunchKernelGGL(rook_set_flag_kernel, dim3(1), dim3(1), 0, st, w.rowsel.get(), cur_row);
        hipLaunchKernelGGL(rook_set_flag_kernel, dim3(1), dim3(1), 0, st, w.colsel.get(), cur_col);
        visit_col(cur_col); // both are already cached: the selected row / column were the last ones visited
        visit_row(cur_row);
        upload_sel();
    }

    const int rank = (int)sel_rows.size();
    if (rank >= full_rank)
        last_error = 0.0;
    else if (rank == max_bond && rank > 0)
        last_error = accepted[rank - 1];
    accepted.push_back(last_error);
    out.rank = rank;
    out.pivot_errors = accepted;
    out.last_error = last_error;
    {
        int p = 0;
        for (int r : sel_rows) out.row_perm[p++] = r;
        for (int i = 0; i < M; ++i)
            if (!row_sel[i]) out.row_perm[p++] = i;
        p = 0;
        for (int c : sel_cols) out.col_perm[p++] = c;
        for (int j = 0; j < N; ++j)
            if (!col_sel[j]) out.col_perm[p++] = j;
    }

    // the LU / triangular-solve descriptors of the factor build travel through pinned memory (two copies out of stack variables and a stream
    // synchronisation before: ~40 us per bond); the staging block is free again behind the synchronisation that ends every search
    auto upload_descriptors = [&](const LuProblem& lp_, const TrsmProblem* tp_) {
        w.hdesc.reserve(sizeof(LuProblem) + 2 * sizeof(TrsmProblem));
        char* const hdp = w.hdesc.get();
        std::memcpy(hdp, &lp_, sizeof(LuProblem));
        std::memcpy(hdp + sizeof(LuProblem), tp_, 2 * sizeof(TrsmProblem));
        T4A_HIP(hipMemcpyAsync(w.lup.get(), hdp, sizeof(LuProblem), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(w.trp.get(), hdp + sizeof(LuProblem), 2 * sizeof(TrsmProblem), hipMemcpyHostToDevice, st));
    };
    // CrossFactors + factors_to_public (factors.rs:58-101, matrix_luci.rs:109-135)
    eng.reserve_factors((size_t)M * std::max(rank, 1), (size_t)std::max(rank, 1) * N);
    if (rank > 0) {
        const int k = rank;
        gather_launch(w.A.get(), M, w.I.get(), k, w.J.get(), k, w.P.get(), k, st); // pivot
        if (opts.left_orthogonal) {
            // left = (P^T \ C^T)^T with C = A[:, J]; right = A[I, :]
            double* Pt = w.X.get();                       // k x k
            double* Ct = w.X.get() + (size_t)kcap * kcap; // k x M
            transpose_launch(w.P.get(), k, k, k, Pt, k, st);
            double* C = eng.left(); // stage C (M x k) in the output buffer, transpose, solve, transpose back
            gather_launch(w.A.get(), M, nullptr, M, w.J.get(), k, C, M, st);
            transpose_launch(C, M, k, M, Ct, k, st);
            lp.A = Pt;
            lp.lda = k;
            lp.n = k;
            lp.piv = w.piv.get();
            lp.info = w.info.get();
            lp.B = Ct;
            lp.ldb = k;
            lp.nrhs = M;
            lp.pmax_bits = nullptr;
            tp[0].T = Pt;
            tp[0].ldt = k;
            tp[0].n = k;
            tp[0].B = Ct;
            tp[0].ldb = k;
            tp[0].nrhs = M;
            tp[0].lower = 1;
            tp[0].unit_diag = 1;
            tp[0].skip_flag = nullptr;
            tp[1] = tp[0];
            tp[1].lower = 0;
            tp[1].unit_diag = 0;
            upload_descriptors(lp, tp);
            lu_batched_launch(w.lup.get(), 1, k, st);
            trsm_left_batched_launch(w.trp.get() + 0, 1, k, M, st);
            trsm_left_batched_launch(w.trp.get() + 1, 1, k, M, st);
            transpose_launch(Ct, k, M, k, eng.left(), M, st);
            gather_rows(eng.right(), k);
        } else {
            // left = A[:, J]; right = P \ A[I, :]
            gather_launch(w.A.get(), M, nullptr, M, w.J.get(), k, eng.left(), M, st);
            gather_rows(eng.right(), k);
            lp.A = w.P.get();
            lp.lda = k;
            lp.n = k;
            lp.piv = w.piv.get();
            lp.info = w.info.get();
            lp.B = eng.right();
            lp.ldb = k;
            lp.nrhs = N;
            lp.pmax_bits = nullptr;
            tp[0].T = w.P.get();
            tp[0].ldt = k;
            tp[0].n = k;
            tp[0].B = eng.right();
            tp[0].ldb = k;
            tp[0].nrhs = N;
            tp[0].lower = 1;
            tp[0].unit_diag = 1;
            tp[0].skip_flag = nullptr;
            tp[1] = tp[0];
            tp[1].lower = 0;
            tp[1].unit_diag = 0;
            upload_descriptors(lp, tp);
            lu_batched_launch(w.lup.get(), 1, k, st);
            trsm_left_batched_launch(w.trp.get() + 0, 1, k, N, st);
            trsm_left_batched_launch(w.trp.get() + 1, 1, k, N, st);
        }
    }
    unsigned long long bits = 0;
    int hinfo = 0;
    {   // (into pinned memory; the sampled maximum of a device-resident search came back with its result block)
        w.hfin.reserve(2);
        unsigned long long* const hf = w.hfin.get();
        hf[0] = hf[1] = 0ull;
        if (!device_search) T4A_HIP(hipMemcpyAsync(hf, w.maxbits.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipMemcpyAsync(hf + 1, w.info.get(), sizeof(int), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
        bits = hf[0];
        std::memcpy(&hinfo, hf + 1, sizeof(int));
    }
    T4A_HIP(hipGetLastError());
    if (hinfo != 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "factor solve failed: singular pivot matrix");
    double mx;
    std::memcpy(&mx, &bits, sizeof(mx));
    if (device_search) mx = device_mx; // (the visited rows / columns, found by the search kernel itself)
    out.abs_max = mx;
    if (sampled_max && mx > *sampled_max) *sampled_max = mx;
    if (n_evaluated) *n_evaluated += evals;
    out.has_factors = true;
    return out;
}

} // namespace t4a
