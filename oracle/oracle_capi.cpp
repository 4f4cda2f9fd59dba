// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY (see t4a_oracle.hpp header).
// Plain C entry points so that tests/ and bench.py's cpu_baseline leg can drive the CPU
// restatement through ctypes.  Nothing in the product (tensor4all-rs_amd/) links this.
#include "t4a_oracle.hpp"
#if defined(_OPENMP)
#include <omp.h>
#endif
#include "t4a_oracle_patch.hpp"
#include "t4a_oracle_tree.hpp"
#include "t4a_oracle_quantics.hpp"

#include "../include/t4a_testfunctions.h"

#include <chrono>
#include <cstring>
#include <memory>

using namespace t4a_oracle;

namespace {
thread_local std::string g_last_error;

template <class F> int guarded(F&& body)
{
    try {
        body();
        return 0;
    } catch (const OracleError& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return ERR_INTERNAL;
    }
}

// Built-in function: integer table-sum front end + t4a_fn_value (include/t4a_testfunctions.h).
struct BuiltinFn {
    int fid = 0, n_acc = 0;
    double params[T4A_FN_MAX_PARAMS] = {0};
    std::vector<uint64_t> weights; // n_acc * total
    std::vector<size_t> offset;    // per site
    size_t total = 0;
    double operator()(const MultiIndex& idx) const
    {
        uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
        for (size_t s = 0; s < idx.size(); ++s)
            for (int k = 0; k < n_acc; ++k) acc[k] += weights[(size_t)k * total + offset[s] + idx[s]];
        return t4a_fn_value(fid, acc, params);
    }
};

typedef double (*scalar_cb_t)(void* ctx, const uint64_t* idx, uint64_t n_sites);
typedef int (*batch_cb_t)(void* ctx, const uint64_t* idx /* n_sites x n_pts col-major */, uint64_t n_sites,
                          uint64_t n_pts, double* out);

struct OracleTci {
    std::unique_ptr<TensorCI2> tci;
    ScalarFn f;
    BatchFn batched;
    bool has_batched = false;
    OptimizationResult last;
    double last_seconds = 0.0;
    int pivot_search = 0; // PivotSearchStrategy applied to every subsequent call
    std::vector<size_t> fn_dims; // function-only holder (adaptive driver): site dimensions without a TensorCI2
    const std::vector<size_t>& dims() const { return tci ? tci->local_dims : fn_dims; }
};

TCI2Options make_options(double tolerance, uint64_t max_iter, uint64_t max_bond_dim, int normalize_error,
                         uint64_t max_nglobal_pivot, uint64_t nsearch, int sweep_strategy, uint64_t ncheck_history,
                         int strictly_nested, double tol_margin, int has_seed, uint64_t seed)
{
    TCI2Options o;
    o.tolerance = tolerance;
    o.max_iter = (size_t)max_iter;
    o.max_bond_dim = (size_t)max_bond_dim;
    o.normalize_error = normalize_error != 0;
    o.max_nglobal_pivot = (size_t)max_nglobal_pivot;
    o.nsearch = (size_t)nsearch;
    o.sweep_strategy = (Sweep2Strategy)sweep_strategy;
    o.ncheck_history = (size_t)ncheck_history;
    o.strictly_nested = strictly_nested != 0;
    o.tol_margin_global_search = tol_margin;
    o.has_seed = has_seed != 0;
    o.seed = seed;
    return o;
}
} // namespace

extern "C" {

const char* oracle_last_error() { return g_last_error.c_str(); }

// ---- the reference's random stream (t4a_oracle_rng.hpp) ----
int oracle_stdrng_sample(uint64_t seed, const uint64_t* dims, uint64_t n, uint64_t* out)
{
    return guarded([&] {
        OracleStdRng rng(seed);
        for (uint64_t i = 0; i < n; ++i) out[i] = (uint64_t)rng.range((size_t)dims[i]);
    });
}
int oracle_stdrng_words(uint64_t seed, uint64_t n_u32, uint32_t* out32, uint64_t n_u64, uint64_t* out64)
{
    // n_u32 words through next_u32, then n_u64 through next_u64 (exercises the BlockRng buffer edge when n_u32 is odd)
    return guarded([&] {
        OracleStdRng rng(seed);
        for (uint64_t i = 0; i < n_u32; ++i) out32[i] = rng.next_u32();
        for (uint64_t i = 0; i < n_u64; ++i) out64[i] = rng.next_u64();
    });
}
int oracle_chacha_block(const uint8_t* key32, uint64_t counter, uint64_t stream, int rounds, uint32_t* out16)
{
    return guarded([&] {
        std::array<uint8_t, 32> k{};
        for (size_t i = 0; i < 32; ++i) k[i] = key32[i];
        const auto b = OracleStdRng::chacha_block(k, counter, stream, (unsigned)rounds / 2);
        for (size_t i = 0; i < 16; ++i) out16[i] = b[i];
    });
}

// ---- the other two random streams (t4a_oracle_rng2.hpp) ----
int oracle_siphash(const uint8_t* msg, uint64_t len, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds, uint64_t* out)
{
    return guarded([&] { *out = siphash(std::vector<uint8_t>(msg, msg + len), k0, k1, c_rounds, d_rounds); });
}
int oracle_smallrng_words(uint64_t seed, const uint64_t* state4, uint64_t n, uint64_t* out)
{
    return guarded([&] {
        OracleSmallRng rng(seed);
        if (state4)
            for (int i = 0; i < 4; ++i) rng.s[i] = state4[i];
        for (uint64_t i = 0; i < n; ++i) out[i] = rng.next_u64();
    });
}
int oracle_smallrng_sample(uint64_t seed, const uint64_t* dims, uint64_t n, uint64_t* out)
{
    return guarded([&] {
        OracleSmallRng rng(seed);
        for (uint64_t i = 0; i < n; ++i) out[i] = (uint64_t)rng.range((size_t)dims[i]);
    });
}
int oracle_smallrng_shuffle(uint64_t seed, uint64_t n, uint64_t* out)
{
    return guarded([&] {
        std::vector<uint64_t> v(n);
        for (uint64_t i = 0; i < n; ++i) v[i] = i;
        OracleSmallRng rng(seed);
        rng2_shuffle(v, rng);
        for (uint64_t i = 0; i < n; ++i) out[i] = v[i];
    });
}
int oracle_tree_edge_seed(uint64_t seed, const char* tag, uint64_t u, uint64_t v, uint64_t history_len, uint64_t ni, uint64_t nj, uint64_t* out)
{
    return guarded([&] {
        HashBytes h;
        h.u64(seed);
        h.str(tag);
        h.u64(std::min(u, v));
        h.u64(std::max(u, v));
        h.u64(history_len);
        h.u64(ni);
        h.u64(nj);
        *out = h.default_hasher_finish();
    });
}
int oracle_chacha8_standard_normal(uint64_t seed, uint64_t n, double* out, uint64_t n_words, uint32_t* out_words)
{
    return guarded([&] {
        if (n) {
            OracleChaCha8Rng rng(seed);
            for (uint64_t i = 0; i < n; ++i) out[i] = oracle_ziggurat().normal(rng);
        }
        if (n_words) {
            OracleChaCha8Rng rng(seed);
            for (uint64_t i = 0; i < n_words; ++i) out_words[i] = rng.next_u32();
        }
    });
}
int oracle_chacha8_block(const uint32_t* key8, uint64_t counter, uint32_t* out16)
{
    return guarded([&] { OracleChaCha8Rng::block(key8, counter, 8, out16); });
}

// ---- dense kernels ----
int oracle_rrlu_f64(double* a_inout, uint64_t m, uint64_t n, uint64_t max_bond_dim, double rel_tol, double abs_tol,
                    int left_orthogonal, uint64_t* row_perm, uint64_t* col_perm, uint64_t* npivots, double* last_error)
{
    return guarded([&] {
        Matrix a(m, n, a_inout);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        // copy the factored buffer back even when NaN is reported (the reference mutates in place)
        RrLU lu;
        try {
            lu = rrlu_mut(a, o);
        } catch (...) {
            std::memcpy(a_inout, a.a.data(), sizeof(double) * m * n);
            throw;
        }
        std::memcpy(a_inout, a.a.data(), sizeof(double) * m * n);
        for (size_t i = 0; i < m; ++i) row_perm[i] = lu.row_permutation[i];
        for (size_t i = 0; i < n; ++i) col_perm[i] = lu.col_permutation[i];
        *npivots = lu.n_pivot;
        *last_error = lu.error;
    });
}

// left: m x rank, right: rank x n (buffers sized for rank = min(m,n)); pivot_errors: min(m,n)+1
int oracle_luci_f64(const double* a, uint64_t m, uint64_t n, uint64_t max_bond_dim, double rel_tol, double abs_tol,
                    int left_orthogonal, uint64_t* rank, uint64_t* rows, uint64_t* cols, double* pivot_errors,
                    double* left, double* right)
{
    return guarded([&] {
        Matrix am(m, n, a);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        MatrixLuciFactors f = matrix_luci_factors_from_matrix(am, o);
        *rank = f.rank;
        for (size_t i = 0; i < f.rank; ++i) {
            rows[i] = f.row_indices[i];
            cols[i] = f.col_indices[i];
        }
        for (size_t i = 0; i < f.pivot_errors.size(); ++i) pivot_errors[i] = f.pivot_errors[i];
        std::memcpy(left, f.left.a.data(), sizeof(double) * f.left.a.size());
        std::memcpy(right, f.right.a.data(), sizeof(double) * f.right.a.size());
    });
}

int oracle_gemm_f64(const double* a, const double* b, uint64_t m, uint64_t k, uint64_t n, double* c)
{
    return guarded([&] {
        Matrix z = mat_mul(Matrix(m, k, a), Matrix(k, n, b));
        std::memcpy(c, z.a.data(), sizeof(double) * m * n);
    });
}

int oracle_trsm_f64(const double* a, uint64_t na, const double* b, uint64_t bm, uint64_t bn, int left_side, int lower,
                    int transpose_a, int unit_diagonal, double* x)
{
    return guarded([&] {
        Matrix r = triangular_solve(Matrix(na, na, a), Matrix(bm, bn, b), left_side != 0, lower != 0, transpose_a != 0,
                                    unit_diagonal != 0);
        std::memcpy(x, r.a.data(), sizeof(double) * bm * bn);
    });
}

int oracle_solve_f64(const double* a, uint64_t n, const double* b, uint64_t nrhs, double* x)
{
    return guarded([&] {
        Matrix r = solve(Matrix(n, n, a), Matrix(n, nrhs, b));
        std::memcpy(x, r.a.data(), sizeof(double) * n * nrhs);
    });
}

int oracle_convergence_criterion(const uint64_t* ranks, const double* errors, const uint64_t* nglobal, uint64_t len,
                                 double tolerance, uint64_t max_bond_dim, uint64_t ncheck_history, int* result)
{
    return guarded([&] {
        std::vector<size_t> r(ranks, ranks + len), g(nglobal, nglobal + len);
        std::vector<double> e(errors, errors + len);
        Termination t;
        const bool done = convergence_criterion(
            r, e, g, tolerance, max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim,
            (size_t)ncheck_history, t);
        *result = done ? (int)t : -1;
    });
}

// ---- built-in function evaluation (checks host/device bit-equality of the workload) ----
int oracle_fn_eval(int fid, int n_acc, const double* params, const uint64_t* weights, const uint64_t* local_dims,
                   uint64_t n_sites, const uint64_t* idx /* n_sites x n_pts col-major */, uint64_t n_pts, double* out)
{
    return guarded([&] {
        BuiltinFn fn;
        fn.fid = fid;
        fn.n_acc = n_acc;
        std::memcpy(fn.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
        fn.offset.resize(n_sites);
        size_t tot = 0;
        for (size_t s = 0; s < n_sites; ++s) {
            fn.offset[s] = tot;
            tot += local_dims[s];
        }
        fn.total = tot;
        fn.weights.assign(weights, weights + (size_t)n_acc * tot);
        MultiIndex mi(n_sites);
        for (size_t p = 0; p < n_pts; ++p) {
            for (size_t s = 0; s < n_sites; ++s) mi[s] = idx[s + n_sites * p];
            out[p] = fn(mi);
        }
    });
}

// ---- TCI2 handle ----
void* oracle_tci2_new(const uint64_t* local_dims, uint64_t n_sites)
{
    void* h = nullptr;
    guarded([&] {
        auto* o = new OracleTci();
        std::vector<size_t> d(local_dims, local_dims + n_sites);
        o->tci.reset(new TensorCI2(d));
        h = o;
    });
    return h;
}

// function holder without a TensorCI2 (any number of sites >= 1): input of oracle_adaptive_interpolate
void* oracle_fn_new(const uint64_t* local_dims, uint64_t n_sites)
{
    auto* o = new OracleTci();
    o->fn_dims.assign(local_dims, local_dims + n_sites);
    return o;
}

void oracle_tci2_release(void* h) { delete static_cast<OracleTci*>(h); }

int oracle_tci2_set_builtin_fn(void* h, int fid, int n_acc, const double* params, const uint64_t* weights)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        BuiltinFn fn;
        fn.fid = fid;
        fn.n_acc = n_acc;
        std::memcpy(fn.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
        const auto& d = o->dims();
        fn.offset.resize(d.size());
        size_t tot = 0;
        for (size_t s = 0; s < d.size(); ++s) {
            fn.offset[s] = tot;
            tot += d[s];
        }
        fn.total = tot;
        fn.weights.assign(weights, weights + (size_t)n_acc * tot);
        o->f = fn;
        o->has_batched = false;
#if defined(_OPENMP)
        if (o->tci) o->tci->parallel_eval = true; // (a function holder made by oracle_fn_new has no TCI state)  built-in functions are pure: candidate matrices and fill sites may use all host threads
#endif
    });
}

// 0: scalar build; otherwise the number of OpenMP threads the `native` build uses
int oracle_openmp_threads(void)
{
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 0;
#endif
}
void oracle_set_threads(int n)
{
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_tci2_set_callback(void* h, scalar_cb_t cb, batch_cb_t bcb, void* ctx)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        o->f = [cb, ctx](const MultiIndex& idx) {
            std::vector<uint64_t> v(idx.begin(), idx.end());
            return cb(ctx, v.data(), v.size());
        };
        o->has_batched = bcb != nullptr;
        if (bcb) {
            o->batched = [bcb, ctx](const std::vector<MultiIndex>& pts) {
                const size_t ns = pts.empty() ? 0 : pts[0].size();
                std::vector<uint64_t> flat(ns * pts.size());
                for (size_t p = 0; p < pts.size(); ++p)
                    for (size_t s = 0; s < ns; ++s) flat[s + ns * p] = pts[p][s];
                std::vector<double> out(pts.size());
                int produced = bcb(ctx, flat.data(), ns, pts.size(), out.data());
                if (produced >= 0 && (size_t)produced != pts.size()) out.resize((size_t)produced);
                return out;
            };
        }
    });
}

int oracle_tci2_add_global_pivots(void* h, const uint64_t* pivots /* n_sites x n col-major */, uint64_t n)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        const size_t ns = o->tci->len();
        std::vector<MultiIndex> p(n, MultiIndex(ns));
        for (size_t k = 0; k < n; ++k)
            for (size_t s = 0; s < ns; ++s) p[k][s] = pivots[s + ns * k];
        o->tci->add_global_pivots(p);
    });
}

int oracle_tci2_crossinterpolate2(void* h, const uint64_t* pivots, uint64_t npivots, double tolerance, uint64_t max_iter,
                                  uint64_t max_bond_dim, int normalize_error, uint64_t max_nglobal_pivot,
                                  uint64_t nsearch, int sweep_strategy, uint64_t ncheck_history, int strictly_nested,
                                  double tol_margin, int has_seed, uint64_t seed)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        const size_t ns = o->tci->len();
        std::vector<MultiIndex> p(npivots, MultiIndex(ns));
        for (size_t k = 0; k < npivots; ++k)
            for (size_t s = 0; s < ns; ++s) p[k][s] = pivots[s + ns * k];
        TCI2Options opt = make_options(tolerance, max_iter, max_bond_dim, normalize_error, max_nglobal_pivot, nsearch,
                                       sweep_strategy, ncheck_history, strictly_nested, tol_margin, has_seed, seed);
        auto t0 = std::chrono::steady_clock::now();
        opt.pivot_search = (PivotSearchStrategy)o->pivot_search;
        o->last = crossinterpolate2(*o->tci, o->f, o->has_batched ? &o->batched : nullptr, p, opt);
        o->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
}

// Run `n_iter` iterations of the optimize_with_finder loop on the current state (no final 1-site sweep).
int oracle_tci2_optimize(void* h, double tolerance, uint64_t max_iter, uint64_t max_bond_dim, int normalize_error,
                         uint64_t max_nglobal_pivot, uint64_t nsearch, int sweep_strategy, uint64_t ncheck_history,
                         int strictly_nested, double tol_margin, int has_seed, uint64_t seed, int final_sweep1site)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        TCI2Options opt = make_options(tolerance, max_iter, max_bond_dim, normalize_error, max_nglobal_pivot, nsearch,
                                       sweep_strategy, ncheck_history, strictly_nested, tol_margin, has_seed, seed);
        auto t0 = std::chrono::steady_clock::now();
        opt.pivot_search = (PivotSearchStrategy)o->pivot_search;
        o->last = optimize(*o->tci, o->f, o->has_batched ? &o->batched : nullptr, opt, final_sweep1site != 0);
        o->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
}

int oracle_tci2_sweep2site(void* h, int forward, double tolerance, uint64_t max_bond_dim)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        TCI2Options opt;
        opt.tolerance = tolerance;
        opt.max_bond_dim = (size_t)max_bond_dim;
        opt.pivot_search = (PivotSearchStrategy)o->pivot_search;
        auto t0 = std::chrono::steady_clock::now();
        o->tci->sweep2site(o->f, o->has_batched ? &o->batched : nullptr, forward != 0, opt);
        o->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
}

int oracle_tci2_sweep1site(void* h, int forward, double rel_tol, double abs_tol, uint64_t max_bond_dim,
                           int update_tensors)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        o->tci->sweep1site(o->f, forward != 0, rel_tol, abs_tol,
                           max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim,
                           update_tensors != 0);
    });
}

// TensorCI2::make_canonical (tensorci2.rs:1201-1221); max_bond_dim == 0 stands for usize::MAX
int oracle_tci2_make_canonical(void* h, double rel_tol, double abs_tol, uint64_t max_bond_dim)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        o->tci->make_canonical(o->f, rel_tol, abs_tol, max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim);
    });
}

int oracle_tci2_fill_site_tensors(void* h)
{
    return guarded([&] { static_cast<OracleTci*>(h)->tci->fill_site_tensors(static_cast<OracleTci*>(h)->f); });
}

double oracle_tci2_last_seconds(void* h) { return static_cast<OracleTci*>(h)->last_seconds; }
uint64_t oracle_tci2_n_evals(void* h) { return static_cast<OracleTci*>(h)->tci->n_evals; }
uint64_t oracle_tci2_rank(void* h) { return static_cast<OracleTci*>(h)->tci->rank(); }
double oracle_tci2_max_sample_value(void* h) { return static_cast<OracleTci*>(h)->tci->max_sample_value; }
int oracle_tci2_termination(void* h) { return (int)static_cast<OracleTci*>(h)->last.termination; }
uint64_t oracle_tci2_n_iterations(void* h) { return static_cast<OracleTci*>(h)->last.errors.size(); }

int oracle_tci2_history(void* h, uint64_t* ranks, double* errors)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        for (size_t i = 0; i < o->last.errors.size(); ++i) {
            ranks[i] = o->last.ranks[i];
            errors[i] = o->last.errors[i];
        }
    });
}

// which: 0 = I set, 1 = J set.  Query count with out == NULL.
int oracle_tci2_index_set(void* h, int which, uint64_t site, uint64_t* count, uint64_t* width, uint64_t* out)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        const auto& set = which == 0 ? o->tci->i_set[site] : o->tci->j_set[site];
        const size_t w = which == 0 ? site : o->tci->len() - site - 1;
        *count = set.size();
        *width = w;
        if (out)
            for (size_t k = 0; k < set.size(); ++k)
                for (size_t s = 0; s < w; ++s) out[s + w * k] = set[k][s];
    });
}

int oracle_tci2_site_tensor(void* h, uint64_t site, uint64_t* dims3, double* out)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        const Tensor3& t = o->tci->site_tensors[site];
        dims3[0] = t.l;
        dims3[1] = t.s;
        dims3[2] = t.r;
        if (out) std::memcpy(out, t.d.data(), sizeof(double) * t.d.size());
    });
}

int oracle_tci2_bond_errors(void* h, double* out)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        for (size_t i = 0; i < o->tci->bond_errors.size(); ++i) out[i] = o->tci->bond_errors[i];
    });
}

int oracle_tci2_pivot_errors(void* h, uint64_t* count, double* out)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        *count = o->tci->pivot_errors.size();
        if (out)
            for (size_t i = 0; i < o->tci->pivot_errors.size(); ++i) out[i] = o->tci->pivot_errors[i];
    });
}

// (M, N, rank) per bond of the most recent 2-site half sweep
int oracle_tci2_last_sweep_shapes(void* h, uint64_t* out /* 3 x (n-1) */)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        for (size_t b = 0; b < o->tci->last_sweep_shapes.size(); ++b)
            for (int k = 0; k < 3; ++k) out[3 * b + k] = o->tci->last_sweep_shapes[b][k];
    });
}

int oracle_tci2_evaluate(void* h, const uint64_t* idx /* n_sites x n_pts col-major */, uint64_t n_pts, double* out)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        SimpleTensorTrain tt = o->tci->to_tensor_train();
        const size_t ns = o->tci->len();
        MultiIndex mi(ns);
        for (size_t p = 0; p < n_pts; ++p) {
            for (size_t s = 0; s < ns; ++s) mi[s] = idx[s + ns * p];
            out[p] = tt.evaluate(mi);
        }
    });
}

int oracle_tci2_sum(void* h, double* out)
{
    return guarded([&] { *out = static_cast<OracleTci*>(h)->tci->to_tensor_train().sum(); });
}

// Replace the I/J sets (resume format == `TensorCI2::from_index_sets`, tensorci2.rs:551-582).
int oracle_tci2_set_index_set(void* h, int which, uint64_t site, uint64_t count, const uint64_t* data)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        const size_t w = which == 0 ? site : o->tci->len() - site - 1;
        std::vector<MultiIndex> set(count, MultiIndex(w));
        for (size_t k = 0; k < count; ++k)
            for (size_t s = 0; s < w; ++s) set[k][s] = data[s + w * k];
        (which == 0 ? o->tci->i_set[site] : o->tci->j_set[site]) = set;
    });
}

int oracle_tci2_set_pivot_search(void* h, int strategy)
{
    return guarded([&] {
        if (strategy < 0 || strategy > 1) throw OracleError(ERR_INVALID_ARGUMENT, "invalid pivot_search");
        static_cast<OracleTci*>(h)->pivot_search = strategy;
    });
}

// lazy_matrix_luci_factors_from_blocks on a dense column-major matrix (block source = gather); also reports the
// largest block (rows x cols) the kernel asked for.
int oracle_luci_rook_f64(const double* a, uint64_t m, uint64_t n, uint64_t max_bond_dim, double rel_tol, double abs_tol,
                         int left_orthogonal, uint64_t* rank, uint64_t* rows, uint64_t* cols, double* pivot_errors,
                         double* left, double* right, uint64_t* max_block)
{
    return guarded([&] {
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        size_t biggest = 0;
        BlockFn src = [&](const std::vector<size_t>& r, const std::vector<size_t>& c, double* out) {
            biggest = std::max(biggest, r.size() * c.size());
            for (size_t j = 0; j < c.size(); ++j)
                for (size_t i = 0; i < r.size(); ++i) out[i + r.size() * j] = a[r[i] + m * c[j]];
        };
        // the factor gathers (full pivot rows/columns) are not part of the pivot search
        PivotSelectionCore sel = rook::factorize_lazy(m, n, src, o);
        if (max_block) *max_block = biggest;
        MatrixLuciFactors f = lazy_matrix_luci_factors_from_blocks(m, n, src, o);
        (void)sel;
        *rank = f.rank;
        for (size_t i = 0; i < f.rank; ++i) {
            rows[i] = f.row_indices[i];
            cols[i] = f.col_indices[i];
        }
        for (size_t i = 0; i < f.pivot_errors.size(); ++i) pivot_errors[i] = f.pivot_errors[i];
        if (!f.left.a.empty()) std::memcpy(left, f.left.a.data(), f.left.a.size() * sizeof(double));
        if (!f.right.a.empty()) std::memcpy(right, f.right.a.data(), f.right.a.size() * sizeof(double));
    });
}

int oracle_tci2_set_max_sample_value(void* h, double v)
{
    return guarded([&] { static_cast<OracleTci*>(h)->tci->max_sample_value = v; });
}

int oracle_tci2_clear_history(void* h)
{
    return guarded([&] {
        auto* o = static_cast<OracleTci*>(h);
        o->tci->i_set_history.clear();
        o->tci->j_set_history.clear();
    });
}


// ---- adaptive patching driver (partitionedtt::adaptiveinterpolate) ----
struct OraclePtt {
    std::vector<SubDomainTT> patches;
    size_t n_sites = 0;
};

// Uses the function (and dims) attached to the TCI2 handle `fn_handle`.  patch_order may be null (natural order).
void* oracle_adaptive_interpolate(void* fn_handle, const uint64_t* pivots, uint64_t npivots, double tolerance,
                                  uint64_t max_iter, uint64_t max_bond_dim, int normalize_error,
                                  uint64_t max_nglobal_pivot, uint64_t nsearch, int sweep_strategy, uint64_t ncheck_history,
                                  int strictly_nested, double tol_margin, int has_seed, uint64_t seed,
                                  const uint64_t* patch_order, uint64_t n_initial_pivots, int recycle_pivots)
{
    void* out = nullptr;
    guarded([&] {
        auto* o = static_cast<OracleTci*>(fn_handle);
        const size_t ns = o->dims().size();
        AdaptiveInterpolateOptions ao;
        ao.tci_options = make_options(tolerance, max_iter, max_bond_dim, normalize_error, max_nglobal_pivot, nsearch,
                                      sweep_strategy, ncheck_history, strictly_nested, tol_margin, has_seed, seed);
        ao.tci_options.pivot_search = (PivotSearchStrategy)o->pivot_search;
        if (patch_order) ao.patch_order.assign(patch_order, patch_order + ns);
        ao.n_initial_pivots = (size_t)n_initial_pivots;
        ao.recycle_pivots = recycle_pivots != 0;
        std::vector<MultiIndex> p(npivots, MultiIndex(ns));
        for (size_t k = 0; k < npivots; ++k)
            for (size_t s = 0; s < ns; ++s) p[k][s] = pivots[s + ns * k];
        auto* r = new OraclePtt;
        r->n_sites = ns;
        try {
            r->patches = adaptiveinterpolate(o->f, o->has_batched ? &o->batched : nullptr, o->dims(), p, ao);
        } catch (...) {
            delete r;
            throw;
        }
        out = r;
    });
    return out;
}
void oracle_ptt_release(void* h) { delete static_cast<OraclePtt*>(h); }
uint64_t oracle_ptt_len(void* h) { return static_cast<OraclePtt*>(h)->patches.size(); }
int oracle_ptt_projector(void* h, uint64_t k, uint64_t* count, uint64_t* positions, uint64_t* values)
{
    return guarded([&] {
        const auto& pr = static_cast<OraclePtt*>(h)->patches.at(k).projector;
        *count = pr.size();
        size_t i = 0;
        for (const auto& kv : pr) {
            if (positions) positions[i] = kv.first;
            if (values) values[i] = kv.second;
            ++i;
        }
    });
}
int oracle_ptt_site_tensor(void* h, uint64_t k, uint64_t site, uint64_t* dims3, double* out)
{
    return guarded([&] {
        const auto& t = static_cast<OraclePtt*>(h)->patches.at(k).tt.tensors.at(site);
        dims3[0] = t.l;
        dims3[1] = t.s;
        dims3[2] = t.r;
        if (out && !t.d.empty()) std::memcpy(out, t.d.data(), t.d.size() * sizeof(double));
    });
}
int oracle_ptt_evaluate(void* h, const uint64_t* idx, uint64_t n_pts, double* out)
{
    return guarded([&] {
        auto* r = static_cast<OraclePtt*>(h);
        MultiIndex mi(r->n_sites);
        for (size_t p = 0; p < n_pts; ++p) {
            for (size_t s = 0; s < r->n_sites; ++s) mi[s] = idx[s + r->n_sites * p];
            double acc = 0.0;
            for (const auto& sd : r->patches) acc = acc + sd.tt.evaluate(mi);
            out[p] = acc;
        }
    });
}

// ---- TreeTCI (t4a_oracle_tree.hpp) ----
struct OracleTree {
    std::unique_ptr<TreeTCI2> st;
    TreeBatchFn eval;
    TreeNetwork net;
    bool has_net = false;
    TreeOptimizeResult last;
};

static TreeTciOptions make_tree_options(double tolerance, uint64_t max_iter, uint64_t max_bond_dim, int normalize_error,
                                        int enable_global_pivots, uint64_t nsearch, uint64_t max_nglobal_pivot,
                                        double tol_margin, int has_seed, uint64_t seed)
{
    TreeTciOptions o;
    o.tolerance = tolerance;
    o.max_iter = (size_t)max_iter;
    o.has_max_bond_dim = max_bond_dim != 0;
    o.max_bond_dim = (size_t)max_bond_dim;
    o.normalize_error = normalize_error != 0;
    o.enable_global_pivots = enable_global_pivots != 0;
    o.nsearch = (size_t)nsearch;
    o.max_nglobal_pivot = (size_t)max_nglobal_pivot;
    o.tol_margin_global_search = tol_margin;
    o.has_seed = has_seed != 0;
    o.seed = seed;
    return o;
}

// edges: 2 * n_edges site numbers; the function is taken from `fn_handle` (oracle_fn_new / oracle_tci2_new holder)
void* oracle_tree_new(void* fn_handle, const uint64_t* local_dims, uint64_t n_sites, const uint64_t* edges, uint64_t n_edges)
{
    void* out = nullptr;
    guarded([&] {
        std::vector<TreeEdge> es;
        for (size_t k = 0; k < n_edges; ++k) es.emplace_back((size_t)edges[2 * k], (size_t)edges[2 * k + 1]);
        TreeGraph g((size_t)n_sites, es);
        std::vector<size_t> d(local_dims, local_dims + n_sites);
        auto* t = new OracleTree();
        try {
            t->st.reset(new TreeTCI2(d, g));
        } catch (...) {
            delete t;
            throw;
        }
        if (fn_handle) t->eval = tree_batch_from_scalar(static_cast<OracleTci*>(fn_handle)->f);
        out = t;
    });
    return out;
}
void oracle_tree_release(void* h) { delete static_cast<OracleTree*>(h); }

int oracle_tree_add_global_pivots(void* h, const uint64_t* pivots, uint64_t n_pivots)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        const size_t n = t->st->local_dims.size();
        std::vector<MultiIndex> pv;
        for (size_t k = 0; k < n_pivots; ++k) pv.emplace_back(pivots + k * n, pivots + (k + 1) * n);
        t->st->add_global_pivots(pv);
    });
}

static void write_index_list(const std::vector<MultiIndex>& v, uint64_t* count, uint64_t* out)
{
    *count = v.size();
    if (!out) return;
    size_t o = 0;
    for (const auto& c : v)
        for (size_t x : c) out[o++] = x;
}

int oracle_tree_subregion(void* h, uint64_t u, uint64_t v, uint64_t* nl, uint64_t* left, uint64_t* nr, uint64_t* right)
{
    return guarded([&] {
        auto keys = static_cast<OracleTree*>(h)->st->graph.subregion_vertices(TreeEdge(u, v));
        *nl = keys.first.size();
        *nr = keys.second.size();
        if (left) std::copy(keys.first.begin(), keys.first.end(), left);
        if (right) std::copy(keys.second.begin(), keys.second.end(), right);
    });
}
// distances to every edge of the graph in sorted edge order
int oracle_tree_distance_edges(void* h, uint64_t u, uint64_t v, uint64_t* out)
{
    return guarded([&] {
        auto& g = static_cast<OracleTree*>(h)->st->graph;
        auto d = g.distance_edges(TreeEdge(u, v));
        size_t k = 0;
        for (const TreeEdge& e : g.edges()) out[k++] = d.at(e);
    });
}
int oracle_tree_candidate_edges(void* h, uint64_t u, uint64_t v, uint64_t* count, uint64_t* out)
{
    return guarded([&] {
        auto c = static_cast<OracleTree*>(h)->st->graph.candidate_edges(TreeEdge(u, v));
        *count = c.size();
        if (out)
            for (size_t k = 0; k < c.size(); ++k) {
                out[2 * k] = c[k].u;
                out[2 * k + 1] = c[k].v;
            }
    });
}
int oracle_tree_edges(void* h, uint64_t* out)
{
    return guarded([&] {
        size_t k = 0;
        for (const TreeEdge& e : static_cast<OracleTree*>(h)->st->graph.edges()) {
            out[2 * k] = e.u;
            out[2 * k + 1] = e.v;
            ++k;
        }
    });
}
int oracle_tree_candidates(void* h, uint64_t u, uint64_t v, uint64_t* nl, uint64_t* left, uint64_t* nr, uint64_t* right)
{
    return guarded([&] {
        std::vector<MultiIndex> l, r;
        tree_candidates(*static_cast<OracleTree*>(h)->st, TreeEdge(u, v), l, r);
        write_index_list(l, nl, left);
        write_index_list(r, nr, right);
    });
}
void oracle_tree_set_proposer(void* h, int kind, uint64_t seed)
{
    auto* t = static_cast<OracleTree*>(h);
    t->st->proposer = kind;
    t->st->proposer_seed = seed;
}
int oracle_tree_push_history(void* h, const uint64_t* key, uint64_t key_len, const uint64_t* cols, uint64_t count)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        SubtreeKey k(key, key + key_len);
        std::vector<MultiIndex> v;
        for (size_t c = 0; c < count; ++c) v.emplace_back(cols + c * key_len, cols + (c + 1) * key_len);
        PivotTable tab;
        tab[k] = v;
        t->st->ijset_history.push_back(tab);
    });
}
int oracle_tree_pivots(void* h, const uint64_t* key, uint64_t key_len, uint64_t* count, uint64_t* out)
{
    return guarded([&] {
        SubtreeKey k(key, key + key_len);
        write_index_list(static_cast<OracleTree*>(h)->st->pivots_of(k), count, out);
    });
}
int oracle_tree_update_edge(void* h, uint64_t u, uint64_t v, uint64_t max_bond_dim, double rel_tol, double abs_tol,
                            uint64_t* rank, uint64_t* rows, uint64_t* cols, double* pivot_errors)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : (size_t)max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = true;
        MatrixLuciFactors sel = tree_update_edge(*t->st, TreeEdge(u, v), t->eval, o);
        *rank = sel.rank;
        for (size_t k = 0; k < sel.row_indices.size(); ++k) {
            if (rows) rows[k] = sel.row_indices[k];
            if (cols) cols[k] = sel.col_indices[k];
        }
        if (pivot_errors)
            for (size_t k = 0; k < sel.pivot_errors.size(); ++k) pivot_errors[k] = sel.pivot_errors[k];
        t->has_net = false;
    });
}
int oracle_tree_optimize(void* h, int with_initial, const uint64_t* pivots, uint64_t n_pivots, double tolerance,
                         uint64_t max_iter, uint64_t max_bond_dim, int normalize_error, int enable_global_pivots,
                         uint64_t nsearch, uint64_t max_nglobal_pivot, double tol_margin, int has_seed, uint64_t seed,
                         uint64_t* n_iter, uint64_t* ranks, double* errors)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        TreeTciOptions o = make_tree_options(tolerance, max_iter, max_bond_dim, normalize_error, enable_global_pivots, nsearch,
                                             max_nglobal_pivot, tol_margin, has_seed, seed);
        if (with_initial) {
            const size_t n = t->st->local_dims.size();
            std::vector<MultiIndex> pv;
            for (size_t k = 0; k < n_pivots; ++k) pv.emplace_back(pivots + k * n, pivots + (k + 1) * n);
            t->last = tree_crossinterpolate2(*t->st, t->eval, pv, o);
        } else {
            t->last = tree_optimize(*t->st, t->eval, o);
        }
        t->has_net = false;
        *n_iter = t->last.ranks.size();
        for (size_t k = 0; k < t->last.ranks.size(); ++k) {
            if (ranks) ranks[k] = t->last.ranks[k];
            if (errors) errors[k] = t->last.errors[k];
        }
    });
}
int oracle_tree_bond_errors(void* h, double* out)
{
    return guarded([&] {
        size_t k = 0;
        for (const auto& kv : static_cast<OracleTree*>(h)->st->bond_errors) out[k++] = kv.second;
    });
}
int oracle_tree_pivot_errors(void* h, uint64_t* count, double* out)
{
    return guarded([&] {
        const auto& e = static_cast<OracleTree*>(h)->st->pivot_errors;
        *count = e.size();
        if (out) std::copy(e.begin(), e.end(), out);
    });
}
double oracle_tree_max_sample_value(void* h) { return static_cast<OracleTree*>(h)->st->max_sample_value; }
void oracle_tree_set_max_sample_value(void* h, double v) { static_cast<OracleTree*>(h)->st->max_sample_value = v; }
double oracle_tree_max_bond_error(void* h) { return static_cast<OracleTree*>(h)->st->max_bond_error(); }
uint64_t oracle_tree_max_bond_dim(void* h) { return static_cast<OracleTree*>(h)->st->max_bond_dim(); }
void oracle_tree_flush_pivot_errors(void* h) { static_cast<OracleTree*>(h)->st->flush_pivot_errors(); }

int oracle_tree_materialize(void* h, uint64_t center_site)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        t->net = tree_materialize(*t->st, t->eval, (size_t)center_site);
        t->has_net = true;
    });
}
int oracle_tree_site_tensor(void* h, uint64_t site, uint64_t* ndims, uint64_t* dims, double* out)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        if (!t->has_net) throw OracleError(ERR_INVALID_ARGUMENT, "materialize first");
        const auto& st = t->net.tensors.at(site);
        *ndims = st.dims.size();
        if (dims) std::copy(st.dims.begin(), st.dims.end(), dims);
        if (out) std::copy(st.data.begin(), st.data.end(), out);
    });
}
int oracle_tree_evaluate(void* h, const uint64_t* idx, uint64_t n_pts, double* out)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        if (!t->has_net) throw OracleError(ERR_INVALID_ARGUMENT, "materialize first");
        const size_t n = t->st->local_dims.size();
        MultiIndex mi(n);
        for (size_t p = 0; p < n_pts; ++p) {
            for (size_t s = 0; s < n; ++s) mi[s] = idx[s + n * p];
            out[p] = t->net.evaluate(mi);
        }
    });
}
int oracle_tree_find_global_pivots(void* h, uint64_t nsearch, uint64_t max_nglobal_pivot, double tol_margin, double abs_tol,
                                   uint64_t seed, uint64_t* count, uint64_t* out)
{
    return guarded([&] {
        auto* t = static_cast<OracleTree*>(h);
        auto pv = tree_find_global_pivots(*t->st, t->eval, nsearch, max_nglobal_pivot, tol_margin, abs_tol, seed);
        write_index_list(pv, count, out);
    });
}
int oracle_solve_right_full_piv_lu(const double* pi1, uint64_t rows, uint64_t cols, const double* p, double* x)
{
    return guarded([&] {
        std::vector<double> a(pi1, pi1 + rows * cols), b(p, p + cols * cols);
        auto r = tree_detail::solve_right_full_piv_lu(a, rows, cols, b, cols, cols);
        std::copy(r.begin(), r.end(), x);
    });
}

// ---- quantics front end (t4a_oracle_quantics.hpp) ----
typedef double (*coord_cb_t)(void* ctx, const double* x, uint64_t n);
typedef double (*grididx_cb_t)(void* ctx, const uint64_t* idx, uint64_t n);

static QtciOptions make_qtci_options(double tolerance, uint64_t max_bond_dim, uint64_t max_iter, uint64_t n_random, int unfolding,
                                     int normalize_error, int has_seed, uint64_t seed)
{
    QtciOptions o;
    o.tolerance = tolerance;
    o.max_bond_dim = (size_t)max_bond_dim;
    o.max_iter = (size_t)max_iter;
    o.n_random_init_pivot = (size_t)n_random;
    o.unfolding = unfolding ? Unfolding::Fused : Unfolding::Interleaved;
    o.normalize_error = normalize_error != 0;
    o.has_seed = has_seed != 0;
    o.seed = seed;
    return o;
}
static std::vector<std::vector<size_t>> grid_pivots(const uint64_t* pivots, uint64_t n_pivots, size_t n_vars)
{
    std::vector<std::vector<size_t>> p;
    for (size_t k = 0; k < n_pivots; ++k) p.emplace_back(pivots + k * n_vars, pivots + (k + 1) * n_vars);
    return p;
}

void* oracle_qtci_continuous(const uint64_t* rs, uint64_t n_vars, const double* lower, const double* upper, int include_endpoint,
                             int grid_unfolding, coord_cb_t cb, void* ctx, int has_pivots, const uint64_t* pivots,
                             uint64_t n_pivots, double tolerance, uint64_t max_bond_dim, uint64_t max_iter, uint64_t n_random,
                             int unfolding, int normalize_error, int has_seed, uint64_t seed)
{
    void* out = nullptr;
    guarded([&] {
        QuanticsGrid grid(std::vector<size_t>(rs, rs + n_vars), grid_unfolding ? Unfolding::Fused : Unfolding::Interleaved, true,
                          std::vector<double>(lower, lower + n_vars), std::vector<double>(upper, upper + n_vars),
                          include_endpoint != 0);
        auto pv = grid_pivots(pivots, has_pivots ? n_pivots : 0, n_vars);
        CoordFn f = [cb, ctx](const std::vector<double>& x) { return cb(ctx, x.data(), x.size()); };
        out = new QuanticsTensorCI2(quanticscrossinterpolate(
            grid, f, has_pivots ? &pv : nullptr,
            make_qtci_options(tolerance, max_bond_dim, max_iter, n_random, unfolding, normalize_error, has_seed, seed)));
    });
    return out;
}
void* oracle_qtci_discrete(const uint64_t* sizes, uint64_t n_vars, grididx_cb_t cb, void* ctx, int has_pivots,
                           const uint64_t* pivots, uint64_t n_pivots, double tolerance, uint64_t max_bond_dim, uint64_t max_iter,
                           uint64_t n_random, int unfolding, int normalize_error, int has_seed, uint64_t seed)
{
    void* out = nullptr;
    guarded([&] {
        auto pv = grid_pivots(pivots, has_pivots ? n_pivots : 0, n_vars);
        GridIdxFn f = [cb, ctx](const std::vector<size_t>& idx) {
            std::vector<uint64_t> v(idx.begin(), idx.end());
            return cb(ctx, v.data(), v.size());
        };
        out = new QuanticsTensorCI2(quanticscrossinterpolate_discrete(
            std::vector<size_t>(sizes, sizes + n_vars), f, has_pivots ? &pv : nullptr,
            make_qtci_options(tolerance, max_bond_dim, max_iter, n_random, unfolding, normalize_error, has_seed, seed)));
    });
    return out;
}
void* oracle_qtci_from_arrays(const double* xvals, const uint64_t* sizes, uint64_t n_vars, coord_cb_t cb, void* ctx, int has_pivots,
                              const uint64_t* pivots, uint64_t n_pivots, double tolerance, uint64_t max_bond_dim,
                              uint64_t max_iter, uint64_t n_random, int unfolding, int normalize_error, int has_seed, uint64_t seed)
{
    void* out = nullptr;
    guarded([&] {
        std::vector<std::vector<double>> xv;
        size_t off = 0;
        for (size_t d = 0; d < n_vars; ++d) {
            xv.emplace_back(xvals + off, xvals + off + sizes[d]);
            off += sizes[d];
        }
        auto pv = grid_pivots(pivots, has_pivots ? n_pivots : 0, n_vars);
        CoordFn f = [cb, ctx](const std::vector<double>& x) { return cb(ctx, x.data(), x.size()); };
        out = new QuanticsTensorCI2(quanticscrossinterpolate_from_arrays(
            xv, f, has_pivots ? &pv : nullptr,
            make_qtci_options(tolerance, max_bond_dim, max_iter, n_random, unfolding, normalize_error, has_seed, seed)));
    });
    return out;
}
void oracle_qtci_release(void* h) { delete static_cast<QuanticsTensorCI2*>(h); }
uint64_t oracle_qtci_n_sites(void* h) { return static_cast<QuanticsTensorCI2*>(h)->tt.len(); }
uint64_t oracle_qtci_n_vars(void* h) { return static_cast<QuanticsTensorCI2*>(h)->grid.n_vars(); }
int oracle_qtci_is_discretized(void* h) { return static_cast<QuanticsTensorCI2*>(h)->grid.discretized ? 1 : 0; }
uint64_t oracle_qtci_cache_size(void* h) { return static_cast<QuanticsTensorCI2*>(h)->cache.size(); }
uint64_t oracle_qtci_n_iterations(void* h) { return static_cast<QuanticsTensorCI2*>(h)->ranks.size(); }
int oracle_qtci_history(void* h, uint64_t* ranks, double* errors)
{
    return guarded([&] {
        auto* q = static_cast<QuanticsTensorCI2*>(h);
        for (size_t k = 0; k < q->ranks.size(); ++k) {
            ranks[k] = q->ranks[k];
            errors[k] = q->errors[k];
        }
    });
}
int oracle_qtci_evaluate(void* h, const uint64_t* grididx, uint64_t n_pts, double* out)
{
    return guarded([&] {
        auto* q = static_cast<QuanticsTensorCI2*>(h);
        const size_t nv = q->grid.n_vars();
        for (size_t p = 0; p < n_pts; ++p) out[p] = q->evaluate(std::vector<size_t>(grididx + p * nv, grididx + (p + 1) * nv));
    });
}
int oracle_qtci_sum(void* h, double* sum, double* integral)
{
    return guarded([&] {
        auto* q = static_cast<QuanticsTensorCI2*>(h);
        *sum = q->sum();
        *integral = q->integral();
    });
}
int oracle_qtci_site_tensor(void* h, uint64_t site, uint64_t* dims3, double* out)
{
    return guarded([&] {
        const auto& t = static_cast<QuanticsTensorCI2*>(h)->tt.tensors.at(site);
        dims3[0] = t.l;
        dims3[1] = t.s;
        dims3[2] = t.r;
        if (out && !t.d.empty()) std::memcpy(out, t.d.data(), t.d.size() * sizeof(double));
    });
}
int oracle_qtci_cachedata(void* h, uint64_t* quantics, double* values)
{
    return guarded([&] {
        auto* q = static_cast<QuanticsTensorCI2*>(h);
        size_t k = 0, o = 0;
        for (const auto& kv : q->cache) {
            for (size_t v : kv.first) quantics[o++] = v;
            values[k++] = kv.second;
        }
    });
}
// which: 0 grididx -> quantics, 1 quantics -> grididx, 2 quantics -> origcoord (out_d), 3 local dimensions, 4 grid step (out_d)
int oracle_qtci_grid(void* h, int which, const uint64_t* in, uint64_t* out_u, double* out_d)
{
    return guarded([&] {
        const QuanticsGrid& g = static_cast<QuanticsTensorCI2*>(h)->grid;
        if (which == 0) {
            auto q = g.grididx_to_quantics(std::vector<size_t>(in, in + g.n_vars()));
            std::copy(q.begin(), q.end(), out_u);
        } else if (which == 1) {
            auto x = g.quantics_to_grididx(MultiIndex(in, in + g.sites.size()));
            std::copy(x.begin(), x.end(), out_u);
        } else if (which == 2) {
            if (!g.discretized) throw OracleError(ERR_INVALID_ARGUMENT, "original coordinates are only available for discretized grids");
            auto x = g.quantics_to_origcoord(MultiIndex(in, in + g.sites.size()));
            std::copy(x.begin(), x.end(), out_d);
        } else if (which == 3) {
            auto d = g.local_dimensions();
            std::copy(d.begin(), d.end(), out_u);
        } else {
            auto st = g.grid_step();
            std::copy(st.begin(), st.end(), out_d);
        }
    });
}
int oracle_qtci_tree_pivots(void* h, const uint64_t* key, uint64_t key_len, uint64_t* count, uint64_t* out)
{
    return guarded([&] {
        SubtreeKey k(key, key + key_len);
        write_index_list(static_cast<QuanticsTensorCI2*>(h)->tci->pivots_of(k), count, out);
    });
}

// ---- batched quantics (vector valued) ----
typedef int (*coord_vec_cb_t)(void* ctx, const double* x, uint64_t n, double* out, uint64_t max_out); // returns the count

void* oracle_qtci_batched(const uint64_t* rs, uint64_t n_vars, const double* lower, const double* upper, int include_endpoint,
                          int grid_unfolding, coord_vec_cb_t cb, void* ctx, const uint64_t* output_dims, uint64_t n_output_dims,
                          int has_pivots, const uint64_t* pivots, uint64_t n_pivots, double tolerance, uint64_t max_bond_dim,
                          uint64_t max_iter, uint64_t n_random, int unfolding, int normalize_error, int has_seed, uint64_t seed)
{
    void* out = nullptr;
    guarded([&] {
        QuanticsGrid grid(std::vector<size_t>(rs, rs + n_vars), grid_unfolding ? Unfolding::Fused : Unfolding::Interleaved, true,
                          std::vector<double>(lower, lower + n_vars), std::vector<double>(upper, upper + n_vars),
                          include_endpoint != 0);
        auto pv = grid_pivots(pivots, has_pivots ? n_pivots : 0, n_vars);
        std::vector<size_t> od(output_dims, output_dims + n_output_dims);
        size_t cap = 1;
        for (size_t d : od) cap *= std::max<size_t>(d, 1);
        CoordVecFn f = [cb, ctx, cap](const std::vector<double>& x) {
            std::vector<double> v(cap + 8);
            const int k = cb(ctx, x.data(), x.size(), v.data(), v.size());
            v.resize(k < 0 ? 0 : (size_t)k);
            return v;
        };
        out = new QuanticsBatchedResult(quanticscrossinterpolate_batched(
            grid, f, od, has_pivots ? &pv : nullptr,
            make_qtci_options(tolerance, max_bond_dim, max_iter, n_random, unfolding, normalize_error, has_seed, seed)));
    });
    return out;
}
void oracle_qtci_batched_release(void* h) { delete static_cast<QuanticsBatchedResult*>(h); }
uint64_t oracle_qtci_batched_len(void* h) { return static_cast<QuanticsBatchedResult*>(h)->tt.len(); }
uint64_t oracle_qtci_batched_user_calls(void* h) { return static_cast<QuanticsBatchedResult*>(h)->n_user_calls; }
uint64_t oracle_qtci_batched_n_iterations(void* h) { return static_cast<QuanticsBatchedResult*>(h)->ranks.size(); }
int oracle_qtci_batched_history(void* h, uint64_t* ranks, double* errors)
{
    return guarded([&] {
        auto* r = static_cast<QuanticsBatchedResult*>(h);
        for (size_t k = 0; k < r->ranks.size(); ++k) ranks[k] = r->ranks[k];
        for (size_t k = 0; k < r->errors.size(); ++k) errors[k] = r->errors[k];
    });
}
int oracle_qtci_batched_site_tensor(void* h, uint64_t site, uint64_t* dims3, double* out)
{
    return guarded([&] {
        const auto& t = static_cast<QuanticsBatchedResult*>(h)->tt.tensors.at(site);
        dims3[0] = t.l;
        dims3[1] = t.s;
        dims3[2] = t.r;
        if (out && !t.d.empty()) std::memcpy(out, t.d.data(), t.d.size() * sizeof(double));
    });
}

} // extern "C"
