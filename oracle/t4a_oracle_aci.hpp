// t4a_oracle_aci.hpp — CPU restatement of tensor4all-aci (Alternating Cross Interpolation for elementwise operations on
// tensor trains): crates/tensor4all-aci/src/{elementwise,state,local,global_guard,random_tt,validation}.rs.
//
// TEST INFRASTRUCTURE ONLY (see t4a_oracle.hpp): nothing in the product path includes or links this file.
//
// Parity notes
//   * Every matrix product of the path (frame x core, core x frame, left factor x right factor, previous core x LUCI left
//     factor) is tenferro matmul in the reference (third party, summation order backend-defined).  Restated as mat_mul of
//     t4a_oracle.hpp: k ascending, separately rounded multiply and add — the device kernels use the same order, so the local
//     candidate matrices are bit-identical on both sides and the rrLU pivots follow.
//   * The default initial guess draws standard normals from rand_chacha::ChaCha8Rng and the global guard draws its starting
//     points from rand::rngs::StdRng (un-vendored crates): both restated from the published algorithms (ChaCha8 + rand_distr's
//     256-layer ziggurat in t4a_oracle_rng2.hpp; StdRng in t4a_oracle_rng.hpp), pinned to published vectors where they exist.
//     Runs with an explicit AciOptions::initial_guess and enable_global_guard = false involve no random numbers.
//   * append_row (state.rs:1343-1359) extends the COLUMN-major buffer of the frame by the new row and re-reads it with one
//     more row; that is a row append only for frames with one column.  Restated as written.
#pragma once

#include <cmath>
#include <functional>

#include "t4a_oracle_tt.hpp"
#include "t4a_oracle_rng2.hpp"

namespace t4a_oracle {

// batch.rs:33-217: values[input + n_inputs * point]; out[point]
using AciOp = std::function<void(const double* values, size_t n_inputs, size_t n_points, double* out)>;

enum class AciTermination : int { Converged = 0, RankLimited = 1, MaxIterations = 2 }; // result.rs

struct AciOptions { // options.rs:37-168
    size_t max_iters = 20;
    size_t min_iters = 2;
    bool has_max_bond_dim = false;
    size_t max_bond_dim = 0;
    double tolerance = 1e-12;
    bool scale_tolerance = true;
    bool has_initial_guess = false;
    SimpleTensorTrain initial_guess;
    uint64_t rng_seed = 0;
    bool enable_global_guard = true;
    size_t nsearch_global_pivots = 5;
    size_t max_nglobal_pivot = 5;
    size_t nsweeps_global_search = 100;
    double tol_margin_global_search = 10.0;
};

struct AciResult { // result.rs
    SimpleTensorTrain tensor_train;
    std::vector<size_t> ranks;
    std::vector<double> errors;
    std::vector<size_t> nglobal_pivots;
    AciTermination termination = AciTermination::MaxIterations;
};

inline void aci_validate_options(const AciOptions& o) // validation.rs:4-46
{
    if (o.max_iters == 0) throw OracleError(ERR_INVALID_ARGUMENT, "max_iters must be at least 1");
    if (o.min_iters == 0) throw OracleError(ERR_INVALID_ARGUMENT, "min_iters must be at least 1");
    if (o.has_max_bond_dim && o.max_bond_dim == 0) throw OracleError(ERR_INVALID_ARGUMENT, "max_bond_dim must be at least 1");
    if (o.min_iters > o.max_iters) throw OracleError(ERR_INVALID_ARGUMENT, "min_iters must be less than or equal to max_iters");
    if (!std::isfinite(o.tolerance) || o.tolerance < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "tolerance must be finite and non-negative");
    if (!std::isfinite(o.tol_margin_global_search) || o.tol_margin_global_search < 0.0)
        throw OracleError(ERR_INVALID_ARGUMENT, "tol_margin_global_search must be finite and non-negative");
}

inline std::vector<size_t> aci_site_dims(const SimpleTensorTrain& tt)
{
    std::vector<size_t> d;
    for (const auto& t : tt.tensors) d.push_back(t.s);
    return d;
}

inline std::vector<size_t> aci_validate_inputs(const std::vector<SimpleTensorTrain>& inputs) // validation.rs:48-92
{
    if (inputs.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "inputs must not be empty");
    const std::vector<size_t> site_dims = aci_site_dims(inputs[0]);
    if (site_dims.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "input tensor trains must have at least one site");
    for (size_t k = 0; k < inputs.size(); ++k) {
        if (inputs[k].len() != site_dims.size()) throw OracleError(ERR_INVALID_ARGUMENT, "input tensor trains must have the same length");
        for (size_t s = 0; s < site_dims.size(); ++s) {
            const Tensor3& c = inputs[k].tensors[s];
            if (c.l == 0 || c.s == 0 || c.r == 0) throw OracleError(ERR_INVALID_ARGUMENT, "core dimensions must be positive");
            if (c.s != site_dims[s]) throw OracleError(ERR_INVALID_ARGUMENT, "site dimension mismatch between inputs");
        }
    }
    return site_dims;
}

// random_tt.rs:97-137
inline std::vector<size_t> aci_default_link_dims(const std::vector<SimpleTensorTrain>& inputs, const std::vector<size_t>& site_dims,
                                                 bool has_cap, size_t cap)
{
    const size_t n = site_dims.size();
    if (n <= 1) return {};
    auto sat_mul = [](size_t a, size_t b) {
        if (a != 0 && b > std::numeric_limits<size_t>::max() / a) throw OracleError(ERR_INVALID_ARGUMENT, "site dimension product overflows usize");
        return a * b;
    };
    std::vector<size_t> lp(n - 1), rp(n - 1, 1);
    size_t acc = 1;
    for (size_t b = 0; b + 1 < n; ++b) {
        acc = sat_mul(acc, site_dims[b]);
        lp[b] = acc;
    }
    acc = 1;
    for (size_t b = n - 1; b-- > 0;) {
        acc = sat_mul(acc, site_dims[b + 1]);
        rp[b] = acc;
    }
    std::vector<size_t> link(n - 1);
    for (size_t b = 0; b + 1 < n; ++b) {
        size_t m = std::numeric_limits<size_t>::max();
        for (const auto& in : inputs) m = std::min(m, in.tensors[b].r);
        size_t d = std::min(lp[b], rp[b]);
        if (has_cap) d = std::min(d, cap);
        link[b] = std::max<size_t>(std::min(d, m), 1);
    }
    return link;
}

// scalar.rs:8-20 sample_standard_normal: rand_distr StandardNormal (ziggurat) on ChaCha8Rng (t4a_oracle_rng2.hpp)
inline double aci_standard_normal(OracleChaCha8Rng& rng) { return oracle_ziggurat().normal(rng); }

constexpr size_t ACI_MAX_INITIAL_GUESS_ENTRIES = 10000000; // random_tt.rs:12-13

inline SimpleTensorTrain aci_initial_guess(const std::vector<SimpleTensorTrain>& inputs, const AciOptions& o) // random_tt.rs:15-39
{
    const std::vector<size_t> site_dims = aci_validate_inputs(inputs);
    aci_validate_options(o);
    if (o.has_initial_guess) { // :41-83
        const SimpleTensorTrain& g = o.initial_guess;
        if (aci_site_dims(g) != site_dims) throw OracleError(ERR_INVALID_ARGUMENT, "initial guess site dimensions must match inputs");
        size_t total = 0;
        for (const auto& c : g.tensors) {
            if (c.l == 0 || c.s == 0 || c.r == 0) throw OracleError(ERR_INVALID_ARGUMENT, "initial guess core dimensions must be positive");
            total += c.d.size();
        }
        for (size_t d : g.link_dims())
            if (o.has_max_bond_dim && d > o.max_bond_dim) throw OracleError(ERR_INVALID_ARGUMENT, "initial guess bond dimension exceeds max_bond_dim");
        if (total > ACI_MAX_INITIAL_GUESS_ENTRIES) throw OracleError(ERR_INVALID_ARGUMENT, "initial guess total size exceeds internal limit");
        return SimpleTensorTrain::make(g.tensors);
    }
    const std::vector<size_t> link = aci_default_link_dims(inputs, site_dims, o.has_max_bond_dim, o.max_bond_dim);
    OracleChaCha8Rng rng(o.rng_seed); // random_tt.rs:31
    std::vector<Tensor3> cores;
    size_t total = 0;
    for (size_t s = 0; s < site_dims.size(); ++s) {
        Tensor3 c(s == 0 ? 1 : link[s - 1], site_dims[s], s < link.size() ? link[s] : 1);
        total += c.d.size();
        if (total > ACI_MAX_INITIAL_GUESS_ENTRIES) throw OracleError(ERR_INVALID_ARGUMENT, "initial guess total size exceeds internal limit");
        for (double& v : c.d) v = aci_standard_normal(rng);
        cores.push_back(std::move(c));
    }
    return SimpleTensorTrain::make(std::move(cores));
}

struct AciFrame {
    bool present = false;
    Matrix m;
};

struct ElementwiseProblem { // state.rs:24-38
    std::vector<SimpleTensorTrain> inputs;
    SimpleTensorTrain solution;
    std::vector<std::vector<AciFrame>> left_frames, right_frames; // [input][site 0..n]
    std::vector<double> pivot_errors, pivot_scales;
    size_t n_op_points = 0; // operator evaluations (test aid)

    size_t len() const { return solution.len(); }
    size_t n_inputs() const { return inputs.size(); }
    size_t rank() const
    {
        size_t r = 1;
        for (size_t d : solution.link_dims()) r = std::max(r, d);
        return r;
    }

    static Matrix left_grouped(const Tensor3& c) { return Matrix(c.l, c.s * c.r, c.d.data()); }  // :46-64
    static Matrix right_grouped(const Tensor3& c) { return Matrix(c.l * c.s, c.r, c.d.data()); }

    ElementwiseProblem(std::vector<SimpleTensorTrain> in, const AciOptions& o) // :69-109
    {
        solution = aci_initial_guess(in, o);
        inputs = std::move(in);
        const size_t n = solution.len();
        left_frames.assign(inputs.size(), std::vector<AciFrame>(n + 1));
        right_frames.assign(inputs.size(), std::vector<AciFrame>(n + 1));
        for (size_t k = 0; k < inputs.size(); ++k) {
            left_frames[k][0].present = right_frames[k][n].present = true;
            left_frames[k][0].m = Matrix(1, 1);
            left_frames[k][0].m(0, 0) = 1.0;
            right_frames[k][n].m = left_frames[k][0].m;
        }
        pivot_errors.assign(n > 0 ? n - 1 : 0, 0.0);
        pivot_scales.assign(n > 0 ? n - 1 : 0, 0.0);
        initialize_right_frames();
    }

    // :215-255 — rows `row_indices` of frame[site] x core[site] (left x (site, right)), re-read as (left*site) x right
    void update_left_frames(size_t site, const std::vector<size_t>& rows)
    {
        for (size_t k = 0; k < n_inputs(); ++k) {
            if (!left_frames[k][site].present) throw OracleError(ERR_INVALID_ARGUMENT, "missing left frame");
            const Matrix& src = left_frames[k][site].m;
            const Tensor3& core = inputs[k].tensors[site];
            if (src.nc != core.l) throw OracleError(ERR_INVALID_ARGUMENT, "left frame/input bond mismatch");
            const size_t full_rows = src.nr * core.s;
            for (size_t r : rows)
                if (r >= full_rows) throw OracleError(ERR_INVALID_ARGUMENT, "row selection out of bounds");
            const Matrix full = mat_mul(src, left_grouped(core));
            Matrix sel(rows.size(), core.r);
            for (size_t i = 0; i < rows.size(); ++i)
                for (size_t r = 0; r < core.r; ++r) sel(i, r) = full.a[rows[i] + full_rows * r];
            left_frames[k][site + 1].present = true;
            left_frames[k][site + 1].m = std::move(sel);
        }
    }
    // :257-299
    void update_right_frames(size_t site, const std::vector<size_t>& cols)
    {
        for (size_t k = 0; k < n_inputs(); ++k) {
            if (!right_frames[k][site + 1].present) throw OracleError(ERR_INVALID_ARGUMENT, "missing right frame");
            const Matrix& src = right_frames[k][site + 1].m;
            const Tensor3& core = inputs[k].tensors[site];
            if (core.r != src.nr) throw OracleError(ERR_INVALID_ARGUMENT, "right frame/input bond mismatch");
            const size_t full_cols = core.s * src.nc;
            for (size_t c : cols)
                if (c >= full_cols) throw OracleError(ERR_INVALID_ARGUMENT, "column selection out of bounds");
            const Matrix full = mat_mul(right_grouped(core), src);
            Matrix sel(core.l, cols.size());
            for (size_t j = 0; j < cols.size(); ++j)
                for (size_t l = 0; l < core.l; ++l) sel(l, j) = full.a[l + core.l * cols[j]];
            right_frames[k][site].present = true;
            right_frames[k][site].m = std::move(sel);
        }
    }

    // :862-925
    void initialize_right_frames()
    {
        const size_t n = solution.len();
        std::vector<Tensor3> cores = solution.tensors;
        for (size_t site = n; site-- > 1;) {
            const Tensor3& cur = cores[site];
            const size_t site_dim = cur.s, right_dim = cur.r;
            const Matrix mat(cur.l, cur.s * cur.r, cur.d.data());
            RrLUOptions lo;
            lo.left_orthogonal = false;
            lo.rel_tol = 0.0;
            lo.abs_tol = 0.0;
            MatrixLuciFactors f = matrix_luci_factors_from_matrix(mat, lo);
            const size_t ncols = site_dim * right_dim;
            size_t new_rank = f.rank;
            Matrix lf = f.left, rf = f.right;
            std::vector<size_t> col_idx = f.col_indices;
            if (f.rank == 0) {
                new_rank = 1;
                lf = Matrix(mat.nr, 1);
                rf = Matrix(1, ncols);
                col_idx = {0};
            }
            Tensor3 nc(new_rank, site_dim, right_dim);
            nc.d = rf.a;
            cores[site] = std::move(nc);
            const Tensor3& prev = cores[site - 1];
            const Matrix prod = mat_mul(Matrix(prev.l * prev.s, prev.r, prev.d.data()), lf);
            Tensor3 np(prev.l, prev.s, new_rank);
            np.d = prod.a;
            cores[site - 1] = std::move(np);
            update_right_frames(site, col_idx);
        }
        solution = SimpleTensorTrain::make(std::move(cores));
    }

    // local.rs:299-394 + state.rs:729-860
    void local_update(size_t bond, bool left_orthogonal, const AciOptions& o, const AciOp& op)
    {
        const size_t n = len();
        if (n < 2 || bond >= n - 1) throw OracleError(ERR_INVALID_ARGUMENT, "bond index out of bounds");
        const Tensor3& lc = solution.tensors[bond];
        const Tensor3& rc = solution.tensors[bond + 1];
        if (lc.r != rc.l) throw OracleError(ERR_INVALID_ARGUMENT, "adjacent solution core bond mismatch");
        const size_t lrank = lc.l, s1 = lc.s, s2 = rc.s, rrank = rc.r;
        const size_t nrows = lrank * s1, ncols = s2 * rrank, K = n_inputs();
        std::vector<double> values(K * nrows * ncols);
        for (size_t k = 0; k < K; ++k) {
            if (!left_frames[k][bond].present || !right_frames[k][bond + 2].present) throw OracleError(ERR_INVALID_ARGUMENT, "missing frame");
            const Matrix& lfr = left_frames[k][bond].m;
            const Matrix& rfr = right_frames[k][bond + 2].m;
            const Tensor3& a = inputs[k].tensors[bond];
            const Tensor3& b = inputs[k].tensors[bond + 1];
            if (lfr.nc != a.l || a.r != b.l || b.r != rfr.nr) throw OracleError(ERR_INVALID_ARGUMENT, "frame/input bond mismatch");
            if (lfr.nr * a.s != nrows || b.s * rfr.nc != ncols) throw OracleError(ERR_INVALID_ARGUMENT, "local block shape mismatch");
            const Matrix lf = mat_mul(lfr, left_grouped(a));   // (R, s1, m): build_left_factor local.rs:692
            const Matrix rf = mat_mul(right_grouped(b), rfr);  // (m, s2, C): build_right_factor :704
            const Matrix v = mat_mul(Matrix(nrows, a.r, lf.a.data()), Matrix(a.r, ncols, rf.a.data()));
            for (size_t p = 0; p < nrows * ncols; ++p) values[k + K * p] = v.a[p];
        }
        Matrix pi(nrows, ncols);
        op(values.data(), K, nrows * ncols, pi.a.data());
        n_op_points += nrows * ncols;
        double scale = 0.0;
        for (double v : pi.a) scale = std::fmax(scale, std::sqrt(v * v));
        RrLUOptions lo;
        lo.max_bond_dim = o.has_max_bond_dim ? o.max_bond_dim : std::numeric_limits<size_t>::max();
        lo.rel_tol = o.scale_tolerance ? o.tolerance : 0.0;
        lo.abs_tol = o.scale_tolerance ? 0.0 : o.tolerance;
        lo.left_orthogonal = left_orthogonal;
        MatrixLuciFactors f = matrix_luci_factors_from_matrix(pi, lo);
        const double pivot_error = f.pivot_errors.empty() ? 0.0 : f.pivot_errors.back();
        size_t new_rank = f.rank;
        if (f.rank == 0) {
            new_rank = 1;
            f.left = Matrix(nrows, 1);
            f.right = Matrix(1, ncols);
            f.row_indices = {0};
            f.col_indices = {0};
        }
        Tensor3 nl(lrank, s1, new_rank), nr(new_rank, s2, rrank);
        nl.d = f.left.a;
        nr.d = f.right.a;
        solution.tensors[bond] = std::move(nl);
        solution.tensors[bond + 1] = std::move(nr);
        if (left_orthogonal) update_left_frames(bond, f.row_indices);
        else update_right_frames(bond + 1, f.col_indices);
        pivot_errors[bond] = pivot_error;
        pivot_scales[bond] = scale;
    }

    // state.rs:1286-1319
    static std::vector<double> left_environment(const SimpleTensorTrain& tt, const size_t* prefix, size_t len)
    {
        Matrix env(1, 1);
        env(0, 0) = 1.0;
        for (size_t site = 0; site < len; ++site) {
            const Tensor3& c = tt.tensors[site];
            Matrix sl(c.l, c.r);
            for (size_t r = 0; r < c.r; ++r)
                for (size_t l = 0; l < c.l; ++l) sl(l, r) = c.at(l, prefix[site], r);
            env = mat_mul(env, sl);
        }
        return env.a;
    }
    static std::vector<double> right_environment(const SimpleTensorTrain& tt, const size_t* suffix, size_t len)
    {
        const size_t start = tt.len() - len;
        Matrix env(1, 1);
        env(0, 0) = 1.0;
        for (size_t off = len; off-- > 0;) {
            const Tensor3& c = tt.tensors[start + off];
            Matrix sl(c.l, c.r);
            for (size_t r = 0; r < c.r; ++r)
                for (size_t l = 0; l < c.l; ++l) sl(l, r) = c.at(l, suffix[off], r);
            env = mat_mul(sl, env);
        }
        return env.a;
    }
    static bool frame_has_row(const AciFrame& f, const std::vector<double>& row) // :1321-1341
    {
        if (!f.present) return false;
        if (f.m.nc != row.size()) throw OracleError(ERR_INVALID_ARGUMENT, "cannot match a row against a frame of another width");
        for (size_t r = 0; r < f.m.nr; ++r) {
            bool all = true;
            for (size_t c = 0; c < f.m.nc; ++c) all = all && (f.m(r, c) == row[c]);
            if (all) return true;
        }
        return false;
    }
    static bool frame_has_col(const AciFrame& f, const std::vector<double>& col) // :1361-1381
    {
        if (!f.present) return false;
        if (f.m.nr != col.size()) throw OracleError(ERR_INVALID_ARGUMENT, "cannot match a column against a frame of another height");
        for (size_t c = 0; c < f.m.nc; ++c) {
            bool all = true;
            for (size_t r = 0; r < f.m.nr; ++r) all = all && (f.m(r, c) == col[r]);
            if (all) return true;
        }
        return false;
    }
    static void append_row(AciFrame& f, const std::vector<double>& row) // :1343-1359 (buffer extended, re-read with rows + 1)
    {
        if (!f.present) throw OracleError(ERR_INVALID_ARGUMENT, "missing left frame");
        if (f.m.nc != row.size()) throw OracleError(ERR_INVALID_ARGUMENT, "cannot append a row of another length");
        f.m.a.insert(f.m.a.end(), row.begin(), row.end());
        f.m.nr += 1;
    }
    static void append_col(AciFrame& f, const std::vector<double>& col) // :1383-1401
    {
        if (!f.present) throw OracleError(ERR_INVALID_ARGUMENT, "missing right frame");
        if (f.m.nr != col.size()) throw OracleError(ERR_INVALID_ARGUMENT, "cannot append a column of another length");
        f.m.a.insert(f.m.a.end(), col.begin(), col.end());
        f.m.nc += 1;
    }

    // state.rs:551-652
    size_t add_global_pivots(const std::vector<MultiIndex>& pivots)
    {
        const size_t n = len();
        size_t new_pivots = 0;
        std::vector<size_t> growth(n + 1, 0), dims(n + 1, 0), lsp(n + 1, 1), rsp(n + 1, 1), bounds(n + 1);
        for (size_t s = 0; s < n; ++s) lsp[s + 1] = lsp[s] * solution.tensors[s].s; // algebraic_bond_bounds :654-681
        for (size_t s = n; s-- > 0;) rsp[s] = rsp[s + 1] * solution.tensors[s].s;
        for (size_t b = 0; b <= n; ++b) bounds[b] = std::min(lsp[b], rsp[b]);
        const std::vector<size_t> link = solution.link_dims();
        for (size_t b = 0; b < link.size(); ++b) dims[b + 1] = link[b];
        for (const MultiIndex& pivot : pivots) {
            if (pivot.size() != n) throw OracleError(ERR_INVALID_ARGUMENT, "global pivot length must match the number of sites");
            bool injected = false;
            for (size_t bond = 1; bond < n; ++bond) {
                if (dims[bond] >= bounds[bond]) continue;
                const bool needs_row = bond + 1 < n, needs_col = bond >= 2;
                bool duplicate = true;
                for (size_t k = 0; k < n_inputs(); ++k) {
                    if (needs_row && !frame_has_row(left_frames[k][bond], left_environment(inputs[k], pivot.data(), bond))) duplicate = false;
                    if (needs_col && !frame_has_col(right_frames[k][bond], right_environment(inputs[k], pivot.data() + bond, n - bond)))
                        duplicate = false;
                }
                if (duplicate) continue;
                for (size_t k = 0; k < n_inputs(); ++k) {
                    if (needs_row) append_row(left_frames[k][bond], left_environment(inputs[k], pivot.data(), bond));
                    if (needs_col) append_col(right_frames[k][bond], right_environment(inputs[k], pivot.data() + bond, n - bond));
                }
                growth[bond] += 1;
                dims[bond] += 1;
                injected = true;
            }
            if (injected) ++new_pivots;
        }
        if (new_pivots > 0) pad_solution_internal_bonds(growth);
        return new_pivots;
    }
    void pad_solution_internal_bonds(const std::vector<size_t>& growth) // :683-727
    {
        const size_t n = len();
        std::vector<Tensor3> cores;
        for (size_t s = 0; s < n; ++s) {
            const Tensor3& c = solution.tensors[s];
            Tensor3 p(s == 0 ? c.l : c.l + growth[s], c.s, s == n - 1 ? c.r : c.r + growth[s + 1]);
            for (size_t r = 0; r < c.r; ++r)
                for (size_t q = 0; q < c.s; ++q)
                    for (size_t l = 0; l < c.l; ++l) p.at(l, q, r) = c.at(l, q, r);
            cores.push_back(std::move(p));
        }
        solution = SimpleTensorTrain::make(std::move(cores));
    }
};

// tensor4all-core/src/floating_zone.rs:46-103
template <class E>
inline std::pair<MultiIndex, double> floating_zone_walk(const std::vector<size_t>& local_dims, const MultiIndex& init, size_t max_sweeps,
                                                        double early_stop_tol, E&& eval_batch)
{
    MultiIndex pivot = init;
    const std::vector<double> e0 = eval_batch(std::vector<MultiIndex>{pivot});
    double max_error = e0.empty() ? 0.0 : e0[0];
    for (size_t sw = 0; sw < max_sweeps; ++sw) {
        const double prev = max_error;
        for (size_t ipos = 0; ipos < local_dims.size(); ++ipos) {
            std::vector<MultiIndex> pts;
            for (size_t v = 0; v < local_dims[ipos]; ++v) {
                MultiIndex p = pivot;
                p[ipos] = v;
                pts.push_back(std::move(p));
            }
            const std::vector<double> errs = eval_batch(pts);
            size_t best_idx = pivot[ipos];
            double best = 0.0;
            for (size_t v = 0; v < errs.size(); ++v)
                if (errs[v] > best) {
                    best = errs[v];
                    best_idx = v;
                }
            pivot[ipos] = best_idx;
            max_error = std::fmax(max_error, best);
        }
        if (max_error == prev || max_error > early_stop_tol) break;
    }
    return {pivot, max_error};
}

inline size_t aci_guard_split(const std::vector<MultiIndex>& pts, size_t n_sites) // global_guard.rs:100-110 (0 == None)
{
    if (pts.size() < 2) return 0;
    for (size_t site = 0; site < n_sites; ++site)
        for (size_t p = 1; p < pts.size(); ++p)
            if (pts[p][site] != pts[0][site]) return site + 1;
    return 0;
}

// global_guard.rs:49-181
inline std::vector<MultiIndex> aci_find_global_pivots(ElementwiseProblem& problem, const AciOp& op, const AciOptions& o, uint64_t seed)
{
    const size_t n = problem.len(), K = problem.n_inputs(), nsearch = o.nsearch_global_pivots;
    if (nsearch == 0 || o.max_nglobal_pivot == 0 || n < 2) return {};
    const std::vector<size_t> site_dims = aci_site_dims(problem.solution);
    OracleStdRng rng(seed); // global_guard.rs:71
    std::vector<MultiIndex> starts(nsearch, MultiIndex(n));
    for (auto& s : starts)
        for (size_t q = 0; q < n; ++q) s[q] = rng.range(site_dims[q]);
    std::vector<TTCache> caches;
    for (const auto& in : problem.inputs) caches.emplace_back(in);
    std::vector<double> sv(K * nsearch), so(nsearch);
    for (size_t k = 0; k < K; ++k) {
        const std::vector<double> v = caches[k].evaluate_many(starts, 0);
        for (size_t p = 0; p < nsearch; ++p) sv[k + K * p] = v[p];
    }
    op(sv.data(), K, nsearch, so.data());
    double max_op = 0.0;
    for (double v : so) max_op = std::fmax(max_op, std::sqrt(v * v));
    const double abs_tol = (o.scale_tolerance && max_op > 0.0) ? o.tolerance * max_op : o.tolerance;
    const double threshold = abs_tol * o.tol_margin_global_search;
    TTCache sol(problem.solution);
    std::vector<std::pair<double, MultiIndex>> best;
    for (const auto& start : starts) {
        auto res = floating_zone_walk(site_dims, start, o.nsweeps_global_search, threshold, [&](const std::vector<MultiIndex>& pts) {
            const size_t np = pts.size();
            const size_t split = aci_guard_split(pts, n);
            std::vector<double> iv(K * np), ov(np);
            for (size_t k = 0; k < K; ++k) {
                const std::vector<double> v = caches[k].evaluate_many(pts, split);
                for (size_t p = 0; p < np; ++p) iv[k + K * p] = v[p];
            }
            op(iv.data(), K, np, ov.data());
            const std::vector<double> s = sol.evaluate_many(pts, split);
            std::vector<double> errs(np);
            for (size_t p = 0; p < np; ++p) {
                const double d = ov[p] - s[p];
                errs[p] = std::sqrt(d * d);
            }
            return errs;
        });
        if (res.second > threshold) best.push_back({res.second, res.first});
    }
    std::stable_sort(best.begin(), best.end(),
                     [](const std::pair<double, MultiIndex>& a, const std::pair<double, MultiIndex>& b) { return a.first > b.first; });
    std::vector<MultiIndex> pivots;
    for (const auto& b : best)
        if (std::find(pivots.begin(), pivots.end(), b.second) == pivots.end()) {
            pivots.push_back(b.second);
            if (pivots.size() >= o.max_nglobal_pivot) break;
        }
    return pivots;
}

// elementwise.rs:381-413
inline bool aci_converged(size_t iteration, const std::vector<size_t>& ranks, const std::vector<double>& errors,
                          const std::vector<size_t>& nglobal, size_t min_iters, double tolerance)
{
    if (iteration == 0 || min_iters == 0 || iteration < min_iters) return false;
    if (errors[iteration - 1] > tolerance) return false;
    const size_t base = ranks[iteration - min_iters];
    for (size_t i = iteration - min_iters; i < iteration; ++i)
        if (ranks[i] > base) return false;
    for (size_t i = iteration - min_iters; i < iteration; ++i)
        if (nglobal[i] != 0) return false;
    return true;
}
// :436-451
inline bool aci_rank_saturated(const std::vector<size_t>& ranks, size_t min_iters, bool has_cap, size_t cap)
{
    if (min_iters == 0 || ranks.size() < min_iters || !has_cap) return false;
    for (size_t i = ranks.size() - min_iters; i < ranks.size(); ++i)
        if (ranks[i] < cap) return false;
    return true;
}
// :453-480
inline double aci_max_error_metric(const std::vector<double>& errs, const std::vector<double>& scales, bool scale_tolerance)
{
    double m = 0.0;
    for (size_t b = 0; b < errs.size(); ++b) {
        const double sc = b < scales.size() ? scales[b] : 0.0;
        const double e = (scale_tolerance && sc > 0.0) ? errs[b] / sc : errs[b];
        m = std::fmax(m, e);
    }
    return m;
}

// elementwise.rs:107-218 (+ one-site path :220-254)
inline AciResult elementwise_batched(const AciOp& op, const std::vector<SimpleTensorTrain>& inputs, const AciOptions& o)
{
    aci_validate_options(o);
    aci_validate_inputs(inputs);
    AciResult res;
    if (inputs[0].len() == 1) {
        const size_t K = inputs.size(), np = inputs[0].tensors[0].s;
        std::vector<double> iv(K * np);
        for (size_t p = 0; p < np; ++p)
            for (size_t k = 0; k < K; ++k) iv[k + K * p] = inputs[k].evaluate({p});
        Tensor3 core(1, np, 1);
        op(iv.data(), K, np, core.d.data());
        res.tensor_train = SimpleTensorTrain::make({core});
        res.termination = AciTermination::Converged;
        return res;
    }
    ElementwiseProblem problem(inputs, o);
    size_t guard_runs = 0;
    for (size_t it = 0; it < o.max_iters; ++it) {
        if (it % 2 == 0)
            for (size_t b = 0; b + 1 < problem.len(); ++b) problem.local_update(b, true, o, op);
        else
            for (size_t b = problem.len() - 1; b-- > 0;) problem.local_update(b, false, o, op);
        res.ranks.push_back(problem.rank());
        res.errors.push_back(aci_max_error_metric(problem.pivot_errors, problem.pivot_scales, o.scale_tolerance));
        const bool capped = o.has_max_bond_dim && problem.rank() >= o.max_bond_dim;
        if (o.enable_global_guard && o.nsearch_global_pivots > 0 && o.max_nglobal_pivot > 0 && !capped) {
            ++guard_runs;
            const std::vector<MultiIndex> pv = aci_find_global_pivots(problem, op, o, o.rng_seed + (uint64_t)guard_runs);
            problem.add_global_pivots(pv);
            res.nglobal_pivots.push_back(pv.size());
        } else {
            res.nglobal_pivots.push_back(0);
        }
        if (aci_converged(it + 1, res.ranks, res.errors, res.nglobal_pivots, o.min_iters, o.tolerance)) {
            res.termination = AciTermination::Converged;
            break;
        }
        if (aci_rank_saturated(res.ranks, o.min_iters, o.has_max_bond_dim, o.max_bond_dim)) {
            res.termination = AciTermination::RankLimited;
            break;
        }
    }
    if (o.has_max_bond_dim && problem.rank() > o.max_bond_dim)
        for (size_t b = 0; b + 1 < problem.len(); ++b) problem.local_update(b, true, o, op);
    res.tensor_train = problem.solution;
    return res;
}

} // namespace t4a_oracle
