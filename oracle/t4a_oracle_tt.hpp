// oracle/t4a_oracle_tt.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp).
// CPU restatement of the tensor-train side of the TCI2 path (SURVEY.md §8 rows a14–a18):
//   * thin QR / thin SVD / complete-pivoting LU facade (tensor4all-tensorbackend/src/backend.rs:715-1060).
//     The reference forwards these to tenferro-rs @ a21a4c6 (cpu-faer), which is not vendored, so they
//     are restated from the published algorithms (Householder QR, one-sided Jacobi SVD, full-pivot LU).
//     Bit-level results: "parity unpinned"; the checks are the reference's own tolerance-level
//     properties (reconstruction 1e-10, orthonormality, descending singular values,
//     backend/tests/mod.rs:58-110, compression/tests/mod.rs:208-291).
//   * SimpleTensorTrain::compress / factorize / factorize_svd (tensor4all-simplett/src/compression.rs)
//   * norm2 (tensor4all-simplett/src/traits.rs:289-354), full_tensor (tensortrain.rs:374-421)
//   * TTCache::evaluate_many (tensor4all-simplett/src/cache.rs:430-744, einsum_helper.rs:192-268)
//   * tensorci2_from_tensor_train (tensor4all-tensorci/src/conversion.rs:66-433)
#pragma once

#include <map>
#include <set>

#include "t4a_oracle.hpp"

namespace t4a_oracle {

// ---------------------------------------------------------------------------------------------
// thin QR: A (m x n) = Q (m x k) R (k x n), k = min(m, n)  — backend.rs:742-760 (tenferro `qr`)
// Householder reflectors; R has the sign convention diag = -sign(x0) * ||x||.
// ---------------------------------------------------------------------------------------------
struct QrResult {
    Matrix q, r;
};
inline QrResult qr_thin(const Matrix& a)
{
    const size_t m = a.nr, n = a.nc, k = std::min(m, n);
    Matrix w = a;
    std::vector<double> tau(k, 0.0), diag(k, 0.0);
    for (size_t j = 0; j < k; ++j) {
        double nrm2 = 0.0;
        for (size_t i = j; i < m; ++i) nrm2 += w(i, j) * w(i, j);
        const double nrm = std::sqrt(nrm2);
        if (nrm == 0.0) {
            tau[j] = 0.0;
            diag[j] = 0.0;
            continue;
        }
        const double x0 = w(j, j);
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        // v = (v0, w[j+1:, j]); tau = 2 / (v.v)
        double vv = v0 * v0;
        for (size_t i = j + 1; i < m; ++i) vv += w(i, j) * w(i, j);
        tau[j] = 2.0 / vv;
        w(j, j) = v0;
        for (size_t c = j + 1; c < n; ++c) {
            double dot = 0.0;
            for (size_t i = j; i < m; ++i) dot += w(i, j) * w(i, c);
            const double f = tau[j] * dot;
            for (size_t i = j; i < m; ++i) w(i, c) -= f * w(i, j);
        }
        diag[j] = alpha;
    }
    QrResult out;
    out.r = Matrix(k, n);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < k && i <= j; ++i) out.r(i, j) = (i == j) ? diag[i] : w(i, j);
    out.q = Matrix(m, k);
    for (size_t i = 0; i < k; ++i) out.q(i, i) = 1.0;
    for (size_t jj = k; jj-- > 0;) {
        if (tau[jj] == 0.0) continue;
        for (size_t c = 0; c < k; ++c) {
            double dot = 0.0;
            for (size_t i = jj; i < m; ++i) dot += w(i, jj) * out.q(i, c);
            const double f = tau[jj] * dot;
            for (size_t i = jj; i < m; ++i) out.q(i, c) -= f * w(i, jj);
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// thin SVD: A (m x n) = U (m x k) diag(S) Vt (k x n), S non-increasing — backend.rs:709-731
// One-sided Jacobi (Hestenes) on the columns of the taller orientation.
// ---------------------------------------------------------------------------------------------
struct SvdResult {
    Matrix u;
    std::vector<double> s;
    Matrix vt;
};

namespace detail {
// Replace the columns of `u` listed in `dead` by unit vectors orthogonal to all other columns.
inline void complete_orthonormal(Matrix& u, const std::vector<size_t>& dead)
{
    const size_t m = u.nr, n = u.nc;
    std::vector<char> live(n, 1);
    for (size_t j : dead) live[j] = 0;
    for (size_t j : dead) {
        // residual norm^2 of e_i against the live columns is 1 - sum_k u(i,k)^2: take the largest
        size_t best = 0;
        double best_res = -1.0;
        for (size_t i = 0; i < m; ++i) {
            double s = 0.0;
            for (size_t k = 0; k < n; ++k)
                if (live[k]) s += u(i, k) * u(i, k);
            if (1.0 - s > best_res) {
                best_res = 1.0 - s;
                best = i;
            }
        }
        std::vector<double> v(m, 0.0);
        v[best] = 1.0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t k = 0; k < n; ++k) {
                if (!live[k]) continue;
                double dot = 0.0;
                for (size_t i = 0; i < m; ++i) dot += u(i, k) * v[i];
                for (size_t i = 0; i < m; ++i) v[i] -= dot * u(i, k);
            }
        double nrm = 0.0;
        for (size_t i = 0; i < m; ++i) nrm += v[i] * v[i];
        nrm = std::sqrt(nrm);
        for (size_t i = 0; i < m; ++i) u(i, j) = nrm > 0.0 ? v[i] / nrm : 0.0;
        live[j] = 1;
    }
}
} // namespace detail

inline SvdResult svd_thin(const Matrix& a_in)
{
    const bool flip = a_in.nr < a_in.nc;
    Matrix w = flip ? transpose(a_in) : a_in; // m >= n
    const size_t m = w.nr, n = w.nc;
    for (double x : w.a)
        if (!std::isfinite(x)) throw OracleError(ERR_INVALID_ARGUMENT, "SVD computation failed: non-finite input");
    Matrix v(n, n);
    for (size_t i = 0; i < n; ++i) v(i, i) = 1.0;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (size_t i = 0; i + 1 < n; ++i)
            for (size_t j = i + 1; j < n; ++j) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (size_t r = 0; r < m; ++r) {
                    alpha += w(r, i) * w(r, i);
                    beta += w(r, j) * w(r, j);
                    gamma += w(r, i) * w(r, j);
                }
                if (gamma == 0.0 || std::fabs(gamma) <= eps * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (size_t r = 0; r < m; ++r) {
                    const double x = w(r, i), y = w(r, j);
                    w(r, i) = c * x - s * y;
                    w(r, j) = s * x + c * y;
                }
                for (size_t r = 0; r < n; ++r) {
                    const double x = v(r, i), y = v(r, j);
                    v(r, i) = c * x - s * y;
                    v(r, j) = s * x + c * y;
                }
            }
        if (!rotated) break;
    }
    std::vector<double> sig(n);
    for (size_t j = 0; j < n; ++j) {
        double s2 = 0.0;
        for (size_t r = 0; r < m; ++r) s2 += w(r, j) * w(r, j);
        sig[j] = std::sqrt(s2);
    }
    std::vector<size_t> order(n);
    for (size_t j = 0; j < n; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return sig[x] > sig[y]; });
    Matrix u(m, n), vs(n, n);
    std::vector<double> s(n);
    std::vector<size_t> dead;
    for (size_t jj = 0; jj < n; ++jj) {
        const size_t j = order[jj];
        s[jj] = sig[j];
        if (sig[j] > 0.0)
            for (size_t r = 0; r < m; ++r) u(r, jj) = w(r, j) / sig[j];
        else
            dead.push_back(jj);
        for (size_t r = 0; r < n; ++r) vs(r, jj) = v(r, j);
    }
    if (!dead.empty()) detail::complete_orthonormal(u, dead);
    SvdResult out;
    out.s = s;
    if (!flip) {
        out.u = u;
        out.vt = transpose(vs);
    } else { // A^T = U S V^T  =>  A = V S U^T
        out.u = vs;
        out.vt = transpose(u);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// complete-pivoting LU facade: P A Q^T = L U with square factors — backend.rs:975-1039.
// Consumers read the permutations as "row k of P has its 1 in column row_perm[k]"
// (core/src/matrixluci/dense.rs:120-139).  Elimination order = rrlu_mut with zero tolerances
// (stops at an exactly zero pivot; the remaining Schur complement is then zero and L is completed
// with identity columns).
// ---------------------------------------------------------------------------------------------
struct FullPivLu {
    Matrix p, l, u, q;
};
inline FullPivLu full_piv_lu(const Matrix& a)
{
    if (a.nr != a.nc) throw OracleError(ERR_INVALID_ARGUMENT, "full_piv_lu expects a square matrix");
    const size_t n = a.nr;
    RrLUOptions o;
    o.rel_tol = 0.0;
    o.abs_tol = 0.0;
    o.left_orthogonal = true;
    RrLU lu = rrlu(a, o);
    const size_t r = lu.npivots();
    FullPivLu out;
    out.p = Matrix(n, n);
    out.q = Matrix(n, n);
    out.l = Matrix(n, n);
    out.u = Matrix(n, n);
    for (size_t k = 0; k < n; ++k) {
        out.p(k, lu.row_permutation[k]) = 1.0;
        out.q(k, lu.col_permutation[k]) = 1.0;
        out.l(k, k) = 1.0;
    }
    for (size_t j = 0; j < r; ++j)
        for (size_t i = 0; i < n; ++i) out.l(i, j) = lu.l(i, j);
    for (size_t i = 0; i < r; ++i)
        for (size_t j = 0; j < n; ++j) out.u(i, j) = lu.u(i, j);
    return out;
}

// ---------------------------------------------------------------------------------------------
// SimpleTensorTrain extras
// ---------------------------------------------------------------------------------------------
// traits.rs:289-354 — <tt|tt> by the O(chi^4) contraction, accumulation order (la, la_c, s) per entry.
inline double tt_norm2(const SimpleTensorTrain& tt)
{
    if (tt.tensors.empty()) return 0.0;
    const Tensor3& first = tt.tensors[0];
    size_t rd = first.r;
    std::vector<double> cur(rd * rd, 0.0);
    for (size_t s = 0; s < first.s; ++s)
        for (size_t ra = 0; ra < rd; ++ra)
            for (size_t rc = 0; rc < rd; ++rc) cur[ra * rd + rc] = cur[ra * rd + rc] + first.at(0, s, ra) * first.at(0, s, rc);
    for (size_t site = 1; site < tt.len(); ++site) {
        const Tensor3& t = tt.tensors[site];
        const size_t ld = t.l;
        rd = t.r;
        std::vector<double> nxt(rd * rd, 0.0);
        for (size_t la = 0; la < ld; ++la)
            for (size_t lc = 0; lc < ld; ++lc) {
                const double c = cur[la * ld + lc];
                for (size_t s = 0; s < t.s; ++s)
                    for (size_t ra = 0; ra < rd; ++ra)
                        for (size_t rc = 0; rc < rd; ++rc)
                            nxt[ra * rd + rc] = nxt[ra * rd + rc] + c * t.at(la, s, ra) * t.at(lc, s, rc);
            }
        cur.swap(nxt);
    }
    return std::sqrt(cur[0] * cur[0]);
}

// tensortrain.rs:374-421 — all values, leftmost site fastest
inline std::vector<double> tt_full_tensor(const SimpleTensorTrain& tt)
{
    std::vector<double> out;
    if (tt.tensors.empty()) return out;
    const size_t n = tt.len();
    MultiIndex idx(n, 0);
    for (const auto& t : tt.tensors)
        if (t.s == 0) return out;
    for (;;) {
        out.push_back(tt.evaluate(idx));
        bool carry = true;
        for (size_t i = 0; i < n && carry; ++i) {
            idx[i] += 1;
            if (idx[i] >= tt.tensors[i].s)
                idx[i] = 0;
            else
                carry = false;
        }
        if (carry) break;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// compression.rs
// ---------------------------------------------------------------------------------------------
enum class CompressionMethod { LU = 0, CI = 1, SVD = 2 }; // compression.rs:40-52
struct CompressionOptions { // :75-125
    CompressionMethod method = CompressionMethod::LU;
    double tolerance = 1e-12;
    size_t max_bond_dim = 0; // 0 == None
    bool normalize_error = true;
};

inline Matrix tensor3_to_left_matrix(const Tensor3& t) // :127-143
{
    Matrix m(t.l * t.s, t.r);
    for (size_t l = 0; l < t.l; ++l)
        for (size_t s = 0; s < t.s; ++s)
            for (size_t r = 0; r < t.r; ++r) m(l * t.s + s, r) = t.at(l, s, r);
    return m;
}
inline Matrix tensor3_to_right_matrix(const Tensor3& t) // :145-161
{
    Matrix m(t.l, t.s * t.r);
    for (size_t l = 0; l < t.l; ++l)
        for (size_t s = 0; s < t.s; ++s)
            for (size_t r = 0; r < t.r; ++r) m(l, s * t.r + r) = t.at(l, s, r);
    return m;
}

struct Factorized {
    Matrix left, right;
    size_t rank = 0;
};

inline Factorized factorize_svd(const Matrix& a, double tolerance, bool normalize_error, size_t max_bond_dim,
                                bool left_orthogonal) // :229-340
{
    const size_t m = a.nr, n = a.nc;
    if (m == 0 || n == 0) throw OracleError(ERR_INVALID_ARGUMENT, "Cannot factorize empty matrix");
    SvdResult d = svd_thin(a);
    const size_t min_dim = std::min(m, n);
    const double s_max = d.s.empty() ? 0.0 : d.s[0];
    const double threshold = normalize_error ? tolerance * s_max : tolerance;
    size_t rank = 0;
    for (size_t i = 0; i < min_dim && i < d.s.size(); ++i) {
        if (max_bond_dim != 0 && rank >= max_bond_dim) break;
        if (d.s[i] < threshold) break;
        ++rank;
    }
    rank = std::max<size_t>(rank, 1);
    Factorized f;
    f.rank = rank;
    f.left = Matrix(m, rank);
    f.right = Matrix(rank, n);
    for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < rank; ++j) f.left(i, j) = left_orthogonal ? d.u(i, j) : d.u(i, j) * d.s[j];
    for (size_t i = 0; i < rank; ++i)
        for (size_t j = 0; j < n; ++j) f.right(i, j) = left_orthogonal ? d.s[i] * d.vt(i, j) : d.vt(i, j);
    return f;
}

inline Factorized factorize(const Matrix& a, CompressionMethod method, double tolerance, bool normalize_error,
                            size_t max_bond_dim, bool left_orthogonal) // :165-227
{
    double reltol, abstol;
    if (tolerance > 0.0 && !normalize_error) {
        reltol = 0.0;
        abstol = tolerance;
    } else if (tolerance > 0.0) {
        reltol = tolerance;
        abstol = 0.0;
    } else {
        reltol = 1e-14;
        abstol = 0.0;
    }
    RrLUOptions o;
    o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
    o.rel_tol = reltol;
    o.abs_tol = abstol;
    o.left_orthogonal = left_orthogonal;
    Factorized f;
    switch (method) {
    case CompressionMethod::LU: {
        RrLU lu = rrlu(a, o);
        f.left = lu.left(true);
        f.right = lu.right(true);
        f.rank = lu.npivots();
        return f;
    }
    case CompressionMethod::CI: {
        MatrixLuciFactors l = matrix_luci_factors_from_matrix(a, o);
        f.left = l.left;
        f.right = l.right;
        f.rank = l.rank;
        return f;
    }
    case CompressionMethod::SVD:
        return factorize_svd(a, tolerance, normalize_error, max_bond_dim, left_orthogonal);
    }
    throw OracleError(ERR_INVALID_ARGUMENT, "unknown compression method");
}

inline void compress(SimpleTensorTrain& tt, const CompressionOptions& opt) // :375-507
{
    const size_t n = tt.len();
    if (n <= 1) return;
    auto& ts = tt.tensors;
    for (size_t ell = 0; ell + 1 < n; ++ell) {
        const size_t ld = ts[ell].l, sd = ts[ell].s;
        Factorized f = factorize(tensor3_to_left_matrix(ts[ell]), opt.method, 0.0, true, 0, true);
        Tensor3 nt(ld, sd, f.rank);
        for (size_t l = 0; l < ld; ++l)
            for (size_t s = 0; s < sd; ++s)
                for (size_t r = 0; r < f.rank; ++r) nt.at(l, s, r) = f.left(l * sd + s, r);
        ts[ell] = nt;
        const size_t nsd = ts[ell + 1].s, nrd = ts[ell + 1].r;
        Matrix c = mat_mul(f.right, tensor3_to_right_matrix(ts[ell + 1]));
        Tensor3 nn(f.rank, nsd, nrd);
        for (size_t l = 0; l < f.rank; ++l)
            for (size_t s = 0; s < nsd; ++s)
                for (size_t r = 0; r < nrd; ++r) nn.at(l, s, r) = c(l, s * nrd + r);
        ts[ell + 1] = nn;
    }
    for (size_t ell = n - 1; ell >= 1; --ell) {
        const size_t sd = ts[ell].s, rd = ts[ell].r;
        Factorized f = factorize(tensor3_to_right_matrix(ts[ell]), opt.method, opt.tolerance, opt.normalize_error,
                                 opt.max_bond_dim, false);
        Tensor3 nt(f.rank, sd, rd);
        for (size_t l = 0; l < f.rank; ++l)
            for (size_t s = 0; s < sd; ++s)
                for (size_t r = 0; r < rd; ++r) nt.at(l, s, r) = f.right(l, s * rd + r);
        ts[ell] = nt;
        const size_t pld = ts[ell - 1].l, psd = ts[ell - 1].s;
        Matrix c = mat_mul(tensor3_to_left_matrix(ts[ell - 1]), f.left);
        Tensor3 np(pld, psd, f.rank);
        for (size_t l = 0; l < pld; ++l)
            for (size_t s = 0; s < psd; ++s)
                for (size_t r = 0; r < f.rank; ++r) np.at(l, s, r) = c(l * psd + s, r);
        ts[ell - 1] = np;
    }
}

// ---------------------------------------------------------------------------------------------
// TTCache — cache.rs (single site index per site; the fused multi-index form is out of scope)
// ---------------------------------------------------------------------------------------------
struct TTCache {
    std::vector<Tensor3> tensors;
    std::vector<std::map<MultiIndex, std::vector<double>>> cache_left, cache_right;

    explicit TTCache(const SimpleTensorTrain& tt) : tensors(tt.tensors)
    {
        cache_left.resize(tensors.size());
        cache_right.resize(tensors.size());
    }
    size_t len() const { return tensors.size(); }

    static std::vector<double> slice_site(const Tensor3& t, size_t s) // types.rs:129-139 (l fastest)
    {
        std::vector<double> v;
        v.reserve(t.l * t.r);
        for (size_t r = 0; r < t.r; ++r)
            for (size_t l = 0; l < t.l; ++l) v.push_back(t.at(l, s, r));
        return v;
    }
    void validate(size_t start, const MultiIndex& idx) const
    {
        if (start + idx.size() > len()) throw OracleError(ERR_INVALID_ARGUMENT, "index length mismatch");
        for (size_t k = 0; k < idx.size(); ++k)
            if (idx[k] >= tensors[start + k].s) throw OracleError(ERR_INVALID_ARGUMENT, "index out of bounds");
    }
    // einsum_helper.rs:192-230
    static std::vector<double> row_vector_times_matrix(const std::vector<double>& v, const std::vector<double>& mat,
                                                       size_t rows, size_t cols)
    {
        std::vector<double> out(cols);
        for (size_t c = 0; c < cols; ++c) {
            double acc = 0.0;
            for (size_t r = 0; r < rows; ++r) acc = acc + v[r] * mat[r + c * rows];
            out[c] = acc;
        }
        return out;
    }
    // einsum_helper.rs:232-268
    static std::vector<double> matrix_times_col_vector(const std::vector<double>& mat, size_t rows, size_t cols,
                                                       const std::vector<double>& v)
    {
        std::vector<double> out(rows);
        for (size_t r = 0; r < rows; ++r) {
            double acc = 0.0;
            for (size_t c = 0; c < cols; ++c) acc = acc + mat[r + c * rows] * v[c];
            out[r] = acc;
        }
        return out;
    }
    std::vector<double> evaluate_left(const MultiIndex& idx) // cache.rs:430-467
    {
        const size_t ell = idx.size();
        validate(0, idx);
        if (ell == 0) return {1.0};
        auto it = cache_left[ell - 1].find(idx);
        if (it != cache_left[ell - 1].end()) return it->second;
        std::vector<double> res;
        if (ell == 1) {
            res = slice_site(tensors[0], idx[0]);
        } else {
            std::vector<double> left = evaluate_left(MultiIndex(idx.begin(), idx.end() - 1));
            const Tensor3& t = tensors[ell - 1];
            res = row_vector_times_matrix(left, slice_site(t, idx[ell - 1]), t.l, t.r);
        }
        cache_left[ell - 1][idx] = res;
        return res;
    }
    std::vector<double> evaluate_right(const MultiIndex& idx) // cache.rs:469-518
    {
        const size_t n = len(), ell = idx.size();
        if (ell > n) throw OracleError(ERR_INVALID_ARGUMENT, "index length mismatch");
        if (ell == 0) return {1.0};
        const size_t start = n - ell;
        validate(start, idx);
        auto it = cache_right[start].find(idx);
        if (it != cache_right[start].end()) return it->second;
        std::vector<double> res;
        if (ell == 1) {
            res = slice_site(tensors[n - 1], idx[0]);
        } else {
            std::vector<double> right = evaluate_right(MultiIndex(idx.begin() + 1, idx.end()));
            const Tensor3& t = tensors[start];
            res = matrix_times_col_vector(slice_site(t, idx[0]), t.l, t.r, right);
        }
        cache_right[start][idx] = res;
        return res;
    }
    double evaluate(const MultiIndex& idx) // cache.rs:520-556
    {
        const size_t n = len();
        if (idx.size() != n) throw OracleError(ERR_INVALID_ARGUMENT, "index length mismatch");
        if (n == 0) throw OracleError(ERR_INVALID_ARGUMENT, "empty tensor train");
        const size_t mid = n / 2;
        std::vector<double> l = evaluate_left(MultiIndex(idx.begin(), idx.begin() + mid));
        std::vector<double> r = evaluate_right(MultiIndex(idx.begin() + mid, idx.end()));
        if (l.size() != r.size()) throw OracleError(ERR_INTERNAL, "Left/right shape mismatch");
        double acc = 0.0;
        for (size_t i = 0; i < l.size(); ++i) acc = acc + l[i] * r[i];
        return acc;
    }
    size_t find_split_heuristic(const std::vector<MultiIndex>& indices) const // cache.rs:690-744
    {
        const size_t n = len();
        if (n <= 1) return std::max<size_t>(n, 1);
        auto cost = [&](size_t split) -> size_t {
            if (split == 0 || split >= n) return std::numeric_limits<size_t>::max();
            std::set<MultiIndex> l, r;
            for (const auto& idx : indices) {
                l.insert(MultiIndex(idx.begin(), idx.begin() + split));
                r.insert(MultiIndex(idx.begin() + split, idx.end()));
            }
            return l.size() + r.size();
        };
        const size_t cand[3] = {n / 4, n / 2, n * 3 / 4};
        bool have = false;
        size_t best_p = 0, best_c = 0;
        for (size_t p : cand) {
            if (p < 1 || p >= n) continue;
            const size_t c = cost(p);
            if (!have || c < best_c) { // min_by_key keeps the first minimum
                have = true;
                best_p = p;
                best_c = c;
            }
        }
        if (!have) throw OracleError(ERR_INTERNAL, "cache heuristic could not choose a valid split");
        return best_p;
    }
    std::vector<double> evaluate_many(const std::vector<MultiIndex>& indices, size_t split_or_zero) // :558-688
    {
        std::vector<double> out;
        if (indices.empty()) return out;
        const size_t n = len();
        if (n == 0) throw OracleError(ERR_INVALID_ARGUMENT, "empty tensor train");
        for (const auto& idx : indices) {
            if (idx.size() != n) throw OracleError(ERR_INVALID_ARGUMENT, "index length mismatch");
            validate(0, idx);
        }
        const size_t split = split_or_zero ? split_or_zero : find_split_heuristic(indices);
        if (split == 0 || split > n) throw OracleError(ERR_INVALID_ARGUMENT, "Invalid split position");
        out.reserve(indices.size());
        for (const auto& idx : indices) {
            std::vector<double> l = evaluate_left(MultiIndex(idx.begin(), idx.begin() + split));
            std::vector<double> r = evaluate_right(MultiIndex(idx.begin() + split, idx.end()));
            double acc = 0.0;
            for (size_t i = 0; i < l.size() && i < r.size(); ++i) acc = acc + l[i] * r[i];
            out.push_back(acc);
        }
        return out;
    }
};

// ---------------------------------------------------------------------------------------------
// tensorci2_from_tensor_train — conversion.rs
// ---------------------------------------------------------------------------------------------
struct FromTensorTrainOptions { // :20-36
    double tolerance = 1e-12;
    size_t max_bond_dim = 0; // 0 == None
    size_t max_iter = 3;
};

namespace detail {
// conversion.rs:273-284 (group_indices): forward != next -> left matrix, else right matrix
inline Matrix group_indices(const Tensor3& t, bool forward, bool next)
{
    return (forward != next) ? tensor3_to_left_matrix(t) : tensor3_to_right_matrix(t);
}
// conversion.rs:286-351
inline Tensor3 split_indices(const Matrix& m, size_t ld, size_t sd, size_t rd, size_t bond, bool forward, bool next)
{
    if (forward != next) {
        if (m.nr != ld * sd || m.nc != bond) throw OracleError(ERR_INVALID_ARGUMENT, "cannot reshape conversion matrix");
        Tensor3 t(ld, sd, bond);
        for (size_t r = 0; r < bond; ++r)
            for (size_t l = 0; l < ld; ++l)
                for (size_t s = 0; s < sd; ++s) t.at(l, s, r) = m(l * sd + s, r);
        return t;
    }
    if (m.nr != bond || m.nc != sd * rd) throw OracleError(ERR_INVALID_ARGUMENT, "cannot reshape conversion matrix");
    Tensor3 t(bond, sd, rd);
    for (size_t l = 0; l < bond; ++l)
        for (size_t s = 0; s < sd; ++s)
            for (size_t r = 0; r < rd; ++r) t.at(l, s, r) = m(l, s * rd + r);
    return t;
}
inline std::vector<MultiIndex> select_multi(const std::vector<MultiIndex>& set, const std::vector<size_t>& pos)
{
    std::vector<MultiIndex> r;
    for (size_t p : pos) {
        if (p >= set.size()) throw OracleError(ERR_INVALID_ARGUMENT, "conversion selected index out of range");
        r.push_back(set[p]);
    }
    return r;
}
// conversion.rs:123-205 + sweep_pair :207-271
inline void sweep1site_get_indices(SimpleTensorTrain& tt, bool forward, std::vector<std::vector<MultiIndex>>* spectators,
                                   const FromTensorTrainOptions& opt, std::vector<std::vector<MultiIndex>>& index_set,
                                   std::vector<double>& pivot_errors)
{
    const size_t n = tt.len();
    index_set.assign(1, std::vector<MultiIndex>(1, MultiIndex()));
    size_t rank = 0;
    for (size_t i = 0; i + 1 < n; ++i) rank = std::max(rank, tt.tensors[i].r);
    if (n > 1 && rank == 0) rank = 1; // traits.rs:116-123 rank() of a product state is 1
    pivot_errors.assign(rank + 1, 0.0);
    for (size_t step = 0; step + 1 < n; ++step) {
        const size_t site = forward ? step : n - step - 1;
        const size_t next_site = forward ? site + 1 : site - 1;
        Tensor3& cur = tt.tensors[site];
        Tensor3& nxt = tt.tensors[next_site];
        const size_t cl = cur.l, cs = cur.s, cr = cur.r, nl = nxt.l, ns = nxt.s, nr = nxt.r;
        RrLUOptions o;
        o.max_bond_dim = opt.max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : opt.max_bond_dim;
        o.rel_tol = opt.tolerance;
        o.abs_tol = 0.0;
        o.left_orthogonal = forward;
        MatrixLuciFactors f = matrix_luci_factors_from_matrix(group_indices(cur, forward, false), o);
        const size_t r = f.rank;
        const std::vector<MultiIndex>& base = index_set.back();
        std::vector<MultiIndex> cand;
        if (forward) {
            for (const auto& b : base)
                for (size_t loc = 0; loc < cs; ++loc) {
                    MultiIndex x = b;
                    x.push_back(loc);
                    cand.push_back(x);
                }
            index_set.push_back(select_multi(cand, f.row_indices));
            if (spectators) (*spectators)[site] = select_multi((*spectators)[site], f.col_indices);
            Matrix upd = mat_mul(f.right, group_indices(nxt, forward, true));
            cur = split_indices(f.left, cl, cs, cr, r, forward, false);
            nxt = split_indices(upd, nl, ns, nr, r, forward, true);
        } else {
            for (size_t loc = 0; loc < cs; ++loc)
                for (const auto& b : base) {
                    MultiIndex x;
                    x.push_back(loc);
                    x.insert(x.end(), b.begin(), b.end());
                    cand.push_back(x);
                }
            index_set.push_back(select_multi(cand, f.col_indices));
            if (spectators) (*spectators)[site] = select_multi((*spectators)[site], f.row_indices);
            Matrix upd = mat_mul(group_indices(nxt, forward, true), f.left);
            cur = split_indices(f.right, cl, cs, cr, r, forward, false);
            nxt = split_indices(upd, nl, ns, nr, r, forward, true);
        }
        if (pivot_errors.size() < f.pivot_errors.size()) pivot_errors.resize(f.pivot_errors.size(), 0.0);
        for (size_t k = 0; k < f.pivot_errors.size(); ++k) pivot_errors[k] = std::fmax(pivot_errors[k], f.pivot_errors[k]);
    }
    if (!forward) std::reverse(index_set.begin(), index_set.end());
}
} // namespace detail

inline TensorCI2 tensorci2_from_tensor_train(SimpleTensorTrain tt, const FromTensorTrainOptions& opt) // :66-121
{
    if (!std::isfinite(opt.tolerance) || opt.tolerance < 0.0)
        throw OracleError(ERR_INVALID_ARGUMENT, "TensorCI2 conversion tolerance must be finite and nonnegative");
    if (opt.max_iter < 2) throw OracleError(ERR_INVALID_ARGUMENT, "TensorCI2 conversion max_iter must be at least 2");
    if (tt.len() < 2)
        throw OracleError(ERR_INVALID_ARGUMENT, "TensorCI2 conversion requires at least 2 tensor-train sites");
    std::vector<size_t> local_dims;
    for (const auto& t : tt.tensors) local_dims.push_back(t.s);
    std::vector<std::vector<MultiIndex>> i_set, j_set, tmp;
    std::vector<double> pivot_errors, errs;
    detail::sweep1site_get_indices(tt, true, nullptr, opt, i_set, errs);
    detail::sweep1site_get_indices(tt, false, nullptr, opt, j_set, pivot_errors);
    for (size_t iter = 3; iter <= opt.max_iter; ++iter) {
        if (iter % 2 == 1) {
            std::vector<std::vector<MultiIndex>> filtered = j_set;
            detail::sweep1site_get_indices(tt, true, &filtered, opt, tmp, errs);
            j_set = filtered;
            pivot_errors = errs;
            if (tmp == i_set) break;
            i_set = tmp;
        } else {
            std::vector<std::vector<MultiIndex>> filtered = i_set;
            detail::sweep1site_get_indices(tt, false, &filtered, opt, tmp, errs);
            i_set = filtered;
            pivot_errors = errs;
            if (tmp == j_set) break;
            j_set = tmp;
        }
    }
    TensorCI2 tci(local_dims);
    tci.i_set = i_set;
    tci.j_set = j_set;
    tci.site_tensors = tt.tensors;
    tci.pivot_errors = pivot_errors;
    tci.bond_errors.assign(local_dims.size() - 1, 0.0);
    double mx = 0.0;
    for (const auto& t : tci.site_tensors)
        for (double v : t.d) mx = std::fmax(mx, std::sqrt(v * v));
    tci.max_sample_value = mx;
    return tci;
}

// ---------------------------------------------------------------------------------------------
// SimpleTensorTrain arithmetic — tensor4all-simplett/src/arithmetic.rs:34-180, tensortrain.rs:264-345, :449-583
// ---------------------------------------------------------------------------------------------
inline SimpleTensorTrain tt_scale(const SimpleTensorTrain& a, double factor) // scale_mut / scale: the LAST core carries it
{
    SimpleTensorTrain r = a;
    if (!r.tensors.empty())
        for (double& v : r.tensors.back().d) v = v * factor;
    return r;
}

inline SimpleTensorTrain tt_add(const SimpleTensorTrain& a, const SimpleTensorTrain& b) // arithmetic.rs:34-159 (direct sum)
{
    if (a.len() != b.len()) throw OracleError(ERR_INVALID_ARGUMENT, "Cannot add tensor trains of different lengths");
    if (a.len() == 0) return b;
    const size_t n = a.len();
    SimpleTensorTrain r;
    for (size_t i = 0; i < n; ++i) {
        const Tensor3& x = a.tensors[i];
        const Tensor3& y = b.tensors[i];
        if (x.s != y.s) throw OracleError(ERR_INVALID_ARGUMENT, "Site dimensions mismatch");
        const bool first = i == 0, last = i == n - 1;
        const size_t l0 = first ? 0 : x.l, r0 = last ? 0 : x.r; // offsets of b's block
        Tensor3 t(first ? 1 : x.l + y.l, x.s, last ? 1 : x.r + y.r);
        if (first && last) {
            for (size_t q = 0; q < x.s; ++q) t.at(0, q, 0) = x.at(0, q, 0) + y.at(0, q, 0);
        } else {
            for (size_t rr = 0; rr < x.r; ++rr)
                for (size_t q = 0; q < x.s; ++q)
                    for (size_t l = 0; l < x.l; ++l) t.at(l, q, rr) = x.at(l, q, rr);
            for (size_t rr = 0; rr < y.r; ++rr)
                for (size_t q = 0; q < y.s; ++q)
                    for (size_t l = 0; l < y.l; ++l) t.at(l0 + l, q, r0 + rr) = y.at(l, q, rr);
        }
        r.tensors.push_back(std::move(t));
    }
    return r; // from_tensors_unchecked
}

inline SimpleTensorTrain tt_sub(const SimpleTensorTrain& a, const SimpleTensorTrain& b) { return tt_add(a, tt_scale(b, -1.0)); } // :161-175

// contraction.rs:82-186 inner_product: the environment ("ij,isk,jsl->kl") is tenferro einsum in the reference (contraction
// order backend-defined); restated as env^T * A followed by (env^T A)^T * B, k ascending
inline double tt_inner_product(const SimpleTensorTrain& a, const SimpleTensorTrain& b)
{
    if (a.len() != b.len()) throw OracleError(ERR_INVALID_ARGUMENT, "Cannot compute inner_product product of tensor trains with different lengths");
    if (a.len() == 0) return 0.0;
    Matrix env(1, 1);
    env(0, 0) = 1.0;
    for (size_t i = 0; i < a.len(); ++i) {
        const Tensor3& x = a.tensors[i];
        const Tensor3& y = b.tensors[i];
        if (x.s != y.s) throw OracleError(ERR_INVALID_ARGUMENT, "Site dimensions mismatch");
        const Matrix tmp = mat_mul(transpose(env), Matrix(x.l, x.s * x.r, x.d.data())); // (L_b) x (S R_a)
        env = mat_mul(transpose(Matrix(y.l * x.s, x.r, tmp.a.data())), Matrix(y.l * y.s, y.r, y.d.data()));
    }
    return env(0, 0);
}

inline SimpleTensorTrain tt_reverse(const SimpleTensorTrain& a) // tensortrain.rs:327-345
{
    SimpleTensorTrain r;
    for (size_t i = a.len(); i-- > 0;) {
        const Tensor3& x = a.tensors[i];
        Tensor3 t(x.r, x.s, x.l);
        for (size_t l = 0; l < x.l; ++l)
            for (size_t q = 0; q < x.s; ++q)
                for (size_t rr = 0; rr < x.r; ++rr) t.at(rr, q, l) = x.at(l, q, rr);
        r.tensors.push_back(std::move(t));
    }
    return r;
}

inline SimpleTensorTrain tt_partial_sum(const SimpleTensorTrain& a, const std::vector<size_t>& dims) // tensortrain.rs:449-583
{
    const size_t n = a.len();
    if (n == 0) return SimpleTensorTrain();
    for (size_t d : dims)
        if (d >= n) throw OracleError(ERR_INVALID_ARGUMENT, "Dimension out of range");
    std::vector<Tensor3> out;
    Matrix tprod(1, 1);
    tprod(0, 0) = 1.0;
    for (size_t site = 0; site < n; ++site) {
        const Tensor3& t = a.tensors[site];
        if (std::find(dims.begin(), dims.end(), site) != dims.end()) {
            Matrix ss(t.l, t.r);
            for (size_t l = 0; l < t.l; ++l)
                for (size_t r = 0; r < t.r; ++r) {
                    double acc = 0.0;
                    for (size_t q = 0; q < t.s; ++q) acc = acc + t.at(l, q, r);
                    ss(l, r) = acc;
                }
            tprod = mat_mul(tprod, ss);
        } else {
            const Matrix prod = mat_mul(tprod, Matrix(t.l, t.s * t.r, t.d.data()));
            Tensor3 nt(tprod.nr, t.s, t.r);
            nt.d = prod.a;
            out.push_back(std::move(nt));
            tprod = Matrix(t.r, t.r);
            for (size_t i = 0; i < t.r; ++i) tprod(i, i) = 1.0;
        }
    }
    if (out.empty()) {
        Tensor3 t(1, 1, 1);
        t.d[0] = tprod(0, 0);
        return SimpleTensorTrain::make({t});
    }
    Tensor3& last = out.back();
    const Matrix c = mat_mul(Matrix(last.l * last.s, last.r, last.d.data()), tprod);
    Tensor3 nl(last.l, last.s, tprod.nc);
    nl.d = c.a;
    last = std::move(nl);
    return SimpleTensorTrain::make(std::move(out));
}

} // namespace t4a_oracle
