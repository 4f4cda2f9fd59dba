// oracle/t4a_oracle_tensor.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp).
// CPU restatement of the dense part of the dynamic-index tensor layer, crates/tensor4all-core/src:
//   defaults/idx_tensor.rs  unfold_split_inner :5278-5345 (left indices first, remaining ones in tensor order, column-major
//                           (m, n) matrix)
//   index_ops.rs            prepare_contraction :660-696 (all common indices are contracted; result = lhs free indices, then
//                           rhs free indices, each in operand order)
//   defaults/contract.rs    contract_pair :334-343 (dense operands: permute + one matrix product)
//   defaults/svd.rs         compute_retained_rank :150-211 (threshold scale x measure x rule), svd_truncated_inner :255-338,
//                           svd_with :347-395 (U [left.., bond], S diagonal, V [right.., bond] = conj(V^H) permuted)
//   defaults/qr.rs          compute_retained_rank_qr_from_dense :74-117 (row norms of R against rtol * max row norm;
//                           the LEADING r rows are kept), qr_with :206-328
//   defaults/factorize.rs   factorize :86-118 over FactorizeAlg {SVD, QR, LU, CI} x Canonical {Left, Right}: SVD :482-559 (S absorbed
//                           into the right / left factor), QR :561-622, LU :624-733 (rrLU, rel_tol 1e-14, permuted L and U),
//                           CI :735-834 (MatrixLUCI factors), the *_full_rank variants (no truncation, rel_tol 0)
//   truncation.rs           SvdTruncationPolicy :137-147 (default: relative, per value, 1e-12 — svd.rs:80-87)
// Indices are integer labels here (no prime levels / tags / structured storage / AD).  The factorisations themselves go
// through tenferro in the reference ("parity unpinned"): restated with the one-sided Jacobi SVD and Householder QR of
// t4a_oracle_tt.hpp, so U / V / Q / R agree with the reference up to the usual sign / rotation freedom and with the
// device to rounding.
#pragma once

#include "t4a_oracle_tt.hpp"

namespace t4a_oracle {

struct DenseTensor {
    std::vector<size_t> dims;
    std::vector<int64_t> labels;
    std::vector<double> data; // column-major
    size_t size() const
    {
        size_t n = 1;
        for (size_t d : dims) n *= d;
        return n;
    }
};

// out index k takes input index perm[k]
inline DenseTensor tensor_permute(const DenseTensor& t, const std::vector<size_t>& perm)
{
    const size_t r = t.dims.size();
    DenseTensor o;
    o.dims.resize(r);
    o.labels.resize(r);
    std::vector<size_t> in_stride(r, 1);
    for (size_t k = 1; k < r; ++k) in_stride[k] = in_stride[k - 1] * t.dims[k - 1];
    for (size_t k = 0; k < r; ++k) {
        o.dims[k] = t.dims[perm[k]];
        o.labels[k] = t.labels[perm[k]];
    }
    o.data.resize(t.size());
    std::vector<size_t> ctr(r, 0);
    for (size_t e = 0; e < o.data.size(); ++e) {
        size_t src = 0;
        for (size_t k = 0; k < r; ++k) src += ctr[k] * in_stride[perm[k]];
        o.data[e] = t.data[src];
        for (size_t k = 0; k < r; ++k) {
            if (++ctr[k] < o.dims[k]) break;
            ctr[k] = 0;
        }
    }
    return o;
}

inline void tensor_validate(const DenseTensor& t)
{
    if (t.dims.size() != t.labels.size()) throw OracleError(ERR_INVALID_ARGUMENT, "dims / labels length mismatch");
    for (size_t a = 0; a < t.labels.size(); ++a)
        for (size_t b = a + 1; b < t.labels.size(); ++b)
            if (t.labels[a] == t.labels[b]) throw OracleError(ERR_INVALID_ARGUMENT, "duplicate index in tensor");
    if (t.data.size() != t.size()) throw OracleError(ERR_INVALID_ARGUMENT, "data length does not match dims");
}

// contract_pair
inline DenseTensor tensor_contract_pair(const DenseTensor& a, const DenseTensor& b)
{
    tensor_validate(a);
    tensor_validate(b);
    std::vector<size_t> axes_a, axes_b;
    for (size_t i = 0; i < a.labels.size(); ++i)
        for (size_t j = 0; j < b.labels.size(); ++j)
            if (a.labels[i] == b.labels[j]) {
                if (a.dims[i] != b.dims[j]) throw OracleError(ERR_INVALID_ARGUMENT, "dimension mismatch of a common index");
                axes_a.push_back(i);
                axes_b.push_back(j);
            }
    std::vector<size_t> pa, pb;
    for (size_t i = 0; i < a.labels.size(); ++i)
        if (std::find(axes_a.begin(), axes_a.end(), i) == axes_a.end()) pa.push_back(i);
    const size_t n_free_a = pa.size();
    for (size_t i : axes_a) pa.push_back(i);
    for (size_t j : axes_b) pb.push_back(j);
    for (size_t j = 0; j < b.labels.size(); ++j)
        if (std::find(axes_b.begin(), axes_b.end(), j) == axes_b.end()) pb.push_back(j);
    const DenseTensor ap = tensor_permute(a, pa), bp = tensor_permute(b, pb);
    size_t M = 1, K = 1, N = 1;
    for (size_t k = 0; k < n_free_a; ++k) M *= ap.dims[k];
    for (size_t k = n_free_a; k < ap.dims.size(); ++k) K *= ap.dims[k];
    for (size_t k = axes_b.size(); k < bp.dims.size(); ++k) N *= bp.dims[k];
    Matrix ma(M, K, ap.data.data()), mb(K, N, bp.data.data());
    Matrix mc = mat_mul(ma, mb);
    DenseTensor o;
    for (size_t k = 0; k < n_free_a; ++k) {
        o.dims.push_back(ap.dims[k]);
        o.labels.push_back(ap.labels[k]);
    }
    for (size_t k = axes_b.size(); k < bp.dims.size(); ++k) {
        o.dims.push_back(bp.dims[k]);
        o.labels.push_back(bp.labels[k]);
    }
    o.data = mc.a;
    return o;
}

struct SvdPolicy { // truncation.rs:137-147
    double threshold = 1e-12;
    int scale = 0;   // 0 Relative, 1 Absolute
    int measure = 0; // 0 Value, 1 SquaredValue
    int rule = 0;    // 0 PerValue, 1 DiscardedTailSum
};

// svd.rs:150-211
inline size_t svd_retained_rank(const std::vector<double>& s, const SvdPolicy& p)
{
    if (s.empty()) return 1;
    std::vector<double> m(s.size());
    bool all_zero = true;
    for (size_t k = 0; k < s.size(); ++k) {
        m[k] = p.measure == 0 ? s[k] : s[k] * s[k];
        if (m[k] != 0.0) all_zero = false;
    }
    if (all_zero) return 1;
    size_t keep = 0;
    if (p.rule == 0) {
        if (p.scale == 0) {
            double ref = 0.0;
            for (double v : m) ref = std::max(ref, v);
            while (keep < m.size() && ref > 0.0 && m[keep] / ref > p.threshold) ++keep;
        } else {
            while (keep < m.size() && m[keep] > p.threshold) ++keep;
        }
    } else {
        double total = 0.0;
        for (double v : m) total += v;
        if (p.scale == 0 && total == 0.0) return 1;
        double discarded = 0.0;
        keep = m.size();
        for (size_t i = m.size(); i-- > 0;) {
            const bool ok = p.scale == 0 ? (discarded + m[i]) / total <= p.threshold : discarded + m[i] <= p.threshold;
            if (!ok) break;
            discarded += m[i];
            keep = i;
        }
    }
    return std::max<size_t>(keep, 1);
}

// qr.rs:74-117; r is k x n column-major
inline size_t qr_retained_rank(const std::vector<double>& r, size_t k, size_t n, double rtol)
{
    if (k == 0 || n == 0) return 1;
    const size_t md = std::min(k, n);
    std::vector<double> norms(md);
    double mx = 0.0;
    for (size_t i = 0; i < md; ++i) {
        double sq = 0.0;
        for (size_t j = i; j < n; ++j) {
            const double v = std::fabs(r[i + j * k]);
            sq += v * v;
        }
        norms[i] = std::sqrt(sq);
        mx = std::max(mx, norms[i]);
    }
    if (mx == 0.0) return 1;
    const double thr = rtol * mx;
    size_t cnt = 0;
    for (double v : norms)
        if (v >= thr) ++cnt;
    return std::max<size_t>(cnt, 1);
}

struct UnfoldResult {
    Matrix m;
    std::vector<size_t> left_dims, right_dims;
    std::vector<int64_t> left_labels, right_labels;
};
inline UnfoldResult tensor_unfold_split(const DenseTensor& t, const std::vector<int64_t>& left)
{
    tensor_validate(t);
    const size_t rank = t.dims.size();
    if (!(rank >= 2)) throw OracleError(ERR_INVALID_ARGUMENT, "Tensor must have rank >= 2");
    if (!(left.size() > 0 && left.size() < rank))
        throw OracleError(ERR_INVALID_ARGUMENT, "Left indices must be a non-empty proper subset of tensor indices");
    std::vector<size_t> perm;
    for (size_t a = 0; a < left.size(); ++a) {
        auto it = std::find(t.labels.begin(), t.labels.end(), left[a]);
        if (it == t.labels.end()) throw OracleError(ERR_INVALID_ARGUMENT, "Index in left_inds not found in tensor");
        for (size_t b = 0; b < a; ++b)
            if (left[a] == left[b]) throw OracleError(ERR_INVALID_ARGUMENT, "Duplicate index in left_inds");
        perm.push_back((size_t)(it - t.labels.begin()));
    }
    for (size_t k = 0; k < rank; ++k)
        if (std::find(perm.begin(), perm.begin() + left.size(), k) == perm.begin() + left.size()) perm.push_back(k);
    DenseTensor u = tensor_permute(t, perm);
    UnfoldResult r;
    size_t m = 1, n = 1;
    for (size_t k = 0; k < rank; ++k) {
        if (k < left.size()) {
            m *= u.dims[k];
            r.left_dims.push_back(u.dims[k]);
            r.left_labels.push_back(u.labels[k]);
        } else {
            n *= u.dims[k];
            r.right_dims.push_back(u.dims[k]);
            r.right_labels.push_back(u.labels[k]);
        }
    }
    r.m = Matrix(m, n, u.data.data());
    return r;
}

struct TensorSvdResult {
    size_t rank = 0;
    std::vector<double> u, s, v; // u: [left.., r], v: [right.., r]
    std::vector<size_t> left_dims, right_dims;
};
// svd_with: truncate == false keeps k = min(m, n); max_bond_dim == 0 <=> None
inline TensorSvdResult tensor_svd(const DenseTensor& t, const std::vector<int64_t>& left, bool truncate, const SvdPolicy& policy,
                                  size_t max_bond_dim, bool has_max_bond_dim)
{
    if (truncate) {
        if (has_max_bond_dim && max_bond_dim == 0) throw OracleError(ERR_INVALID_ARGUMENT, "max_bond_dim must be positive");
        if (!std::isfinite(policy.threshold) || policy.threshold < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "Invalid SVD truncation threshold");
    }
    UnfoldResult un = tensor_unfold_split(t, left);
    const size_t m = un.m.nr, n = un.m.nc, k = std::min(m, n);
    SvdResult d = svd_thin(un.m);
    size_t r = k;
    if (truncate) {
        r = svd_retained_rank(d.s, policy);
        if (has_max_bond_dim) r = std::min(r, max_bond_dim);
        r = std::max<size_t>(r, 1);
    } else {
        r = std::max<size_t>(k, 1);
    }
    r = std::min(r, d.s.size());
    TensorSvdResult o;
    o.rank = r;
    o.left_dims = un.left_dims;
    o.right_dims = un.right_dims;
    o.u.assign(d.u.a.begin(), d.u.a.begin() + m * r);
    o.s.assign(d.s.begin(), d.s.begin() + r);
    o.v.resize(n * r);
    for (size_t j = 0; j < r; ++j)
        for (size_t i = 0; i < n; ++i) o.v[i + n * j] = d.vt(j, i);
    return o;
}

struct TensorQrResult {
    size_t rank = 0;
    std::vector<double> q, r; // q: [left.., rank], r: [rank, right..]
    std::vector<size_t> left_dims, right_dims;
};
inline TensorQrResult tensor_qr(const DenseTensor& t, const std::vector<int64_t>& left, bool truncate, double rtol)
{
    UnfoldResult un = tensor_unfold_split(t, left);
    const size_t m = un.m.nr, n = un.m.nc, k = std::min(m, n);
    QrResult d = qr_thin(un.m);
    size_t r = k;
    if (truncate) {
        if (!std::isfinite(rtol) || rtol < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "Invalid rtol value");
        r = qr_retained_rank(d.r.a, k, n, rtol);
    }
    r = std::min(r, k);
    TensorQrResult o;
    o.rank = r;
    o.left_dims = un.left_dims;
    o.right_dims = un.right_dims;
    o.q.assign(d.q.a.begin(), d.q.a.begin() + m * r);
    o.r.resize(r * n);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < r; ++i) o.r[i + r * j] = d.r(i, j);
    return o;
}

struct TensorFactorizeResult {
    size_t rank = 0;
    std::vector<double> left, right; // left: [left.., rank], right: [rank, right..]
    std::vector<double> singular_values; // SVD only
    std::vector<size_t> left_dims, right_dims;
};
// alg: 0 SVD, 1 QR, 2 LU, 3 CI; canonical: 0 Left, 1 Right
inline TensorFactorizeResult tensor_factorize(const DenseTensor& t, const std::vector<int64_t>& left, int alg, int canonical,
                                              bool full_rank, const SvdPolicy& policy, size_t max_bond_dim, bool has_max_bond_dim,
                                              double qr_rtol)
{
    TensorFactorizeResult o;
    if (alg == 0) {
        TensorSvdResult d = tensor_svd(t, left, !full_rank, policy, max_bond_dim, has_max_bond_dim && !full_rank);
        const size_t r = d.rank;
        size_t m = 1, n = 1;
        for (size_t x : d.left_dims) m *= x;
        for (size_t x : d.right_dims) n *= x;
        o.rank = r;
        o.left_dims = d.left_dims;
        o.right_dims = d.right_dims;
        o.singular_values = d.s;
        o.left.resize(m * r);
        o.right.resize(r * n);
        for (size_t j = 0; j < r; ++j)
            for (size_t i = 0; i < m; ++i) o.left[i + m * j] = canonical == 0 ? d.u[i + m * j] : d.u[i + m * j] * d.s[j];
        for (size_t j = 0; j < n; ++j)
            for (size_t i = 0; i < r; ++i) o.right[i + r * j] = canonical == 0 ? d.s[i] * d.v[j + n * i] : d.v[j + n * i];
        return o;
    }
    if (alg == 1) {
        TensorQrResult d = tensor_qr(t, left, !full_rank, qr_rtol);
        o.rank = d.rank;
        o.left = d.q;
        o.right = d.r;
        o.left_dims = d.left_dims;
        o.right_dims = d.right_dims;
        return o;
    }
    UnfoldResult un = tensor_unfold_split(t, left);
    RrLUOptions lo;
    lo.max_bond_dim = (full_rank || !has_max_bond_dim) ? std::numeric_limits<size_t>::max() : max_bond_dim;
    lo.rel_tol = full_rank ? 0.0 : 1e-14;
    lo.abs_tol = 0.0;
    lo.left_orthogonal = canonical == 0;
    o.left_dims = un.left_dims;
    o.right_dims = un.right_dims;
    if (alg == 2) {
        RrLU lu = rrlu(un.m, lo);
        o.rank = lu.npivots();
        o.left = lu.left(true).a;
        o.right = lu.right(true).a;
    } else if (alg == 3) {
        MatrixLuciFactors f = matrix_luci_factors_from_matrix(un.m, lo);
        o.rank = f.rank;
        o.left = f.left.a;
        o.right = f.right.a;
    } else {
        throw OracleError(ERR_INVALID_ARGUMENT, "unknown factorization algorithm");
    }
    if (o.rank == 0) throw OracleError(ERR_INVALID_ARGUMENT, "Failed to create bond index: dimension 0");
    return o;
}


// ---- N-ary contraction of a connected tensor network (tensor4all-core/src/defaults/contract.rs) ----
// Restated from the reference text, independently of the device path (which reduces the network pair by pair with GEMMs): validation
// and result indices as contract_with_options_impl :530-572 / build_contraction_plan :885-941 / find_tensor_connected_components_with_retained
// :1167-1230 prescribe them, the values by direct summation over every summed label (the reference hands the network to tenferro's
// einsum: backend-defined summation order, "parity unpinned" at the bit level; its own tests pin exact small-integer cases,
// contract/tests/mod.rs:173-196, :235-281, :338-392, :410-449).
struct NetworkContraction {
    DenseTensor result;
    size_t components = 1;
};
inline DenseTensor tensor_contract_network(const std::vector<DenseTensor>& ts, const std::vector<int64_t>& retain)
{
    if (ts.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "No tensors to contract");
    for (const DenseTensor& t : ts) tensor_validate(t);
    auto has = [](const DenseTensor& t, int64_t l) { return std::find(t.labels.begin(), t.labels.end(), l) != t.labels.end(); };
    for (int64_t r : retain) {
        bool found = false;
        for (const DenseTensor& t : ts) found = found || has(t, r);
        if (!found) throw OracleError(ERR_INVALID_ARGUMENT, "Retained index does not appear in the input tensors");
    }
    if (ts.size() == 1) return ts[0];
    const size_t n = ts.size();
    // connected components by depth-first search over "shares a label" (contractable or retained: both are common labels)
    std::vector<int> comp(n, -1);
    int ncomp = 0;
    for (size_t s0 = 0; s0 < n; ++s0) {
        if (comp[s0] >= 0) continue;
        std::vector<size_t> stack{s0};
        comp[s0] = ncomp;
        while (!stack.empty()) {
            const size_t u = stack.back();
            stack.pop_back();
            for (size_t v = 0; v < n; ++v) {
                if (comp[v] >= 0) continue;
                bool common = false;
                for (int64_t l : ts[u].labels) common = common || has(ts[v], l);
                if (common) {
                    comp[v] = ncomp;
                    stack.push_back(v);
                }
            }
        }
        ++ncomp;
    }
    if (ncomp > 1) throw OracleError(ERR_INVALID_ARGUMENT, "Disconnected tensor network: " + std::to_string(ncomp) + " components found");
    // labels in order of first appearance, their sizes and counts
    std::vector<int64_t> labels;
    std::vector<size_t> sizes, counts;
    for (const DenseTensor& t : ts)
        for (size_t a = 0; a < t.labels.size(); ++a) {
            const auto it = std::find(labels.begin(), labels.end(), t.labels[a]);
            if (it == labels.end()) {
                labels.push_back(t.labels[a]);
                sizes.push_back(t.dims[a]);
                counts.push_back(1);
            } else {
                const size_t k = (size_t)(it - labels.begin());
                if (sizes[k] != t.dims[a]) throw OracleError(ERR_INVALID_ARGUMENT, "Internal label shape mismatch");
                ++counts[k];
            }
        }
    std::vector<size_t> out_pos, sum_pos; // positions in `labels`
    for (size_t k = 0; k < labels.size(); ++k) {
        const bool retained = std::find(retain.begin(), retain.end(), labels[k]) != retain.end();
        (counts[k] == 1 || retained ? out_pos : sum_pos).push_back(k);
    }
    DenseTensor out;
    for (size_t k : out_pos) {
        out.dims.push_back(sizes[k]);
        out.labels.push_back(labels[k]);
    }
    out.data.assign(out.size(), 0.0);
    // operand axis -> position in `labels`, operand strides
    std::vector<std::vector<size_t>> where(n), stride(n);
    for (size_t t = 0; t < n; ++t) {
        size_t st = 1;
        for (size_t a = 0; a < ts[t].labels.size(); ++a) {
            where[t].push_back((size_t)(std::find(labels.begin(), labels.end(), ts[t].labels[a]) - labels.begin()));
            stride[t].push_back(st);
            st *= ts[t].dims[a];
        }
    }
    size_t n_sum = 1;
    for (size_t k : sum_pos) n_sum *= sizes[k];
    std::vector<size_t> val(labels.size(), 0);
    for (size_t o = 0; o < out.data.size(); ++o) {
        size_t rem = o;
        for (size_t k : out_pos) {
            val[k] = rem % sizes[k];
            rem /= sizes[k];
        }
        double acc = 0.0;
        for (size_t q = 0; q < n_sum; ++q) {
            size_t r2 = q;
            for (size_t k : sum_pos) {
                val[k] = r2 % sizes[k];
                r2 /= sizes[k];
            }
            double prod = 1.0;
            for (size_t t = 0; t < n; ++t) {
                size_t off = 0;
                for (size_t a = 0; a < where[t].size(); ++a) off += val[where[t][a]] * stride[t][a];
                prod = prod * ts[t].data[off];
            }
            acc = acc + prod;
        }
        out.data[o] = acc;
    }
    return out;
}

} // namespace t4a_oracle
