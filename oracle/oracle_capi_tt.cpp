// oracle/oracle_capi_tt.cpp — TEST INFRASTRUCTURE ONLY (see t4a_oracle_tt.hpp header).
// C entry points for the tensor-train side of the CPU restatement (SURVEY.md §8 rows a14–a18).
#include "t4a_oracle_tt.hpp"
#include "t4a_oracle_tensor.hpp"
#include "t4a_oracle_aci.hpp"
#include "t4a_oracle_treeaci.hpp"
#include "t4a_oracle_search.hpp"

#include <cstring>
#include <memory>

using namespace t4a_oracle;

namespace {
thread_local std::string g_tt_error;

template <class F> int guarded(F&& body)
{
    try {
        body();
        return 0;
    } catch (const OracleError& e) {
        g_tt_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_tt_error = e.what();
        return ERR_INTERNAL;
    }
}

Matrix from_ptr(const double* a, size_t m, size_t n)
{
    Matrix x(m, n);
    if (m != 0 && n != 0) std::memcpy(x.a.data(), a, m * n * sizeof(double));
    return x;
}
void to_ptr(const Matrix& x, double* out)
{
    if (!x.a.empty()) std::memcpy(out, x.a.data(), x.a.size() * sizeof(double));
}

struct OracleTT {
    SimpleTensorTrain tt;
};
} // namespace

extern "C" {

const char* oracle_tt_last_error() { return g_tt_error.c_str(); }

int oracle_qr_f64(const double* a, uint64_t m, uint64_t n, double* q, double* r)
{
    return guarded([&] {
        QrResult d = qr_thin(from_ptr(a, m, n));
        to_ptr(d.q, q);
        to_ptr(d.r, r);
    });
}

int oracle_svd_f64(const double* a, uint64_t m, uint64_t n, double* u, double* s, double* vt)
{
    return guarded([&] {
        SvdResult d = svd_thin(from_ptr(a, m, n));
        to_ptr(d.u, u);
        for (size_t i = 0; i < d.s.size(); ++i) s[i] = d.s[i];
        to_ptr(d.vt, vt);
    });
}

int oracle_full_piv_lu_f64(const double* a, uint64_t n, double* p, double* l, double* u, double* q)
{
    return guarded([&] {
        FullPivLu d = full_piv_lu(from_ptr(a, n, n));
        to_ptr(d.p, p);
        to_ptr(d.l, l);
        to_ptr(d.u, u);
        to_ptr(d.q, q);
    });
}

// ---- SimpleTensorTrain handle ----
void* oracle_tt_new(uint64_t n_sites, const uint64_t* dims3 /* 3 x n_sites */, const double* data)
{
    void* out = nullptr;
    guarded([&] {
        std::vector<Tensor3> ts;
        size_t off = 0;
        for (size_t s = 0; s < n_sites; ++s) {
            Tensor3 t(dims3[3 * s], dims3[3 * s + 1], dims3[3 * s + 2]);
            if (!t.d.empty()) std::memcpy(t.d.data(), data + off, t.d.size() * sizeof(double));
            off += t.d.size();
            ts.push_back(std::move(t));
        }
        auto* h = new OracleTT;
        try {
            h->tt = SimpleTensorTrain::make(std::move(ts));
        } catch (...) {
            delete h;
            throw;
        }
        out = h;
    });
    return out;
}
void oracle_tt_release(void* h) { delete static_cast<OracleTT*>(h); }
uint64_t oracle_tt_len(void* h) { return static_cast<OracleTT*>(h)->tt.len(); }

int oracle_tt_dims(void* h, uint64_t* dims3)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        for (size_t s = 0; s < tt.len(); ++s) {
            dims3[3 * s] = tt.tensors[s].l;
            dims3[3 * s + 1] = tt.tensors[s].s;
            dims3[3 * s + 2] = tt.tensors[s].r;
        }
    });
}
int oracle_tt_site_tensor(void* h, uint64_t site, double* out)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        if (site >= tt.len()) throw OracleError(ERR_INVALID_ARGUMENT, "site out of range");
        const auto& t = tt.tensors[site];
        if (!t.d.empty()) std::memcpy(out, t.d.data(), t.d.size() * sizeof(double));
    });
}
int oracle_tt_evaluate(void* h, const uint64_t* idx /* n_sites x n_pts col-major */, uint64_t n_pts, double* out)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        const size_t n = tt.len();
        MultiIndex mi(n);
        for (size_t p = 0; p < n_pts; ++p) {
            for (size_t s = 0; s < n; ++s) mi[s] = idx[s + n * p];
            out[p] = tt.evaluate(mi);
        }
    });
}
int oracle_tt_sum(void* h, double* out)
{
    return guarded([&] { *out = static_cast<OracleTT*>(h)->tt.sum(); });
}
int oracle_tt_norm2(void* h, double* out)
{
    return guarded([&] { *out = tt_norm2(static_cast<OracleTT*>(h)->tt); });
}
int oracle_tt_full_tensor(void* h, double* out)
{
    return guarded([&] {
        std::vector<double> v = tt_full_tensor(static_cast<OracleTT*>(h)->tt);
        if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(double));
    });
}
int oracle_tt_compress(void* h, int method, double tolerance, uint64_t max_bond_dim, int normalize_error)
{
    return guarded([&] {
        if (method < 0 || method > 2) throw OracleError(ERR_INVALID_ARGUMENT, "unknown compression method");
        CompressionOptions o;
        o.method = (CompressionMethod)method;
        o.tolerance = tolerance;
        o.max_bond_dim = max_bond_dim;
        o.normalize_error = normalize_error != 0;
        compress(static_cast<OracleTT*>(h)->tt, o);
    });
}
// TTCache::evaluate_many on a fresh cache; split == 0 -> heuristic. *used_split reports the split.
int oracle_tt_evaluate_many(void* h, const uint64_t* idx, uint64_t n_pts, uint64_t split, double* out,
                            uint64_t* used_split)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        const size_t n = tt.len();
        std::vector<MultiIndex> pts(n_pts, MultiIndex(n));
        for (size_t p = 0; p < n_pts; ++p)
            for (size_t s = 0; s < n; ++s) pts[p][s] = idx[s + n * p];
        TTCache cache(tt);
        if (used_split) *used_split = (split || pts.empty()) ? split : cache.find_split_heuristic(pts);
        std::vector<double> v = cache.evaluate_many(pts, split);
        for (size_t p = 0; p < v.size(); ++p) out[p] = v[p];
    });
}

// tensorci2_from_tensor_train: returns the pieces of the resulting TensorCI2 through query calls.
struct OracleConv {
    std::unique_ptr<TensorCI2> tci;
};
void* oracle_tci2_from_tt(void* tt_h, double tolerance, uint64_t max_bond_dim, uint64_t max_iter)
{
    void* out = nullptr;
    guarded([&] {
        FromTensorTrainOptions o;
        o.tolerance = tolerance;
        o.max_bond_dim = max_bond_dim;
        o.max_iter = max_iter;
        auto* c = new OracleConv;
        try {
            c->tci.reset(new TensorCI2(tensorci2_from_tensor_train(static_cast<OracleTT*>(tt_h)->tt, o)));
        } catch (...) {
            delete c;
            throw;
        }
        out = c;
    });
    return out;
}
void oracle_conv_release(void* h) { delete static_cast<OracleConv*>(h); }
int oracle_conv_index_set(void* h, int which, uint64_t site, uint64_t* count, uint64_t* width, uint64_t* out)
{
    return guarded([&] {
        auto& t = *static_cast<OracleConv*>(h)->tci;
        const auto& set = which == 0 ? t.i_set.at(site) : t.j_set.at(site);
        const size_t w = which == 0 ? site : t.len() - site - 1;
        *count = set.size();
        *width = w;
        if (out)
            for (size_t k = 0; k < set.size(); ++k)
                for (size_t s = 0; s < w; ++s) out[s + w * k] = set[k][s];
    });
}
int oracle_conv_site_tensor(void* h, uint64_t site, uint64_t* dims3, double* out)
{
    return guarded([&] {
        auto& t = *static_cast<OracleConv*>(h)->tci;
        const auto& x = t.site_tensors.at(site);
        dims3[0] = x.l;
        dims3[1] = x.s;
        dims3[2] = x.r;
        if (out && !x.d.empty()) std::memcpy(out, x.d.data(), x.d.size() * sizeof(double));
    });
}
int oracle_conv_scalars(void* h, double* max_sample_value, uint64_t* n_pivot_errors, double* pivot_errors)
{
    return guarded([&] {
        auto& t = *static_cast<OracleConv*>(h)->tci;
        *max_sample_value = t.max_sample_value;
        *n_pivot_errors = t.pivot_errors.size();
        if (pivot_errors)
            for (size_t k = 0; k < t.pivot_errors.size(); ++k) pivot_errors[k] = t.pivot_errors[k];
    });
}

// ---- dense labelled tensors (t4a_oracle_tensor.hpp) ----
static DenseTensor make_tensor(const double* d, const uint64_t* dims, const int64_t* labels, uint64_t rank)
{
    DenseTensor t;
    t.dims.assign(dims, dims + rank);
    t.labels.assign(labels, labels + rank);
    t.data.assign(d, d + t.size());
    return t;
}
int oracle_tensor_contract(const double* a, const uint64_t* adims, const int64_t* alabels, uint64_t ra, const double* b,
                           const uint64_t* bdims, const int64_t* blabels, uint64_t rb, double* out, uint64_t* out_dims,
                           int64_t* out_labels, uint64_t* out_rank)
{
    return guarded([&] {
        DenseTensor o = tensor_contract_pair(make_tensor(a, adims, alabels, ra), make_tensor(b, bdims, blabels, rb));
        *out_rank = o.dims.size();
        for (size_t k = 0; k < o.dims.size(); ++k) {
            out_dims[k] = o.dims[k];
            out_labels[k] = o.labels[k];
        }
        if (out) std::copy(o.data.begin(), o.data.end(), out);
    });
}
// N-ary contraction: operands concatenated (data, dims, labels back to back; ranks[t] axes each).  out may be null (shape query).
int oracle_tensor_contract_many(uint64_t n_tensors, const double* data, const uint64_t* dims, const int64_t* labels, const uint64_t* ranks,
                                const int64_t* retain, uint64_t n_retain, double* out, uint64_t* out_dims, int64_t* out_labels,
                                uint64_t* out_rank)
{
    return guarded([&] {
        std::vector<DenseTensor> ts;
        size_t dpos = 0, apos = 0;
        for (size_t t = 0; t < n_tensors; ++t) {
            DenseTensor x = make_tensor(data + dpos, dims + apos, labels + apos, ranks[t]);
            dpos += x.size();
            apos += ranks[t];
            ts.push_back(std::move(x));
        }
        DenseTensor o = tensor_contract_network(ts, std::vector<int64_t>(retain, retain + n_retain));
        *out_rank = o.dims.size();
        for (size_t k = 0; k < o.dims.size(); ++k) {
            out_dims[k] = o.dims[k];
            out_labels[k] = o.labels[k];
        }
        if (out) std::copy(o.data.begin(), o.data.end(), out);
    });
}
uint64_t oracle_svd_retained_rank(const double* s, uint64_t n, double threshold, int scale, int measure, int rule)
{
    SvdPolicy p;
    p.threshold = threshold;
    p.scale = scale;
    p.measure = measure;
    p.rule = rule;
    return svd_retained_rank(std::vector<double>(s, s + n), p);
}
uint64_t oracle_qr_retained_rank(const double* r, uint64_t k, uint64_t n, double rtol)
{
    return qr_retained_rank(std::vector<double>(r, r + k * n), k, n, rtol);
}
// u: m x min(m,n) capacity, s: min(m,n), v: n x min(m,n) capacity
int oracle_tensor_svd(const double* t, const uint64_t* dims, const int64_t* labels, uint64_t rank, const int64_t* left,
                      uint64_t n_left, int truncate, double threshold, int scale, int measure, int rule, uint64_t max_bond_dim,
                      uint64_t* r_out, double* u, double* s, double* v)
{
    return guarded([&] {
        SvdPolicy p;
        p.threshold = threshold;
        p.scale = scale;
        p.measure = measure;
        p.rule = rule;
        TensorSvdResult o = tensor_svd(make_tensor(t, dims, labels, rank), std::vector<int64_t>(left, left + n_left), truncate != 0, p,
                                       max_bond_dim == (uint64_t)-1 ? 0 : max_bond_dim, max_bond_dim != (uint64_t)-1);
        *r_out = o.rank;
        std::copy(o.u.begin(), o.u.end(), u);
        std::copy(o.s.begin(), o.s.end(), s);
        std::copy(o.v.begin(), o.v.end(), v);
    });
}
int oracle_tensor_qr(const double* t, const uint64_t* dims, const int64_t* labels, uint64_t rank, const int64_t* left,
                     uint64_t n_left, int truncate, double rtol, uint64_t* r_out, double* q, double* r)
{
    return guarded([&] {
        TensorQrResult o = tensor_qr(make_tensor(t, dims, labels, rank), std::vector<int64_t>(left, left + n_left), truncate != 0, rtol);
        *r_out = o.rank;
        std::copy(o.q.begin(), o.q.end(), q);
        std::copy(o.r.begin(), o.r.end(), r);
    });
}

int oracle_tensor_factorize(const double* t, const uint64_t* dims, const int64_t* labels, uint64_t rank, const int64_t* left,
                            uint64_t n_left, int alg, int canonical, int full_rank, double threshold, int scale, int measure, int rule,
                            uint64_t max_bond_dim, double qr_rtol, uint64_t* r_out, double* lout, double* rout, double* sv)
{
    return guarded([&] {
        SvdPolicy p;
        p.threshold = threshold;
        p.scale = scale;
        p.measure = measure;
        p.rule = rule;
        TensorFactorizeResult o = tensor_factorize(make_tensor(t, dims, labels, rank), std::vector<int64_t>(left, left + n_left), alg,
                                                   canonical, full_rank != 0, p, max_bond_dim == (uint64_t)-1 ? 0 : max_bond_dim,
                                                   max_bond_dim != (uint64_t)-1, qr_rtol);
        *r_out = o.rank;
        std::copy(o.left.begin(), o.left.end(), lout);
        std::copy(o.right.begin(), o.right.end(), rout);
        if (sv) std::copy(o.singular_values.begin(), o.singular_values.end(), sv);
    });
}

// ---- SimpleTensorTrain arithmetic ----
void* oracle_tt_binary(void* a, void* b, int op /* 0 add, 1 sub */)
{
    OracleTT* h = nullptr;
    const int rc = guarded([&] {
        auto r = std::make_unique<OracleTT>();
        const SimpleTensorTrain& x = static_cast<OracleTT*>(a)->tt;
        const SimpleTensorTrain& y = static_cast<OracleTT*>(b)->tt;
        r->tt = op == 0 ? tt_add(x, y) : tt_sub(x, y);
        h = r.release();
    });
    return rc == 0 ? h : nullptr;
}
int oracle_tt_inner_product(void* a, void* b, double* out)
{
    return guarded([&] { *out = tt_inner_product(static_cast<OracleTT*>(a)->tt, static_cast<OracleTT*>(b)->tt); });
}
void* oracle_tt_unary(void* a, int op /* 0 scale, 1 reverse, 2 partial_sum */, double factor, const uint64_t* dims, uint64_t n_dims)
{
    OracleTT* h = nullptr;
    const int rc = guarded([&] {
        auto r = std::make_unique<OracleTT>();
        const SimpleTensorTrain& x = static_cast<OracleTT*>(a)->tt;
        if (op == 0) r->tt = tt_scale(x, factor);
        else if (op == 1) r->tt = tt_reverse(x);
        else r->tt = tt_partial_sum(x, std::vector<size_t>(dims, dims + n_dims));
        h = r.release();
    });
    return rc == 0 ? h : nullptr;
}

// ---- tensor4all-aci: elementwise / elementwise_batched (t4a_oracle_aci.hpp) ----
struct oracle_aci_options {
    uint64_t max_iters, min_iters;
    int32_t has_max_bond_dim;
    uint64_t max_bond_dim;
    double tolerance;
    int32_t scale_tolerance;
    uint64_t rng_seed;
    int32_t enable_global_guard;
    uint64_t nsearch_global_pivots, max_nglobal_pivot, nsweeps_global_search;
    double tol_margin_global_search;
};
typedef int (*oracle_aci_op_fn)(void* user, const double* values, uint64_t n_inputs, uint64_t n_points, double* out);

namespace {
struct OracleAci {
    AciResult res;
};
AciOp make_aci_op(int op_kind, oracle_aci_op_fn cb, void* user)
{
    if (op_kind == 1)
        return [](const double* v, size_t K, size_t np, double* out) {
            for (size_t p = 0; p < np; ++p) {
                double acc = v[K * p];
                for (size_t k = 1; k < K; ++k) acc = acc * v[k + K * p];
                out[p] = acc;
            }
        };
    if (op_kind == 2)
        return [](const double* v, size_t K, size_t np, double* out) {
            for (size_t p = 0; p < np; ++p) {
                double acc = v[K * p];
                for (size_t k = 1; k < K; ++k) acc = acc + v[k + K * p];
                out[p] = acc;
            }
        };
    if (!cb) throw OracleError(ERR_INVALID_ARGUMENT, "operator callback is null");
    return [cb, user](const double* v, size_t K, size_t np, double* out) {
        if (cb(user, v, K, np, out) != 0) throw OracleError(ERR_INVALID_ARGUMENT, "operator callback failed");
    };
}
AciOptions convert_aci(const oracle_aci_options* o, void* guess)
{
    AciOptions a;
    if (o) {
        a.max_iters = o->max_iters;
        a.min_iters = o->min_iters;
        a.has_max_bond_dim = o->has_max_bond_dim != 0;
        a.max_bond_dim = o->max_bond_dim;
        a.tolerance = o->tolerance;
        a.scale_tolerance = o->scale_tolerance != 0;
        a.rng_seed = o->rng_seed;
        a.enable_global_guard = o->enable_global_guard != 0;
        a.nsearch_global_pivots = o->nsearch_global_pivots;
        a.max_nglobal_pivot = o->max_nglobal_pivot;
        a.nsweeps_global_search = o->nsweeps_global_search;
        a.tol_margin_global_search = o->tol_margin_global_search;
    }
    if (guess) {
        a.has_initial_guess = true;
        a.initial_guess = static_cast<OracleTT*>(guess)->tt;
    }
    return a;
}
} // namespace

void* oracle_aci_elementwise(void* const* inputs, uint64_t n_inputs, int op_kind, oracle_aci_op_fn cb, void* user,
                             const oracle_aci_options* opts, void* initial_guess)
{
    OracleAci* h = nullptr;
    const int rc = guarded([&] {
        std::vector<SimpleTensorTrain> in;
        for (uint64_t k = 0; k < n_inputs; ++k) in.push_back(static_cast<OracleTT*>(inputs[k])->tt);
        auto r = std::make_unique<OracleAci>();
        r->res = elementwise_batched(make_aci_op(op_kind, cb, user), in, convert_aci(opts, initial_guess));
        h = r.release();
    });
    return rc == 0 ? h : nullptr;
}
void oracle_aci_release(void* h) { delete static_cast<OracleAci*>(h); }
void* oracle_aci_tensor_train(void* h)
{
    auto* t = new OracleTT;
    t->tt = static_cast<OracleAci*>(h)->res.tensor_train;
    return t;
}
uint64_t oracle_aci_n_iters(void* h) { return static_cast<OracleAci*>(h)->res.ranks.size(); }
int oracle_aci_termination(void* h) { return (int)static_cast<OracleAci*>(h)->res.termination; }
void oracle_aci_history(void* h, uint64_t* ranks, double* errors, uint64_t* nglobal)
{
    const AciResult& r = static_cast<OracleAci*>(h)->res;
    for (size_t i = 0; i < r.ranks.size(); ++i) {
        ranks[i] = r.ranks[i];
        errors[i] = r.errors[i];
        nglobal[i] = r.nglobal_pivots[i];
    }
}

// ElementwiseProblem stepping interface (state.rs) for white-box parity tests
namespace {
struct OracleAciProblem {
    std::unique_ptr<ElementwiseProblem> p;
    AciOptions o;
    AciOp op;
};
} // namespace
void* oracle_aci_problem_new(void* const* inputs, uint64_t n_inputs, int op_kind, oracle_aci_op_fn cb, void* user,
                             const oracle_aci_options* opts, void* initial_guess)
{
    OracleAciProblem* h = nullptr;
    const int rc = guarded([&] {
        std::vector<SimpleTensorTrain> in;
        for (uint64_t k = 0; k < n_inputs; ++k) in.push_back(static_cast<OracleTT*>(inputs[k])->tt);
        auto r = std::make_unique<OracleAciProblem>();
        r->o = convert_aci(opts, initial_guess);
        r->op = make_aci_op(op_kind, cb, user);
        r->p = std::make_unique<ElementwiseProblem>(std::move(in), r->o);
        h = r.release();
    });
    return rc == 0 ? h : nullptr;
}
void oracle_aci_problem_release(void* h) { delete static_cast<OracleAciProblem*>(h); }
int oracle_aci_problem_local_update(void* h, uint64_t bond, int left_orthogonal)
{
    return guarded([&] {
        auto* s = static_cast<OracleAciProblem*>(h);
        s->p->local_update(bond, left_orthogonal != 0, s->o, s->op);
    });
}
int oracle_aci_problem_add_global_pivots(void* h, const uint64_t* pivots /* n_sites x n col-major */, uint64_t n, uint64_t* added)
{
    return guarded([&] {
        auto* s = static_cast<OracleAciProblem*>(h);
        const size_t ns = s->p->len();
        std::vector<MultiIndex> pv(n, MultiIndex(ns));
        for (size_t p = 0; p < n; ++p)
            for (size_t q = 0; q < ns; ++q) pv[p][q] = pivots[q + ns * p];
        *added = s->p->add_global_pivots(pv);
    });
}
int oracle_aci_problem_find_global_pivots(void* h, uint64_t seed, uint64_t* count, uint64_t* out /* n_sites x max_nglobal_pivot */)
{
    return guarded([&] {
        auto* s = static_cast<OracleAciProblem*>(h);
        const std::vector<MultiIndex> pv = aci_find_global_pivots(*s->p, s->op, s->o, seed);
        *count = pv.size();
        const size_t ns = s->p->len();
        for (size_t p = 0; p < pv.size(); ++p)
            for (size_t q = 0; q < ns; ++q) out[q + ns * p] = pv[p][q];
    });
}
void* oracle_aci_problem_solution(void* h)
{
    auto* t = new OracleTT;
    t->tt = static_cast<OracleAciProblem*>(h)->p->solution;
    return t;
}
// frame shape (rows, cols) or (0, 0) when absent; data column-major
int oracle_aci_problem_frame(void* h, int right, uint64_t input, uint64_t site, uint64_t* rows, uint64_t* cols, double* out)
{
    return guarded([&] {
        auto* s = static_cast<OracleAciProblem*>(h);
        const auto& fr = right ? s->p->right_frames : s->p->left_frames;
        if (input >= fr.size() || site >= fr[input].size()) throw OracleError(ERR_INVALID_ARGUMENT, "frame index out of range");
        const AciFrame& f = fr[input][site];
        *rows = f.present ? f.m.nr : 0;
        *cols = f.present ? f.m.nc : 0;
        if (out && f.present) to_ptr(f.m, out);
    });
}
void oracle_aci_problem_errors(void* h, double* pivot_errors, double* pivot_scales)
{
    auto* s = static_cast<OracleAciProblem*>(h);
    for (size_t b = 0; b < s->p->pivot_errors.size(); ++b) {
        pivot_errors[b] = s->p->pivot_errors[b];
        pivot_scales[b] = s->p->pivot_scales[b];
    }
}

// ---- TreeACI local step (t4a_oracle_treeaci.hpp).  Capacities as for t4a_gpu_treeaci_local_update_f64; batch (may be null): what the
// operator saw, n_inputs x (row_count * col_count) ----
int oracle_treeaci_local_update(uint64_t n_inputs, const uint64_t* bond_dims, const double* const* row_frames, const double* const* col_frames,
                                uint64_t row_count, uint64_t col_count, int op_kind, oracle_aci_op_fn cb, void* user, uint64_t max_bond_dim,
                                double tolerance, int scale_tolerance, int left_orthogonal, uint64_t* rank, uint64_t* row_indices,
                                uint64_t* col_indices, double* pivot_errors, uint64_t* n_pivot_errors, double* left, double* right,
                                double* sampled_scale, double* local_values, double* batch)
{
    return guarded([&] {
        std::vector<size_t> bd(bond_dims, bond_dims + n_inputs);
        std::vector<const double*> rf(row_frames, row_frames + n_inputs), cf(col_frames, col_frames + n_inputs);
        const AciOp op = make_aci_op(op_kind, cb, user);
        TreeAciLocalUpdate u = treeaci_local_update(bd, rf, cf, row_count, col_count, op, max_bond_dim != 0, max_bond_dim, tolerance,
                                                    scale_tolerance != 0, left_orthogonal != 0);
        *rank = u.row_indices.size();
        for (size_t i = 0; i < u.row_indices.size(); ++i) {
            row_indices[i] = u.row_indices[i];
            col_indices[i] = u.col_indices[i];
        }
        *n_pivot_errors = u.pivot_errors.size();
        for (size_t i = 0; i < u.pivot_errors.size(); ++i) pivot_errors[i] = u.pivot_errors[i];
        std::memcpy(left, u.left.a.data(), sizeof(double) * u.left.a.size());
        std::memcpy(right, u.right.a.data(), sizeof(double) * u.right.a.size());
        *sampled_scale = u.sampled_scale;
        if (local_values) std::memcpy(local_values, u.local_values.data(), sizeof(double) * u.local_values.size());
        if (batch) std::memcpy(batch, u.batch.data(), sizeof(double) * u.batch.size());
    });
}


// ---- floating_zone / estimate_true_error / opt_first_pivot (t4a_oracle_search.hpp) ----
typedef double (*oracle_scalar_fn)(void* ctx, const uint64_t* idx, uint64_t n);
static ScalarFn wrap_scalar(oracle_scalar_fn f, void* ctx)
{
    return [f, ctx](const MultiIndex& idx) {
        std::vector<uint64_t> u(idx.begin(), idx.end());
        return f(ctx, u.data(), (uint64_t)u.size());
    };
}
int oracle_tt_floating_zone(void* h, oracle_scalar_fn f, void* ctx, const uint64_t* local_dims, uint64_t n_sites, const uint64_t* init_p,
                            uint64_t seed, double early_stop_tol, uint64_t* pivot_out, double* error_out)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        std::vector<size_t> ld(local_dims, local_dims + n_sites);
        MultiIndex init;
        if (init_p) init.assign(init_p, init_p + n_sites);
        auto r = floating_zone(tt, wrap_scalar(f, ctx), ld, init_p ? &init : nullptr, seed, early_stop_tol);
        for (size_t s = 0; s < r.first.size(); ++s) pivot_out[s] = r.first[s];
        *error_out = r.second;
    });
}
int oracle_tt_estimate_true_error(void* h, oracle_scalar_fn f, void* ctx, uint64_t nsearch, const uint64_t* initial_points, uint64_t n_initial,
                                  uint64_t seed, uint64_t* pivots_out, double* errors_out, uint64_t capacity, uint64_t* n_out)
{
    return guarded([&] {
        auto& tt = static_cast<OracleTT*>(h)->tt;
        const size_t n = tt.len();
        std::vector<MultiIndex> init;
        if (initial_points)
            for (size_t k = 0; k < n_initial; ++k) init.emplace_back(initial_points + k * n, initial_points + (k + 1) * n);
        auto res = estimate_true_error(tt, wrap_scalar(f, ctx), (size_t)nsearch, initial_points ? &init : nullptr, seed);
        *n_out = res.size();
        if (res.size() > capacity) throw OracleError(ERR_INVALID_ARGUMENT, "estimate_true_error: capacity too small");
        for (size_t k = 0; k < res.size(); ++k) {
            for (size_t s = 0; s < n; ++s) pivots_out[s + n * k] = res[k].first[s];
            errors_out[k] = res[k].second;
        }
    });
}
int oracle_opt_first_pivot(oracle_scalar_fn f, void* ctx, const uint64_t* local_dims, uint64_t n_sites, const uint64_t* first_pivot,
                           uint64_t max_sweep, uint64_t* pivot_out)
{
    return guarded([&] {
        std::vector<size_t> ld(local_dims, local_dims + n_sites);
        MultiIndex fp(first_pivot, first_pivot + n_sites);
        MultiIndex r = opt_first_pivot(wrap_scalar(f, ctx), ld, fp, (size_t)max_sweep);
        for (size_t s = 0; s < r.size(); ++s) pivot_out[s] = r[s];
    });
}

} // extern "C"
