// aci.hip — see aci.hpp.  Reference: crates/tensor4all-aci/src/{elementwise,state,local,global_guard,random_tt,validation}.rs.
#include "aci.hpp"
#include "stdrng.hpp"
#include "smallrng.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

#include "kernels.hpp"

namespace t4a {

namespace {

// C (M x N, ldc) = A (M x K, lda) * B (K x N, ldb); one thread per entry, k ascending, multiply and add rounded separately
// (-ffp-contract=off): the order of the oracle's mat_mul, which stands in for the reference's tenferro matmul.
__global__ void __launch_bounds__(256) aci_matmul_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                         double* __restrict__ C, int ldc, int M, int N, int K)
{
    const size_t total = (size_t)M * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (size_t)M), j = (int)(e / (size_t)M);
        const double* a = A + i;
        const double* b = B + (size_t)ldb * j;
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc = acc + a[(size_t)lda * k] * b[k];
        C[i + (size_t)ldc * j] = acc;
    }
}

struct AciPiArgs {
    const double* lf[ACI_MAX_INPUTS]; // (nrows x mid_k), column-major
    const double* rf[ACI_MAX_INPUTS]; // (mid_k x ncols)
    int mid[ACI_MAX_INPUTS];
    int n_inputs, nrows, ncols, op;
    double* pi;   // nrows x ncols (built-in operators)
    double* vals; // n_inputs x (nrows * ncols) (callback operator)
};

// LocalBlockEvaluator::materialize_local_matrix (local.rs:299-394) with the operator fused in: one thread per entry of the
// candidate matrix evaluates every input's two-site value (left factor row x right factor column) and combines them
__global__ void __launch_bounds__(256) aci_pi_kernel(AciPiArgs p)
{
    const size_t total = (size_t)p.nrows * p.ncols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e % (size_t)p.nrows), col = (int)(e / (size_t)p.nrows);
        double out = 0.0;
#pragma unroll
        for (int k = 0; k < ACI_MAX_INPUTS; ++k) {
            if (k < p.n_inputs) {
                const double* a = p.lf[k] + row;
                const double* b = p.rf[k] + (size_t)p.mid[k] * col;
                double acc = 0.0;
                for (int m = 0; m < p.mid[k]; ++m) acc = acc + a[(size_t)p.nrows * m] * b[m];
                if (p.op == 0) p.vals[k + (size_t)p.n_inputs * e] = acc;
                else if (k == 0) out = acc;
                else out = (p.op == 1) ? out * acc : out + acc;
            }
        }
        if (p.op != 0) p.pi[e] = out;
    }
}

void matmul(const double* A, int lda, const double* B, int ldb, double* C, int ldc, size_t M, size_t N, size_t K, hipStream_t st)
{
    const size_t total = M * N;
    if (total == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(aci_matmul_kernel, dim3(blocks), dim3(256), 0, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K);
}

constexpr size_t MAX_GUESS_ENTRIES = 10000000; // random_tt.rs:12-13

void validate_inputs(const std::vector<TensorTrain*>& in) // validation.rs:48-119
{
    if (in.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "inputs must not be empty");
    if (in.size() > (size_t)ACI_MAX_INPUTS) throw Error(T4A_GPU_NOT_IMPLEMENTED, "more than 8 input tensor trains are not supported");
    for (TensorTrain* t : in)
        if (!t) throw Error(T4A_GPU_NULL_POINTER, "input tensor train is null");
    const size_t n = in[0]->len();
    if (n == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "input tensor trains must have at least one site");
    for (size_t k = 0; k < in.size(); ++k) {
        if (in[k]->len() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "input tensor trains must have the same length");
        for (size_t s = 0; s < n; ++s) {
            const DevCore& c = in[k]->cores[s];
            if (c.l == 0 || c.s == 0 || c.r == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "input core dimensions must be positive");
            if (c.s != in[0]->cores[s].s) throw Error(T4A_GPU_INVALID_ARGUMENT, "site dimension mismatch between inputs");
        }
    }
}

void host_builtin(AciOpKind kind, const double* v, size_t K, size_t np, double* out)
{
    for (size_t p = 0; p < np; ++p) {
        double acc = v[K * p];
        for (size_t k = 1; k < K; ++k) acc = kind == AciOpKind::Product ? acc * v[k + K * p] : acc + v[k + K * p];
        out[p] = acc;
    }
}

} // namespace

void seq_matmul_launch(const double* A, int lda, const double* B, int ldb, double* C, int ldc, size_t M, size_t N, size_t K,
                       hipStream_t stream)
{
    matmul(A, lda, B, ldb, C, ldc, M, N, K, stream);
}

void AciOptions::validate() const
{
    if (max_iters == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_iters must be at least 1");
    if (min_iters == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "min_iters must be at least 1");
    if (has_max_bond_dim && max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be at least 1");
    if (min_iters > max_iters) throw Error(T4A_GPU_INVALID_ARGUMENT, "min_iters must be less than or equal to max_iters");
    if (!std::isfinite(tolerance) || tolerance < 0.0) throw Error(T4A_GPU_INVALID_ARGUMENT, "tolerance must be finite and non-negative");
    if (!std::isfinite(tol_margin_global_search) || tol_margin_global_search < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "tol_margin_global_search must be finite and non-negative");
}

void AciProblem::apply_op_host(const double* values, size_t n_points, double* out)
{
    if (kind_ == AciOpKind::Callback) host_op_(values, n_inputs(), n_points, out);
    else host_builtin(kind_, values, n_inputs(), n_points, out);
}

void AciProblem::set_frame(AciFrame& f, const std::vector<double>& host, size_t nr, size_t nc)
{
    f.buf.reserve(std::max<size_t>(nr * nc, 1));
    if (nr * nc) T4A_HIP(hipMemcpyAsync(f.buf.get(), host.data(), nr * nc * sizeof(double), hipMemcpyHostToDevice, eng_.stream()));
    eng_.sync(); // `host` may be a temporary
    f.nr = nr;
    f.nc = nc;
    f.present = true;
}

AciProblem::AciProblem(const std::vector<TensorTrain*>& inputs, const TensorTrain* guess, const AciOptions& options, AciOpKind kind,
                       AciHostOp host_op)
    : inputs_(inputs), opt_(options), kind_(kind), host_op_(std::move(host_op))
{
    opt_.validate();
    validate_inputs(inputs_);
    if (kind_ == AciOpKind::Callback && !host_op_) throw Error(T4A_GPU_NULL_POINTER, "operator callback is null");
    for (TensorTrain* t : inputs_) t->eng.sync();
    const size_t n = inputs_[0]->len(), K = inputs_.size();
    hipStream_t st = eng_.stream();
    sol_.resize(n);
    if (guess) { // random_tt.rs:41-83
        if (guess->len() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess site dimensions must match inputs");
        size_t total = 0;
        for (size_t s = 0; s < n; ++s) {
            const DevCore& c = guess->cores[s];
            if (c.s != inputs_[0]->cores[s].s) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess site dimensions must match inputs");
            if (c.l == 0 || c.s == 0 || c.r == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess core dimensions must be positive");
            if (s + 1 < n && opt_.has_max_bond_dim && c.r > opt_.max_bond_dim)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess bond dimension exceeds max_bond_dim");
            total += c.size();
        }
        if (total > MAX_GUESS_ENTRIES) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess total size exceeds internal limit");
        const_cast<TensorTrain*>(guess)->eng.sync();
        for (size_t s = 0; s < n; ++s) {
            const DevCore& c = guess->cores[s];
            sol_[s].l = c.l;
            sol_[s].s = c.s;
            sol_[s].r = c.r;
            sol_[s].buf.reserve(c.size());
            T4A_HIP(hipMemcpyAsync(sol_[s].buf.get(), c.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
    } else { // :15-39, default_link_dims :97-137
        std::vector<size_t> link(n > 0 ? n - 1 : 0);
        auto sat_mul = [](size_t a, size_t b) {
            if (a != 0 && b > std::numeric_limits<size_t>::max() / a) throw Error(T4A_GPU_INVALID_ARGUMENT, "site dimension product overflows usize");
            return a * b;
        };
        std::vector<size_t> lp(link.size()), rp(link.size(), 1);
        size_t acc = 1;
        for (size_t b = 0; b + 1 < n; ++b) lp[b] = acc = sat_mul(acc, inputs_[0]->cores[b].s);
        acc = 1;
        for (size_t b = n - 1; b-- > 0;) rp[b] = acc = sat_mul(acc, inputs_[0]->cores[b + 1].s);
        for (size_t b = 0; b + 1 < n; ++b) {
            size_t m = std::numeric_limits<size_t>::max();
            for (TensorTrain* t : inputs_) m = std::min(m, t->cores[b].r);
            size_t d = std::min(lp[b], rp[b]);
            if (opt_.has_max_bond_dim) d = std::min(d, opt_.max_bond_dim);
            link[b] = std::max<size_t>(std::min(d, m), 1);
        }
        ChaCha8Rng rng(opt_.rng_seed); // random_tt.rs:31 (smallrng.hpp)
        size_t total = 0;
        std::vector<double> host;
        for (size_t s = 0; s < n; ++s) {
            DevCore& c = sol_[s];
            c.l = s == 0 ? 1 : link[s - 1];
            c.s = inputs_[0]->cores[s].s;
            c.r = s < link.size() ? link[s] : 1;
            total += c.size();
            if (total > MAX_GUESS_ENTRIES) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial guess total size exceeds internal limit");
            host.resize(c.size());
            for (double& v : host) v = StandardNormal::sample(rng); // scalar.rs:8-20
            c.buf.reserve(c.size());
            T4A_HIP(hipMemcpyAsync(c.buf.get(), host.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, st));
            eng_.sync();
        }
    }
    eng_.sync();
    lframes_.resize(K);
    rframes_.resize(K);
    lf_.resize(K);
    rf_.resize(K);
    for (size_t k = 0; k < K; ++k) {
        lframes_[k].resize(n + 1);
        rframes_[k].resize(n + 1);
        set_frame(lframes_[k][0], {1.0}, 1, 1);
        set_frame(rframes_[k][n], {1.0}, 1, 1);
    }
    pivot_errors.assign(n - 1, 0.0);
    pivot_scales.assign(n - 1, 0.0);
    initialize_right_frames();
}

size_t AciProblem::rank() const
{
    size_t r = 1;
    for (size_t s = 0; s + 1 < sol_.size(); ++s) r = std::max(r, sol_[s].r);
    return r;
}

std::vector<size_t> AciProblem::link_dims() const
{
    std::vector<size_t> v;
    for (size_t s = 0; s + 1 < sol_.size(); ++s) v.push_back(sol_[s].r);
    return v;
}

void AciProblem::left_factor(size_t k, size_t site, DevBuf<double>& out) // local.rs:692-702
{
    const AciFrame& f = lframes_[k][site];
    const DevCore& c = inputs_[k]->cores[site];
    if (!f.present) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing left frame");
    if (f.nc != c.l) throw Error(T4A_GPU_INVALID_ARGUMENT, "left frame/input bond mismatch");
    out.reserve(std::max<size_t>(f.nr * c.s * c.r, 1));
    matmul(f.buf.get(), (int)f.nr, c.buf.get(), (int)c.l, out.get(), (int)f.nr, f.nr, c.s * c.r, c.l, eng_.stream());
}

void AciProblem::right_factor(size_t k, size_t site, DevBuf<double>& out) // local.rs:704-714
{
    const AciFrame& f = rframes_[k][site + 1];
    const DevCore& c = inputs_[k]->cores[site];
    if (!f.present) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing right frame");
    if (c.r != f.nr) throw Error(T4A_GPU_INVALID_ARGUMENT, "right frame/input bond mismatch");
    out.reserve(std::max<size_t>(c.l * c.s * f.nc, 1));
    matmul(c.buf.get(), (int)(c.l * c.s), f.buf.get(), (int)f.nr, out.get(), (int)(c.l * c.s), c.l * c.s, f.nc, c.r, eng_.stream());
}

void AciProblem::select_left_frames(size_t site, const std::vector<int>& rows) // state.rs:215-255 (lf_ = frame x core of `site`)
{
    hipStream_t st = eng_.stream();
    d_idx_.reserve(rows.size());
    T4A_HIP(hipMemcpyAsync(d_idx_.get(), rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, st));
    for (size_t k = 0; k < n_inputs(); ++k) {
        const DevCore& c = inputs_[k]->cores[site];
        const size_t full_rows = lframes_[k][site].nr * c.s;
        for (int r : rows)
            if (r < 0 || (size_t)r >= full_rows) throw Error(T4A_GPU_INVALID_ARGUMENT, "row index out of bounds for the left frame");
        AciFrame& f = lframes_[k][site + 1];
        f.buf.reserve(std::max<size_t>(rows.size() * c.r, 1));
        gather_launch(lf_[k].get(), (int)full_rows, d_idx_.get(), (int)rows.size(), nullptr, (int)c.r, f.buf.get(), (int)rows.size(), st);
        f.nr = rows.size();
        f.nc = c.r;
        f.present = true;
    }
    T4A_HIP(hipGetLastError());
    eng_.sync(); // `rows` is read asynchronously
}

void AciProblem::select_right_frames(size_t site, const std::vector<int>& cols) // state.rs:257-299 (rf_ = core of `site` x frame)
{
    hipStream_t st = eng_.stream();
    d_idx_.reserve(cols.size());
    T4A_HIP(hipMemcpyAsync(d_idx_.get(), cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice, st));
    for (size_t k = 0; k < n_inputs(); ++k) {
        const DevCore& c = inputs_[k]->cores[site];
        const size_t full_cols = c.s * rframes_[k][site + 1].nc;
        for (int q : cols)
            if (q < 0 || (size_t)q >= full_cols) throw Error(T4A_GPU_INVALID_ARGUMENT, "column index out of bounds for the right frame");
        AciFrame& f = rframes_[k][site];
        f.buf.reserve(std::max<size_t>(c.l * cols.size(), 1));
        gather_launch(rf_[k].get(), (int)c.l, nullptr, (int)c.l, d_idx_.get(), (int)cols.size(), f.buf.get(), (int)c.l, st);
        f.nr = c.l;
        f.nc = cols.size();
        f.present = true;
    }
    T4A_HIP(hipGetLastError());
    eng_.sync();
}

void AciProblem::initialize_right_frames() // state.rs:862-925
{
    const size_t n = len();
    hipStream_t st = eng_.stream();
    for (size_t site = n; site-- > 1;) {
        DevCore& cur = sol_[site];
        const size_t nrows = cur.l, ncols = cur.s * cur.r;
        RrLUOptions lo;
        lo.max_bond_dim = std::numeric_limits<size_t>::max();
        lo.rel_tol = 0.0;
        lo.abs_tol = 0.0;
        lo.left_orthogonal = false;
        LuciResult r = eng_.luci(cur.buf.get(), (int)nrows, (int)ncols, lo, true, false);
        size_t new_rank = (size_t)r.rank;
        std::vector<int> cols(r.col_perm.begin(), r.col_perm.begin() + r.rank);
        if (r.rank == 0) {
            new_rank = 1;
            cols = {0};
            eng_.reserve_factors(nrows, ncols);
            fill_launch(eng_.left(), nrows, 0.0, st);
            fill_launch(eng_.right(), ncols, 0.0, st);
        }
        // previous core x left factor first (it reads the engine's factor buffer), then the new right core
        DevCore& prev = sol_[site - 1];
        DevBuf<double> np;
        np.reserve(prev.l * prev.s * new_rank);
        matmul(prev.buf.get(), (int)(prev.l * prev.s), eng_.left(), (int)nrows, np.get(), (int)(prev.l * prev.s), prev.l * prev.s, new_rank,
               prev.r, st);
        DevBuf<double> nc;
        nc.reserve(new_rank * ncols);
        T4A_HIP(hipMemcpyAsync(nc.get(), eng_.right(), new_rank * ncols * sizeof(double), hipMemcpyDeviceToDevice, st));
        eng_.sync();
        cur.buf = std::move(nc);
        cur.l = new_rank;
        prev.buf = std::move(np);
        prev.r = new_rank;
        for (size_t k = 0; k < n_inputs(); ++k) right_factor(k, site, rf_[k]);
        select_right_frames(site, cols);
    }
}

void AciProblem::local_update(size_t bond, bool left_orthogonal) // state.rs:729-860
{
    const size_t n = len(), K = n_inputs();
    if (n < 2 || bond >= n - 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "bond index out of bounds for the tensor train");
    DevCore& lc = sol_[bond];
    DevCore& rc = sol_[bond + 1];
    if (lc.r != rc.l) throw Error(T4A_GPU_INVALID_ARGUMENT, "adjacent solution core bond mismatch");
    const size_t lrank = lc.l, s1 = lc.s, s2 = rc.s, rrank = rc.r;
    const size_t nrows = lrank * s1, ncols = s2 * rrank, np = nrows * ncols;
    if (nrows > 65535 || ncols > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "local blocks above 65535 rows or columns are not supported");
    hipStream_t st = eng_.stream();
    AciPiArgs a{};
    for (size_t k = 0; k < K; ++k) {
        const AciFrame& fl = lframes_[k][bond];
        const AciFrame& fr = rframes_[k][bond + 2];
        const DevCore& ca = inputs_[k]->cores[bond];
        const DevCore& cb = inputs_[k]->cores[bond + 1];
        if (!fl.present || !fr.present) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing frame for the local update");
        if (fl.nr * ca.s != nrows || cb.s * fr.nc != ncols)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "local block shape mismatch between the solution and the frames");
        left_factor(k, bond, lf_[k]);
        right_factor(k, bond + 1, rf_[k]);
        a.lf[k] = lf_[k].get();
        a.rf[k] = rf_[k].get();
        a.mid[k] = (int)ca.r;
    }
    a.n_inputs = (int)K;
    a.nrows = (int)nrows;
    a.ncols = (int)ncols;
    a.op = (int)kind_;
    double* d_pi = eng_.pi(np);
    a.pi = d_pi;
    if (kind_ == AciOpKind::Callback) {
        d_vals_.reserve(K * np);
        a.vals = d_vals_.get();
    }
    const unsigned blocks = (unsigned)std::min<size_t>((np + 255) / 256, 8192);
    hipLaunchKernelGGL(aci_pi_kernel, dim3(blocks), dim3(256), 0, st, a);
    T4A_HIP(hipGetLastError());
    if (kind_ == AciOpKind::Callback) {
        std::vector<double> hv(K * np), ho(np);
        T4A_HIP(hipMemcpyAsync(hv.data(), d_vals_.get(), hv.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        eng_.sync();
        host_op_(hv.data(), K, np, ho.data());
        T4A_HIP(hipMemcpyAsync(d_pi, ho.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
        eng_.sync();
    }
    RrLUOptions lo;
    lo.max_bond_dim = opt_.has_max_bond_dim ? opt_.max_bond_dim : std::numeric_limits<size_t>::max();
    lo.rel_tol = opt_.scale_tolerance ? opt_.tolerance : 0.0;
    lo.abs_tol = opt_.scale_tolerance ? 0.0 : opt_.tolerance;
    lo.left_orthogonal = left_orthogonal;
    LuciResult r = eng_.luci(d_pi, (int)nrows, (int)ncols, lo, true, false);
    const double pivot_error = r.pivot_errors.empty() ? 0.0 : r.pivot_errors.back();
    size_t new_rank = (size_t)r.rank;
    std::vector<int> rows(r.row_perm.begin(), r.row_perm.begin() + r.rank), cols(r.col_perm.begin(), r.col_perm.begin() + r.rank);
    if (r.rank == 0) { // zero block: rank-one zero update keeps every frame non-empty (:806-822)
        new_rank = 1;
        rows = {0};
        cols = {0};
        eng_.reserve_factors(nrows, ncols);
        fill_launch(eng_.left(), nrows, 0.0, st);
        fill_launch(eng_.right(), ncols, 0.0, st);
    }
    DevBuf<double> nl, nr;
    nl.reserve(nrows * new_rank);
    nr.reserve(new_rank * ncols);
    T4A_HIP(hipMemcpyAsync(nl.get(), eng_.left(), nrows * new_rank * sizeof(double), hipMemcpyDeviceToDevice, st));
    T4A_HIP(hipMemcpyAsync(nr.get(), eng_.right(), new_rank * ncols * sizeof(double), hipMemcpyDeviceToDevice, st));
    eng_.sync();
    lc.buf = std::move(nl);
    lc.r = new_rank;
    rc.buf = std::move(nr);
    rc.l = new_rank;
    if (left_orthogonal) select_left_frames(bond, rows);
    else select_right_frames(bond + 1, cols);
    pivot_errors[bond] = pivot_error;
    pivot_scales[bond] = r.abs_max;
}

std::vector<double> AciProblem::frame_host(bool right, size_t input, size_t site, size_t* nr, size_t* nc)
{
    if (input >= n_inputs() || site > len()) throw Error(T4A_GPU_INVALID_ARGUMENT, "frame index out of range");
    const AciFrame& f = right ? rframes_[input][site] : lframes_[input][site];
    *nr = f.present ? f.nr : 0;
    *nc = f.present ? f.nc : 0;
    std::vector<double> h(f.present ? f.nr * f.nc : 0);
    if (!h.empty()) {
        T4A_HIP(hipMemcpyAsync(h.data(), f.buf.get(), h.size() * sizeof(double), hipMemcpyDeviceToHost, eng_.stream()));
        eng_.sync();
    }
    return h;
}

std::unique_ptr<TensorTrain> AciProblem::solution_tt()
{
    eng_.sync();
    return std::make_unique<TensorTrain>(sol_, eng_.stream());
}

size_t AciProblem::add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots) // state.rs:551-652
{
    const size_t n = len(), K = n_inputs();
    hipStream_t st = eng_.stream();
    std::vector<size_t> growth(n + 1, 0), dims(n + 1, 0), lsp(n + 1, 1), rsp(n + 1, 1), bounds(n + 1);
    for (size_t s = 0; s < n; ++s) lsp[s + 1] = lsp[s] * sol_[s].s; // algebraic_bond_bounds :654-681
    for (size_t s = n; s-- > 0;) rsp[s] = rsp[s + 1] * sol_[s].s;
    for (size_t b = 0; b <= n; ++b) bounds[b] = std::min(lsp[b], rsp[b]);
    for (size_t b = 0; b + 1 < n; ++b) dims[b + 1] = sol_[b].r;
    // host mirrors of the frames an injection can touch
    struct HostFrame {
        std::vector<double> a;
        size_t nr = 0, nc = 0;
        bool present = false, dirty = false;
    };
    std::vector<std::vector<HostFrame>> hl(K, std::vector<HostFrame>(n + 1)), hr(K, std::vector<HostFrame>(n + 1));
    for (size_t k = 0; k < K; ++k)
        for (size_t b = 1; b < n; ++b) {
            hl[k][b].a = frame_host(false, k, b, &hl[k][b].nr, &hl[k][b].nc);
            hl[k][b].present = lframes_[k][b].present;
            hr[k][b].a = frame_host(true, k, b, &hr[k][b].nr, &hr[k][b].nc);
            hr[k][b].present = rframes_[k][b].present;
        }
    size_t new_pivots = 0;
    for (const auto& pivot : pivots) {
        if (pivot.size() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot length must match the number of sites");
        for (size_t s = 0; s < n; ++s)
            if (pivot[s] >= sol_[s].s) throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot index out of bounds");
        // left / right environments of every cut (left_environment / right_environment, :1286-1319) as device chains of
        // vector x core-slice products; env offsets per input
        std::vector<std::vector<size_t>> loff(K, std::vector<size_t>(n + 1)), roff(K, std::vector<size_t>(n + 1));
        size_t total = 0;
        for (size_t k = 0; k < K; ++k) {
            for (size_t b = 0; b <= n; ++b) {
                loff[k][b] = total;
                total += b == 0 ? 1 : inputs_[k]->cores[b - 1].r;
            }
            for (size_t b = 0; b <= n; ++b) {
                roff[k][b] = total;
                total += b == n ? 1 : inputs_[k]->cores[b].l;
            }
        }
        d_env_.reserve(total);
        for (size_t k = 0; k < K; ++k) {
            fill_launch(d_env_.get() + loff[k][0], 1, 1.0, st);
            fill_launch(d_env_.get() + roff[k][n], 1, 1.0, st);
            for (size_t b = 1; b <= n; ++b) { // env_b (1 x r) = env_{b-1} (1 x l) * core[b-1][:, idx, :]
                const DevCore& c = inputs_[k]->cores[b - 1];
                matmul(d_env_.get() + loff[k][b - 1], 1, c.buf.get() + c.l * pivot[b - 1], (int)(c.l * c.s), d_env_.get() + loff[k][b], 1, 1, c.r,
                       c.l, st);
            }
            for (size_t b = n; b-- > 0;) { // env_b (l x 1) = core[b][:, idx, :] (l x r) * env_{b+1} (r x 1)
                const DevCore& c = inputs_[k]->cores[b];
                matmul(c.buf.get() + c.l * pivot[b], (int)(c.l * c.s), d_env_.get() + roff[k][b + 1], (int)c.r, d_env_.get() + roff[k][b],
                       (int)c.l, c.l, 1, c.r, st);
            }
        }
        T4A_HIP(hipGetLastError());
        std::vector<double> env(total);
        T4A_HIP(hipMemcpyAsync(env.data(), d_env_.get(), total * sizeof(double), hipMemcpyDeviceToHost, st));
        eng_.sync();
        bool injected = false;
        for (size_t bond = 1; bond < n; ++bond) {
            if (dims[bond] >= bounds[bond]) continue;
            const bool needs_row = bond + 1 < n, needs_col = bond >= 2;
            bool duplicate = true;
            for (size_t k = 0; k < K; ++k) {
                if (needs_row) { // frame_has_row :1321-1341
                    const HostFrame& f = hl[k][bond];
                    const double* row = env.data() + loff[k][bond];
                    bool has = false;
                    if (f.present) {
                        if (f.nc != inputs_[k]->cores[bond - 1].r) throw Error(T4A_GPU_INVALID_ARGUMENT, "cannot match a row against a frame of another width");
                        for (size_t r = 0; r < f.nr && !has; ++r) {
                            bool all = true;
                            for (size_t c = 0; c < f.nc; ++c) all = all && (f.a[r + f.nr * c] == row[c]);
                            has = all;
                        }
                    }
                    if (!has) duplicate = false;
                }
                if (needs_col) { // frame_has_col :1361-1381
                    const HostFrame& f = hr[k][bond];
                    const double* col = env.data() + roff[k][bond];
                    bool has = false;
                    if (f.present) {
                        if (f.nr != inputs_[k]->cores[bond].l) throw Error(T4A_GPU_INVALID_ARGUMENT, "cannot match a column against a frame of another height");
                        for (size_t c = 0; c < f.nc && !has; ++c) {
                            bool all = true;
                            for (size_t r = 0; r < f.nr; ++r) all = all && (f.a[r + f.nr * c] == col[r]);
                            has = all;
                        }
                    }
                    if (!has) duplicate = false;
                }
            }
            if (duplicate) continue;
            for (size_t k = 0; k < K; ++k) {
                if (needs_row) { // append_row :1343-1359: the column-major buffer is extended and re-read with one more row
                    HostFrame& f = hl[k][bond];
                    if (!f.present) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing left frame for a global pivot");
                    const double* row = env.data() + loff[k][bond];
                    f.a.insert(f.a.end(), row, row + f.nc);
                    f.nr += 1;
                    f.dirty = true;
                }
                if (needs_col) { // append_col :1383-1401
                    HostFrame& f = hr[k][bond];
                    if (!f.present) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing right frame for a global pivot");
                    const double* col = env.data() + roff[k][bond];
                    f.a.insert(f.a.end(), col, col + f.nr);
                    f.nc += 1;
                    f.dirty = true;
                }
            }
            growth[bond] += 1;
            dims[bond] += 1;
            injected = true;
        }
        if (injected) ++new_pivots;
    }
    for (size_t k = 0; k < K; ++k)
        for (size_t b = 1; b < n; ++b) {
            if (hl[k][b].dirty) set_frame(lframes_[k][b], hl[k][b].a, hl[k][b].nr, hl[k][b].nc);
            if (hr[k][b].dirty) set_frame(rframes_[k][b], hr[k][b].a, hr[k][b].nr, hr[k][b].nc);
        }
    if (new_pivots > 0) { // pad_solution_internal_bonds :683-727
        for (size_t s = 0; s < n; ++s) {
            DevCore& c = sol_[s];
            const size_t nl = s == 0 ? c.l : c.l + growth[s], nr = s == n - 1 ? c.r : c.r + growth[s + 1];
            if (nl == c.l && nr == c.r) continue;
            DevBuf<double> p;
            p.reserve(nl * c.s * nr);
            fill_launch(p.get(), nl * c.s * nr, 0.0, st);
            gather_launch(c.buf.get(), (int)c.l, nullptr, (int)c.l, nullptr, (int)(c.s * c.r), p.get(), (int)nl, st);
            T4A_HIP(hipGetLastError());
            eng_.sync();
            c.buf = std::move(p);
            c.l = nl;
            c.r = nr;
        }
    }
    return new_pivots;
}

std::vector<std::vector<uint32_t>> AciProblem::find_global_pivots(uint64_t seed) // global_guard.rs:49-181
{
    const size_t n = len(), K = n_inputs(), nsearch = opt_.nsearch_global_pivots;
    if (nsearch == 0 || opt_.max_nglobal_pivot == 0 || n < 2) return {};
    std::vector<size_t> site_dims(n);
    for (size_t s = 0; s < n; ++s) site_dims[s] = sol_[s].s;
    StdRng rng(seed); // global_guard.rs:71 (stdrng.hpp)
    std::vector<std::vector<uint32_t>> starts(nsearch, std::vector<uint32_t>(n));
    for (auto& sp : starts)
        for (size_t q = 0; q < n; ++q) sp[q] = (uint32_t)rng.random_range(site_dims[q]);
    auto flat = [&](const std::vector<std::vector<uint32_t>>& pts) {
        std::vector<uint32_t> f(pts.size() * n);
        for (size_t p = 0; p < pts.size(); ++p) std::copy(pts[p].begin(), pts[p].end(), f.begin() + p * n);
        return f;
    };
    // operator values at `pts`: every input evaluated through its cache-style split evaluation (TTCache::evaluate_many)
    auto op_values = [&](const std::vector<std::vector<uint32_t>>& pts, size_t split) {
        const size_t np = pts.size();
        const std::vector<uint32_t> f = flat(pts);
        std::vector<double> iv(K * np), tmp(np), ov(np);
        for (size_t k = 0; k < K; ++k) {
            inputs_[k]->evaluate_many(f.data(), np, split, tmp.data());
            for (size_t p = 0; p < np; ++p) iv[k + K * p] = tmp[p];
        }
        apply_op_host(iv.data(), np, ov.data());
        return ov;
    };
    const std::vector<double> so = op_values(starts, 0);
    double max_op = 0.0;
    for (double v : so) max_op = std::fmax(max_op, std::sqrt(v * v));
    const double abs_tol = (opt_.scale_tolerance && max_op > 0.0) ? opt_.tolerance * max_op : opt_.tolerance;
    const double threshold = abs_tol * opt_.tol_margin_global_search;
    std::unique_ptr<TensorTrain> sol = solution_tt();
    auto errors_at = [&](const std::vector<std::vector<uint32_t>>& pts) {
        const size_t np = pts.size();
        size_t split = 0; // :100-110: first site where the batch differs, + 1
        if (np >= 2)
            for (size_t site = 0; site < n && split == 0; ++site)
                for (size_t p = 1; p < np; ++p)
                    if (pts[p][site] != pts[0][site]) {
                        split = site + 1;
                        break;
                    }
        const std::vector<double> ov = op_values(pts, split);
        const std::vector<uint32_t> f = flat(pts);
        std::vector<double> sv(np), errs(np);
        sol->evaluate_many(f.data(), np, split, sv.data());
        for (size_t p = 0; p < np; ++p) {
            const double d = ov[p] - sv[p];
            errs[p] = std::sqrt(d * d);
        }
        return errs;
    };
    std::vector<std::pair<double, std::vector<uint32_t>>> best;
    for (const auto& start : starts) { // floating_zone_walk (tensor4all-core/src/floating_zone.rs:46-103)
        std::vector<uint32_t> pivot = start;
        const std::vector<double> e0 = errors_at({pivot});
        double max_error = e0.empty() ? 0.0 : e0[0];
        for (size_t sw = 0; sw < opt_.nsweeps_global_search; ++sw) {
            const double prev = max_error;
            for (size_t ipos = 0; ipos < n; ++ipos) {
                std::vector<std::vector<uint32_t>> pts(site_dims[ipos], pivot);
                for (size_t v = 0; v < site_dims[ipos]; ++v) pts[v][ipos] = (uint32_t)v;
                const std::vector<double> errs = errors_at(pts);
                uint32_t best_idx = pivot[ipos];
                double be = 0.0;
                for (size_t v = 0; v < errs.size(); ++v)
                    if (errs[v] > be) {
                        be = errs[v];
                        best_idx = (uint32_t)v;
                    }
                pivot[ipos] = best_idx;
                max_error = std::fmax(max_error, be);
            }
            if (max_error == prev || max_error > threshold) break;
        }
        if (max_error > threshold) best.push_back({max_error, pivot});
    }
    std::stable_sort(best.begin(), best.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    std::vector<std::vector<uint32_t>> out;
    for (const auto& b : best)
        if (std::find(out.begin(), out.end(), b.second) == out.end()) {
            out.push_back(b.second);
            if (out.size() >= opt_.max_nglobal_pivot) break;
        }
    return out;
}

void AciProblem::run() // elementwise.rs:126-210
{
    const size_t n = len();
    size_t guard_runs = 0;
    ranks.clear();
    errors.clear();
    nglobal_pivots.clear();
    termination = AciTermination::MaxIterations;
    for (size_t it = 0; it < opt_.max_iters; ++it) {
        if (it % 2 == 0)
            for (size_t b = 0; b + 1 < n; ++b) local_update(b, true);
        else
            for (size_t b = n - 1; b-- > 0;) local_update(b, false);
        double metric = 0.0; // max_error_metric :467-480
        for (size_t b = 0; b < pivot_errors.size(); ++b) {
            const double e = (opt_.scale_tolerance && pivot_scales[b] > 0.0) ? pivot_errors[b] / pivot_scales[b] : pivot_errors[b];
            metric = std::fmax(metric, e);
        }
        ranks.push_back(rank());
        errors.push_back(metric);
        const bool capped = opt_.has_max_bond_dim && rank() >= opt_.max_bond_dim;
        if (opt_.enable_global_guard && opt_.nsearch_global_pivots > 0 && opt_.max_nglobal_pivot > 0 && !capped) {
            ++guard_runs;
            const auto pv = find_global_pivots(opt_.rng_seed + (uint64_t)guard_runs);
            add_global_pivots(pv);
            nglobal_pivots.push_back(pv.size());
        } else {
            nglobal_pivots.push_back(0);
        }
        const size_t iter = it + 1, mi = opt_.min_iters;
        bool converged = iter >= mi && errors[iter - 1] <= opt_.tolerance; // convergence_criterion_like_julia :381-413
        if (converged) {
            const size_t base = ranks[iter - mi];
            for (size_t i = iter - mi; i < iter; ++i) converged = converged && ranks[i] <= base && nglobal_pivots[i] == 0;
        }
        if (converged) {
            termination = AciTermination::Converged;
            break;
        }
        bool saturated = opt_.has_max_bond_dim && ranks.size() >= mi; // rank_is_saturated :436-451
        if (saturated)
            for (size_t i = ranks.size() - mi; i < ranks.size(); ++i) saturated = saturated && ranks[i] >= opt_.max_bond_dim;
        if (saturated) {
            termination = AciTermination::RankLimited;
            break;
        }
    }
    if (opt_.has_max_bond_dim && rank() > opt_.max_bond_dim) // cleanup sweep after a late injection (:197-209)
        for (size_t b = 0; b + 1 < n; ++b) local_update(b, true);
}

TreeAciLocalResult treeaci_local_update(Engine& eng, const std::vector<size_t>& bond_dims, const std::vector<const double*>& row_frames,
                                        const std::vector<const double*>& col_frames, size_t row_count, size_t col_count, AciOpKind kind,
                                        const AciHostOp& host_op, size_t max_bond_dim, double tolerance, bool scale_tolerance,
                                        bool left_orthogonal)
{
    const size_t K = bond_dims.size();
    if (K == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "at least one input is required"); // TreeAciError::NoInputs
    if (K > (size_t)ACI_MAX_INPUTS) throw Error(T4A_GPU_NOT_IMPLEMENTED, "more than eight inputs are not supported");
    if (row_frames.size() != K || col_frames.size() != K) throw Error(T4A_GPU_INVALID_ARGUMENT, "one row / column frame block per input");
    if (kind == AciOpKind::Callback && !host_op) throw Error(T4A_GPU_NULL_POINTER, "operator callback is null");
    if (row_count == 0 || col_count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "the local matrix has no rows or no columns");
    if (row_count > 65535 || col_count > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "local blocks above 65535 rows or columns are not supported");
    const size_t np = row_count * col_count;
    hipStream_t st = eng.stream();
    // frames -> device: the row frames transposed into (row_count x bond) factors, the column frames as they are
    size_t total = 0;
    for (size_t k = 0; k < K; ++k) total += bond_dims[k] * (2 * row_count + col_count);
    DevBuf<double> buf, vals;
    buf.reserve(std::max<size_t>(total, 1));
    AciPiArgs a{};
    double* cur = buf.get();
    for (size_t k = 0; k < K; ++k) {
        const size_t b = bond_dims[k];
        if (b > 0x7FFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "bond dimension out of range");
        double* d_rf = cur;                      // bond x row_count, as uploaded
        double* d_lf = d_rf + b * row_count;     // row_count x bond
        double* d_cf = d_lf + b * row_count;     // bond x col_count
        cur = d_cf + b * col_count;
        if (b) {
            T4A_HIP(hipMemcpyAsync(d_rf, row_frames[k], b * row_count * sizeof(double), hipMemcpyHostToDevice, st));
            T4A_HIP(hipMemcpyAsync(d_cf, col_frames[k], b * col_count * sizeof(double), hipMemcpyHostToDevice, st));
            transpose_launch(d_rf, (int)b, (int)row_count, (int)b, d_lf, (int)row_count, st);
        }
        a.lf[k] = d_lf;
        a.rf[k] = d_cf;
        a.mid[k] = (int)b;
    }
    a.n_inputs = (int)K;
    a.nrows = (int)row_count;
    a.ncols = (int)col_count;
    a.op = (int)kind;
    double* d_pi = eng.pi(np);
    a.pi = d_pi;
    if (kind == AciOpKind::Callback) {
        vals.reserve(K * np);
        a.vals = vals.get();
    }
    const unsigned blocks = (unsigned)std::min<size_t>((np + 255) / 256, 8192);
    hipLaunchKernelGGL(aci_pi_kernel, dim3(blocks), dim3(256), 0, st, a);
    T4A_HIP(hipGetLastError());
    TreeAciLocalResult out;
    out.local_values.resize(np);
    if (kind == AciOpKind::Callback) {
        std::vector<double> hv(K * np);
        T4A_HIP(hipMemcpyAsync(hv.data(), vals.get(), hv.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        eng.sync();
        host_op(hv.data(), K, np, out.local_values.data());
        T4A_HIP(hipMemcpyAsync(d_pi, out.local_values.data(), np * sizeof(double), hipMemcpyHostToDevice, st));
        eng.sync();
    } else {
        T4A_HIP(hipMemcpyAsync(out.local_values.data(), d_pi, np * sizeof(double), hipMemcpyDeviceToHost, st));
        eng.sync();
    }
    for (double v : out.local_values) out.sampled_scale = std::fmax(out.sampled_scale, std::fabs(v)); // (fold with f64::max: a NaN is dropped)
    RrLUOptions lo;
    lo.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
    lo.rel_tol = scale_tolerance ? tolerance : 0.0;
    lo.abs_tol = scale_tolerance ? 0.0 : tolerance;
    lo.left_orthogonal = left_orthogonal;
    LuciResult r = eng.luci(d_pi, (int)row_count, (int)col_count, lo, true, false);
    out.pivot_errors = r.pivot_errors;
    if (r.rank == 0) { // local_update.rs:230-238
        out.rank = 1;
        out.row_indices = {0};
        out.col_indices = {0};
        out.left.assign(row_count, 0.0);
        out.right.assign(col_count, 0.0);
        return out;
    }
    out.rank = (size_t)r.rank;
    out.row_indices.assign(r.row_perm.begin(), r.row_perm.begin() + r.rank);
    out.col_indices.assign(r.col_perm.begin(), r.col_perm.begin() + r.rank);
    out.left.resize(row_count * out.rank);
    out.right.resize(out.rank * col_count);
    T4A_HIP(hipMemcpyAsync(out.left.data(), eng.left(), out.left.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    T4A_HIP(hipMemcpyAsync(out.right.data(), eng.right(), out.right.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    eng.sync();
    return out;
}

std::unique_ptr<TensorTrain> aci_one_site(const std::vector<TensorTrain*>& inputs, AciOpKind kind, const AciHostOp& host_op)
{
    validate_inputs(inputs);
    if (kind == AciOpKind::Callback && !host_op) throw Error(T4A_GPU_NULL_POINTER, "operator callback is null");
    const size_t K = inputs.size(), np = inputs[0]->cores[0].s;
    std::vector<uint32_t> idx(np);
    for (size_t p = 0; p < np; ++p) idx[p] = (uint32_t)p;
    std::vector<double> iv(K * np), out(np);
    for (size_t k = 0; k < K; ++k) {
        const std::vector<double> v = inputs[k]->evaluate(idx.data(), np);
        for (size_t p = 0; p < np; ++p) iv[k + K * p] = v[p];
    }
    if (kind == AciOpKind::Callback) host_op(iv.data(), K, np, out.data());
    else host_builtin(kind, iv.data(), K, np, out.data());
    return std::make_unique<TensorTrain>(std::vector<std::array<size_t, 3>>{{1, np, 1}}, out.data());
}

} // namespace t4a
