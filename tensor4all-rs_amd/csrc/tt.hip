// tt.hip — SimpleTensorTrain<f64> on the device (see tt.hpp): evaluate / sum / norm2 / compress /
// TTCache::evaluate_many.  Index bookkeeping (unique halves, split heuristic) is host integer work,
// every floating-point operation runs in the gfx950 kernels of kernels_tt.hip / kernels_dense.hip /
// kernels_linalg.hip / kernels_rrlu*.hip.
#include "tt.hpp"
#include "tensorops.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>

namespace t4a {

namespace {
void copy_core(DevCore& dst, const DevCore& src, hipStream_t st)
{
    dst.l = src.l;
    dst.s = src.s;
    dst.r = src.r;
    dst.buf.reserve(std::max<size_t>(src.size(), 1));
    if (src.size())
        T4A_HIP(hipMemcpyAsync(dst.buf.get(), src.buf.get(), src.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
}

void validate_chain(const std::vector<DevCore>& cores) // SimpleTensorTrain::new, tensortrain.rs:97-124
{
    if (cores.empty()) return;
    if (cores.front().l != 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "First tensor must have left dimension 1");
    if (cores.back().r != 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "Last tensor must have right dimension 1");
    for (size_t i = 0; i + 1 < cores.size(); ++i)
        if (cores[i].r != cores[i + 1].l)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "tensor train bond dimension mismatch at bond " + std::to_string(i));
}

// FNV-1a over a run of digits
inline uint64_t hash_digits(const uint32_t* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= (uint64_t)p[i] + 0x9E3779B97F4A7C15ull;
        h *= 1099511628211ull;
    }
    return h;
}

// First-occurrence unique map of fixed-width digit strings (cache.rs IndexMapper): `first` receives the point
// index of each unique string in order of first appearance, `which[p]` the position of point p's string.
struct UniqueMap {
    std::vector<uint32_t> first, which;
    void build(const uint32_t* idx, size_t stride, size_t off, size_t width, size_t n_pts)
    {
        first.clear();
        which.assign(n_pts, 0);
        size_t cap = 16;
        while (cap < 2 * n_pts + 2) cap <<= 1;
        std::vector<uint32_t> table(cap, 0xFFFFFFFFu);
        for (size_t p = 0; p < n_pts; ++p) {
            const uint32_t* key = idx + p * stride + off;
            size_t slot = (size_t)hash_digits(key, width) & (cap - 1);
            for (;;) {
                const uint32_t u = table[slot];
                if (u == 0xFFFFFFFFu) {
                    table[slot] = (uint32_t)first.size();
                    which[p] = (uint32_t)first.size();
                    first.push_back((uint32_t)p);
                    break;
                }
                const uint32_t* other = idx + (size_t)first[u] * stride + off;
                if (std::equal(key, key + width, other)) {
                    which[p] = u;
                    break;
                }
                slot = (slot + 1) & (cap - 1);
            }
        }
    }
};
} // namespace

TensorTrain::TensorTrain(const std::vector<std::array<size_t, 3>>& dims3, const double* host_data)
{
    cores.resize(dims3.size());
    size_t off = 0;
    for (size_t s = 0; s < dims3.size(); ++s) {
        DevCore& c = cores[s];
        c.l = dims3[s][0];
        c.s = dims3[s][1];
        c.r = dims3[s][2];
        if (c.l > 65535 || c.r > 65535 || c.s > 65535)
            throw Error(T4A_GPU_NOT_IMPLEMENTED, "tensor train dimensions above 65535 are not supported");
        off += c.size();
    }
    validate_chain(cores);
    if (off && !host_data) throw Error(T4A_GPU_NULL_POINTER, "core data is null");
    off = 0;
    for (auto& c : cores) {
        c.buf.reserve(std::max<size_t>(c.size(), 1));
        if (c.size())
            T4A_HIP(hipMemcpyAsync(c.buf.get(), host_data + off, c.size() * sizeof(double), hipMemcpyHostToDevice,
                                   eng.stream()));
        off += c.size();
    }
    eng.sync();
}

TensorTrain::TensorTrain(const std::vector<DevCore>& src, hipStream_t src_stream)
{
    if (src_stream) T4A_HIP(hipStreamSynchronize(src_stream));
    validate_chain(src);
    cores.resize(src.size());
    for (size_t s = 0; s < src.size(); ++s) copy_core(cores[s], src[s], eng.stream());
    eng.sync();
}

std::vector<size_t> TensorTrain::link_dims() const // traits.rs:89-97
{
    std::vector<size_t> v;
    for (size_t i = 1; i < cores.size(); ++i) v.push_back(cores[i].l);
    return v;
}
std::vector<size_t> TensorTrain::site_dims() const
{
    std::vector<size_t> v;
    for (const auto& c : cores) v.push_back(c.s);
    return v;
}
size_t TensorTrain::rank() const // traits.rs:116-123
{
    size_t r = 1;
    const auto ld = link_dims();
    if (!ld.empty()) r = *std::max_element(ld.begin(), ld.end());
    return r;
}
size_t TensorTrain::max_bond() const
{
    size_t m = 1;
    for (const auto& c : cores) m = std::max(m, std::max(c.l, c.r));
    return m;
}

std::vector<double> TensorTrain::site_tensor_host(size_t site)
{
    if (site >= cores.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, "site out of range");
    const DevCore& c = cores[site];
    std::vector<double> h(c.size());
    if (!h.empty()) {
        T4A_HIP(hipMemcpyAsync(h.data(), c.buf.get(), h.size() * sizeof(double), hipMemcpyDeviceToHost, eng.stream()));
        eng.sync();
    }
    return h;
}

void TensorTrain::upload_descs()
{
    const size_t n = cores.size();
    std::vector<TtCoreDesc> desc(n);
    for (size_t s = 0; s < n; ++s) {
        desc[s].data = cores[s].buf.get();
        desc[s].l = (int)cores[s].l;
        desc[s].d = (int)cores[s].s;
        desc[s].r = (int)cores[s].r;
    }
    d_desc_.reserve(std::max<size_t>(n, 1));
    T4A_HIP(hipMemcpyAsync(d_desc_.get(), desc.data(), n * sizeof(TtCoreDesc), hipMemcpyHostToDevice, eng.stream()));
    eng.sync(); // `desc` is pageable host memory
}

std::vector<double> TensorTrain::evaluate(const uint32_t* idx, size_t n_pts) // traits.rs:146-212
{
    const size_t n = cores.size();
    if (n == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate: empty tensor train");
    std::vector<double> out(n_pts);
    if (n_pts == 0) return out;
    for (size_t p = 0; p < n_pts; ++p)
        for (size_t s = 0; s < n; ++s)
            if (idx[p * n + s] >= cores[s].s) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate: index out of bounds");
    hipStream_t st = eng.stream();
    upload_descs();
    d_idx_.reserve(n_pts * n);
    d_vals_.reserve(n_pts);
    T4A_HIP(hipMemcpyAsync(d_idx_.get(), idx, n_pts * n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    tt_evaluate_launch(d_desc_.get(), (int)n, (int)max_bond(), d_idx_.get(), (int)n_pts, d_vals_.get(), st);
    T4A_HIP(hipMemcpyAsync(out.data(), d_vals_.get(), n_pts * sizeof(double), hipMemcpyDeviceToHost, st));
    eng.sync();
    T4A_HIP(hipGetLastError());
    return out;
}

double TensorTrain::sum() // traits.rs:231-275
{
    if (cores.empty()) return 0.0;
    for (const auto& c : cores)
        if (c.size() == 0) return 0.0;
    upload_descs();
    d_vals_.reserve(1);
    tt_sum_launch(d_desc_.get(), (int)cores.size(), (int)max_bond(), d_vals_.get(), eng.stream());
    double v = 0.0;
    T4A_HIP(hipMemcpyAsync(&v, d_vals_.get(), sizeof(double), hipMemcpyDeviceToHost, eng.stream()));
    eng.sync();
    T4A_HIP(hipGetLastError());
    return v;
}

double TensorTrain::norm2() // traits.rs:289-354
{
    if (cores.empty()) return 0.0;
    const size_t mb = max_bond();
    d_m1_.reserve(mb * mb);
    d_m2_.reserve(mb * mb);
    double* cur = d_m1_.get();
    double* nxt = d_m2_.get();
    for (size_t s = 0; s < cores.size(); ++s) {
        TtCoreDesc c;
        c.data = cores[s].buf.get();
        c.l = (int)cores[s].l;
        c.d = (int)cores[s].s;
        c.r = (int)cores[s].r;
        tt_norm2_step_launch(c, cur, s == 0, nxt, eng.stream());
        std::swap(cur, nxt);
    }
    double v = 0.0;
    T4A_HIP(hipMemcpyAsync(&v, cur, sizeof(double), hipMemcpyDeviceToHost, eng.stream()));
    eng.sync();
    T4A_HIP(hipGetLastError());
    return std::sqrt(v * v);
}

// ------------------------------------------------------------------------------------------------
// compression.rs:165-340
// ------------------------------------------------------------------------------------------------
size_t TensorTrain::factorize(const double* d_mat, int M, int N, CompressionMethod method, double tolerance,
                              bool normalize_error, size_t max_bond_dim, bool left_orthogonal)
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
    switch (method) {
    case CompressionMethod::LU: {
        LuciResult r = eng.luci(d_mat, M, N, o, false, true);
        eng.lu_permuted_factors(r, left_orthogonal);
        return (size_t)r.rank;
    }
    case CompressionMethod::CI: {
        LuciResult r = eng.luci(d_mat, M, N, o, true, false);
        return (size_t)r.rank;
    }
    case CompressionMethod::SVD: {
        if (M == 0 || N == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "Cannot factorize empty matrix");
        const int k = std::min(M, N);
        d_svdu_.reserve((size_t)M * k);
        d_svds_.reserve(k);
        d_svdvt_.reserve((size_t)k * N);
        eng.svd(d_mat, M, N, d_svdu_.get(), d_svds_.get(), d_svdvt_.get());
        std::vector<double> s(k);
        T4A_HIP(hipMemcpyAsync(s.data(), d_svds_.get(), sizeof(double) * k, hipMemcpyDeviceToHost, eng.stream()));
        eng.sync();
        const double s_max = s[0];
        const double threshold = normalize_error ? tolerance * s_max : tolerance;
        size_t rank = 0;
        for (int i = 0; i < k; ++i) {
            if (max_bond_dim != 0 && rank >= max_bond_dim) break;
            if (s[i] < threshold) break;
            ++rank;
        }
        rank = std::max<size_t>(rank, 1);
        // left (M x rank) = U[:, :rank] (* S if right-orthogonal); right (rank x N) = (S *) Vt[:rank, :]
        // the scaling is a GEMM with diag(S[:rank]) built on the device
        eng.reserve_factors((size_t)M * rank, (size_t)rank * N);
        double* left = eng.left();
        double* right = eng.right();
        eng.d_tmp.reserve(rank * rank);
        double* dg = eng.d_tmp.get();
        std::vector<double> hd(rank * rank, 0.0);
        for (size_t i = 0; i < rank; ++i) hd[i + rank * i] = s[i];
        T4A_HIP(hipMemcpyAsync(dg, hd.data(), sizeof(double) * rank * rank, hipMemcpyHostToDevice, eng.stream()));
        eng.sync();
        GemmDesc g;
        g.strideA = g.strideB = g.strideC = 0;
        g.transA = g.transB = 0;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        if (left_orthogonal) {
            gather_launch(d_svdu_.get(), M, nullptr, M, nullptr, (int)rank, left, M, eng.stream());
            g.m = (int)rank;
            g.n = N;
            g.k = (int)rank;
            g.A = dg;
            g.lda = (int)rank;
            g.B = d_svdvt_.get();
            g.ldb = k;
            g.C = right;
            g.ldc = (int)rank;
            gemm_launch(g, eng.stream());
        } else {
            g.m = M;
            g.n = (int)rank;
            g.k = (int)rank;
            g.A = d_svdu_.get();
            g.lda = M;
            g.B = dg;
            g.ldb = (int)rank;
            g.C = left;
            g.ldc = M;
            gemm_launch(g, eng.stream());
            gather_launch(d_svdvt_.get(), k, nullptr, (int)rank, nullptr, N, right, (int)rank, eng.stream());
        }
        T4A_HIP(hipGetLastError());
        return rank;
    }
    }
    throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown compression method");
}

void TensorTrain::compress(const CompressionOptions& opt) // compression.rs:375-507
{
    const size_t n = cores.size();
    if (n <= 1) return;
    hipStream_t st = eng.stream();
    auto gemm = [&](const double* A, int m, int k, const double* B, int nn, double* C) {
        GemmDesc g;
        g.m = m;
        g.n = nn;
        g.k = k;
        g.A = A;
        g.lda = m;
        g.strideA = 0;
        g.transA = 0;
        g.B = B;
        g.ldb = k;
        g.strideB = 0;
        g.transB = 0;
        g.C = C;
        g.ldc = m;
        g.strideC = 0;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        gemm_launch(g, st);
    };
    // left-to-right: orthogonalise without truncation
    for (size_t ell = 0; ell + 1 < n; ++ell) {
        DevCore& c = cores[ell];
        DevCore& nx = cores[ell + 1];
        const int L = (int)c.l, S = (int)c.s, R = (int)c.r;
        d_m1_.reserve(std::max<size_t>(c.size(), 1));
        core_reshape_launch(c.buf.get(), L, S, R, 0, d_m1_.get(), st);
        const size_t rk = factorize(d_m1_.get(), L * S, R, opt.method, 0.0, true, 0, true);
        if (rk == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "compress: factorisation returned rank 0 (zero bond matrix)");
        // current <- left factor (L*S x rk)
        DevCore nc;
        nc.l = L;
        nc.s = S;
        nc.r = rk;
        nc.buf.reserve(std::max<size_t>(nc.size(), 1));
        core_reshape_launch(eng.left(), L, S, (int)rk, 1, nc.buf.get(), st);
        // next <- right factor (rk x R) * right matrix of next (R x S'*R')
        const int NS = (int)nx.s, NR = (int)nx.r;
        d_m1_.reserve(std::max<size_t>(nx.size(), 1));
        core_reshape_launch(nx.buf.get(), (int)nx.l, NS, NR, 2, d_m1_.get(), st);
        d_m2_.reserve(std::max<size_t>(rk * NS * NR, 1));
        gemm(eng.right(), (int)rk, R, d_m1_.get(), NS * NR, d_m2_.get());
        DevCore nn;
        nn.l = rk;
        nn.s = NS;
        nn.r = NR;
        nn.buf.reserve(std::max<size_t>(nn.size(), 1));
        core_reshape_launch(d_m2_.get(), (int)rk, NS, NR, 3, nn.buf.get(), st);
        eng.sync(); // the old buffers are released below
        cores[ell] = std::move(nc);
        cores[ell + 1] = std::move(nn);
    }
    // right-to-left: truncate
    for (size_t ell = n - 1; ell >= 1; --ell) {
        DevCore& c = cores[ell];
        DevCore& pv = cores[ell - 1];
        const int L = (int)c.l, S = (int)c.s, R = (int)c.r;
        d_m1_.reserve(std::max<size_t>(c.size(), 1));
        core_reshape_launch(c.buf.get(), L, S, R, 2, d_m1_.get(), st);
        const size_t rk = factorize(d_m1_.get(), L, S * R, opt.method, opt.tolerance, opt.normalize_error,
                                    opt.max_bond_dim, false);
        if (rk == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "compress: factorisation returned rank 0 (zero bond matrix)");
        DevCore nc;
        nc.l = rk;
        nc.s = S;
        nc.r = R;
        nc.buf.reserve(std::max<size_t>(nc.size(), 1));
        core_reshape_launch(eng.right(), (int)rk, S, R, 3, nc.buf.get(), st);
        const int PL = (int)pv.l, PS = (int)pv.s;
        d_m1_.reserve(std::max<size_t>(pv.size(), 1));
        core_reshape_launch(pv.buf.get(), PL, PS, (int)pv.r, 0, d_m1_.get(), st);
        d_m2_.reserve(std::max<size_t>((size_t)PL * PS * rk, 1));
        gemm(d_m1_.get(), PL * PS, L, eng.left(), (int)rk, d_m2_.get());
        DevCore np;
        np.l = PL;
        np.s = PS;
        np.r = rk;
        np.buf.reserve(std::max<size_t>(np.size(), 1));
        core_reshape_launch(d_m2_.get(), PL, PS, (int)rk, 1, np.buf.get(), st);
        eng.sync();
        cores[ell] = std::move(nc);
        cores[ell - 1] = std::move(np);
    }
    T4A_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// cache.rs:558-744
// ------------------------------------------------------------------------------------------------
size_t TensorTrain::find_split_heuristic(const uint32_t* idx, size_t n_pts) const
{
    const size_t n = cores.size();
    if (n <= 1) return std::max<size_t>(n, 1);
    const size_t cand[3] = {n / 4, n / 2, n * 3 / 4};
    bool have = false;
    size_t best_p = 0, best_c = 0;
    UniqueMap ul, ur;
    for (size_t p : cand) {
        if (p < 1 || p >= n) continue;
        ul.build(idx, n, 0, p, n_pts);
        ur.build(idx, n, p, n - p, n_pts);
        const size_t c = ul.first.size() + ur.first.size();
        if (!have || c < best_c) { // min_by_key keeps the first minimum
            have = true;
            best_p = p;
            best_c = c;
        }
    }
    if (!have) throw Error(T4A_GPU_INTERNAL_ERROR, "cache heuristic could not choose a valid split");
    return best_p;
}

size_t TensorTrain::evaluate_many(const uint32_t* idx, size_t n_pts, size_t split, double* out)
{
    const size_t n = cores.size();
    if (n_pts == 0) return split;
    if (n == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate_many: empty tensor train");
    for (size_t p = 0; p < n_pts; ++p)
        for (size_t s = 0; s < n; ++s)
            if (idx[p * n + s] >= cores[s].s) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate_many: index out of bounds");
    if (split == 0) split = find_split_heuristic(idx, n_pts);
    if (split == 0 || split > n)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid split position: " + std::to_string(split) +
                                                  " (n_sites=" + std::to_string(n) + ")");
    hipStream_t st = eng.stream();
    UniqueMap ul, ur;
    ul.build(idx, n, 0, split, n_pts);
    ur.build(idx, n, split, n - split, n_pts);
    const size_t nl = ul.first.size(), nr = ur.first.size();
    const size_t wl = split, wr = n - split;
    // pack the unique halves
    std::vector<uint32_t> hl(nl * wl), hr(std::max<size_t>(nr * wr, 1));
    for (size_t u = 0; u < nl; ++u) std::copy_n(idx + (size_t)ul.first[u] * n, wl, hl.data() + u * wl);
    for (size_t u = 0; u < nr; ++u) std::copy_n(idx + (size_t)ur.first[u] * n + split, wr, hr.data() + u * wr);
    const size_t bond = split < n ? cores[split].l : 1; // length of the environments
    const size_t ld = bond;
    upload_descs();
    d_idx_.reserve(hl.size() + hr.size());
    d_il_.reserve(n_pts);
    d_ir_.reserve(n_pts);
    d_envl_.reserve(std::max<size_t>(nl * ld, 1));
    d_envr_.reserve(std::max<size_t>(nr * ld, 1));
    d_vals_.reserve(n_pts);
    uint32_t* d_hl = d_idx_.get();
    uint32_t* d_hr = d_hl + hl.size();
    T4A_HIP(hipMemcpyAsync(d_hl, hl.data(), hl.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (wr) T4A_HIP(hipMemcpyAsync(d_hr, hr.data(), nr * wr * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    T4A_HIP(hipMemcpyAsync(d_il_.get(), ul.which.data(), n_pts * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    T4A_HIP(hipMemcpyAsync(d_ir_.get(), ur.which.data(), n_pts * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    const int mb = (int)max_bond();
    tt_env_left_launch(d_desc_.get(), (int)split, mb, d_hl, (int)nl, d_envl_.get(), (int)ld, st);
    if (wr)
        tt_env_right_launch(d_desc_.get(), (int)n, (int)split, mb, d_hr, (int)nr, d_envr_.get(), (int)ld, st);
    else
        fill_launch(d_envr_.get(), nr * ld, 1.0, st); // evaluate_right(&[]) == [1] (cache.rs:479-481)
    tt_env_dot_launch(d_envl_.get(), d_envr_.get(), (int)bond, (int)ld, d_il_.get(), d_ir_.get(), n_pts, d_vals_.get(),
                      st);
    T4A_HIP(hipMemcpyAsync(out, d_vals_.get(), n_pts * sizeof(double), hipMemcpyDeviceToHost, st));
    eng.sync();
    T4A_HIP(hipGetLastError());
    return split;
}

// ---------------------------------------------------------------------------------------------
// arithmetic (simplett/src/arithmetic.rs:34-180, tensortrain.rs:264-345, :449-583)
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) tt_scale_kernel(double* p, size_t n, double f)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = p[e] * f;
}
__global__ void __launch_bounds__(256) tt_add2_kernel(const double* a, const double* b, double* out, size_t n)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) out[e] = a[e] + b[e];
}
// out (l x r) = sum over the site index, s ascending from 0.0 (tensortrain.rs:486-496)
__global__ void __launch_bounds__(256) tt_site_sum_kernel(const double* core, int l, int s, int r, double* out)
{
    const size_t total = (size_t)l * r;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t li = e % (size_t)l, ri = e / (size_t)l;
        double acc = 0.0;
        for (int q = 0; q < s; ++q) acc = acc + core[li + (size_t)l * (q + (size_t)s * ri)];
        out[e] = acc;
    }
}
unsigned blocks_for(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 4096); }
} // namespace

void TensorTrain::scale(double factor)
{
    if (cores.empty()) return;
    DevCore& c = cores.back();
    if (c.size()) hipLaunchKernelGGL(tt_scale_kernel, dim3(blocks_for(c.size())), dim3(256), 0, eng.stream(), c.buf.get(), c.size(), factor);
    T4A_HIP(hipGetLastError());
    eng.sync();
}

std::unique_ptr<TensorTrain> TensorTrain::add(TensorTrain& other, bool subtract)
{
    if (len() != other.len())
        throw Error(T4A_GPU_INVALID_ARGUMENT, "Cannot add tensor trains of different lengths: " + std::to_string(len()) + " vs " +
                                                  std::to_string(other.len()));
    other.eng.sync();
    if (cores.empty()) return std::make_unique<TensorTrain>(other.cores, other.eng.stream());
    const size_t n = len();
    for (size_t i = 0; i < n; ++i)
        if (cores[i].s != other.cores[i].s)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Site dimensions mismatch at site " + std::to_string(i) + ": " + std::to_string(cores[i].s) +
                                                      " vs " + std::to_string(other.cores[i].s));
    hipStream_t st = eng.stream();
    // sub = add(other.scale(-1)) (arithmetic.rs:161-175): the factor sits on the last core of `other`
    DevBuf<double> neg;
    const DevCore& ol = other.cores.back();
    const double* other_last = ol.buf.get();
    if (subtract) {
        neg.reserve(std::max<size_t>(ol.size(), 1));
        T4A_HIP(hipMemcpyAsync(neg.get(), ol.buf.get(), ol.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (ol.size()) hipLaunchKernelGGL(tt_scale_kernel, dim3(blocks_for(ol.size())), dim3(256), 0, st, neg.get(), ol.size(), -1.0);
        other_last = neg.get();
    }
    std::vector<DevCore> out(n);
    for (size_t i = 0; i < n; ++i) {
        const DevCore& x = cores[i];
        const DevCore& y = other.cores[i];
        const double* yp = i == n - 1 ? other_last : y.buf.get();
        const bool first = i == 0, last = i == n - 1;
        DevCore& t = out[i];
        t.l = first ? 1 : x.l + y.l;
        t.s = x.s;
        t.r = last ? 1 : x.r + y.r;
        t.buf.reserve(std::max<size_t>(t.size(), 1));
        if (first && last) {
            hipLaunchKernelGGL(tt_add2_kernel, dim3(blocks_for(t.size())), dim3(256), 0, st, x.buf.get(), yp, t.buf.get(), t.size());
            continue;
        }
        fill_launch(t.buf.get(), t.size(), 0.0, st);
        // a core (l, s, r) viewed as l x (s r): the block of `other` starts at row l0 and column s * r0
        const size_t l0 = first ? 0 : x.l, r0 = last ? 0 : x.r;
        gather_launch(x.buf.get(), (int)x.l, nullptr, (int)x.l, nullptr, (int)(x.s * x.r), t.buf.get(), (int)t.l, st);
        gather_launch(yp, (int)y.l, nullptr, (int)y.l, nullptr, (int)(y.s * y.r), t.buf.get() + l0 + t.l * (t.s * r0), (int)t.l, st);
    }
    T4A_HIP(hipGetLastError());
    eng.sync();
    return std::make_unique<TensorTrain>(out, st);
}

double TensorTrain::inner_product(TensorTrain& other)
{
    if (len() != other.len())
        throw Error(T4A_GPU_INVALID_ARGUMENT, "Cannot compute inner_product product of tensor trains with different lengths: " +
                                                  std::to_string(len()) + " vs " + std::to_string(other.len()));
    if (cores.empty()) return 0.0;
    other.eng.sync();
    hipStream_t st = eng.stream();
    DevBuf<double> env, tmp, nxt;
    env.reserve(1);
    fill_launch(env.get(), 1, 1.0, st);
    size_t ea = 1, eb = 1; // env is ea x eb (bond of this, bond of other)
    for (size_t i = 0; i < len(); ++i) {
        const DevCore& x = cores[i];
        const DevCore& y = other.cores[i];
        if (x.s != y.s)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Site dimensions mismatch at site " + std::to_string(i) + ": " + std::to_string(x.s) + " vs " +
                                                      std::to_string(y.s));
        // tmp (eb x S R_a) = env^T * X,  then env' (R_a x R_b) = tmp^T (R_a x eb S) * Y (eb S x R_b)
        tmp.reserve(std::max<size_t>(eb * x.s * x.r, 1));
        nxt.reserve(std::max<size_t>(x.r * y.r, 1));
        GemmDesc g{};
        g.m = (int)eb;
        g.n = (int)(x.s * x.r);
        g.k = (int)ea;
        g.A = env.get();
        g.lda = (int)ea;
        g.transA = 1;
        g.B = x.buf.get();
        g.ldb = (int)x.l;
        g.transB = 0;
        g.C = tmp.get();
        g.ldc = (int)eb;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        gemm_launch(g, st);
        GemmDesc h{};
        h.m = (int)x.r;
        h.n = (int)y.r;
        h.k = (int)(eb * x.s);
        h.A = tmp.get();
        h.lda = (int)(eb * x.s);
        h.transA = 1;
        h.B = y.buf.get();
        h.ldb = (int)(y.l * y.s);
        h.transB = 0;
        h.C = nxt.get();
        h.ldc = (int)x.r;
        h.alpha = 1.0;
        h.beta = 0.0;
        h.batch = 1;
        gemm_launch(h, st);
        eng.sync();
        std::swap(env, nxt);
        ea = x.r;
        eb = y.r;
    }
    double v = 0.0;
    T4A_HIP(hipGetLastError());
    T4A_HIP(hipMemcpyAsync(&v, env.get(), sizeof(double), hipMemcpyDeviceToHost, st));
    eng.sync();
    return v;
}

std::unique_ptr<TensorTrain> TensorTrain::reverse()
{
    hipStream_t st = eng.stream();
    const size_t n = len();
    std::vector<DevCore> out(n);
    for (size_t i = 0; i < n; ++i) {
        const DevCore& x = cores[n - 1 - i];
        DevCore& t = out[i];
        t.l = x.r;
        t.s = x.s;
        t.r = x.l;
        t.buf.reserve(std::max<size_t>(t.size(), 1));
        TensorView v;
        v.d_data = x.buf.get();
        v.dims = {x.l, x.s, x.r};
        v.labels = {0, 1, 2};
        if (t.size()) tensor_permute(eng, v, {2, 1, 0}, t.buf.get());
    }
    T4A_HIP(hipGetLastError());
    eng.sync();
    return std::make_unique<TensorTrain>(out, st);
}

std::unique_ptr<TensorTrain> TensorTrain::partial_sum(const std::vector<size_t>& dims)
{
    const size_t n = len();
    hipStream_t st = eng.stream();
    if (n == 0) return std::make_unique<TensorTrain>(std::vector<DevCore>{}, st);
    for (size_t d : dims)
        if (d >= n) throw Error(T4A_GPU_INVALID_ARGUMENT, "Dimension " + std::to_string(d) + " out of range (0.." + std::to_string(n) + ")");
    std::vector<DevCore> out;
    DevBuf<double> tprod, next, ssum;
    size_t tr = 1, tc = 1; // tprod is tr x tc
    tprod.reserve(1);
    fill_launch(tprod.get(), 1, 1.0, st);
    for (size_t site = 0; site < n; ++site) {
        const DevCore& t = cores[site];
        if (std::find(dims.begin(), dims.end(), site) != dims.end()) {
            ssum.reserve(std::max<size_t>(t.l * t.r, 1));
            hipLaunchKernelGGL(tt_site_sum_kernel, dim3(blocks_for(t.l * t.r)), dim3(256), 0, st, t.buf.get(), (int)t.l, (int)t.s, (int)t.r,
                               ssum.get());
            next.reserve(std::max<size_t>(tr * t.r, 1));
            seq_matmul_launch(tprod.get(), (int)tr, ssum.get(), (int)t.l, next.get(), (int)tr, tr, t.r, t.l, st);
            eng.sync(); // `tprod` is replaced: nothing may still read the old buffer
            std::swap(tprod, next);
            tc = t.r;
        } else {
            DevCore c;
            c.l = tr;
            c.s = t.s;
            c.r = t.r;
            c.buf.reserve(std::max<size_t>(c.size(), 1));
            seq_matmul_launch(tprod.get(), (int)tr, t.buf.get(), (int)t.l, c.buf.get(), (int)tr, tr, t.s * t.r, t.l, st);
            out.push_back(std::move(c));
            tr = tc = t.r;
            eng.sync();
            tprod.reserve(tr * tr);
            set_identity_launch(tprod.get(), (int)tr, (int)tr, (int)tr, st);
        }
    }
    if (out.empty()) { // everything summed: a one-site train wrapping the scalar
        DevCore c;
        c.l = c.s = c.r = 1;
        c.buf.reserve(1);
        T4A_HIP(hipMemcpyAsync(c.buf.get(), tprod.get(), sizeof(double), hipMemcpyDeviceToDevice, st));
        out.push_back(std::move(c));
    } else { // the trailing product goes into the last kept core
        DevCore& last = out.back();
        DevBuf<double> nl;
        nl.reserve(std::max<size_t>(last.l * last.s * tc, 1));
        seq_matmul_launch(last.buf.get(), (int)(last.l * last.s), tprod.get(), (int)tr, nl.get(), (int)(last.l * last.s), last.l * last.s, tc,
                          last.r, st);
        eng.sync();
        last.buf = std::move(nl);
        last.r = tc;
    }
    T4A_HIP(hipGetLastError());
    eng.sync();
    return std::make_unique<TensorTrain>(out, st);
}

} // namespace t4a
