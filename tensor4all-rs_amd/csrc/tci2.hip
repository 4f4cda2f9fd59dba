// tci2.hip — TensorCI2 driver on the gfx950 engine.
// Mirrors crates/tensor4all-tensorci/src/tensorci2.rs function by function (line references inline).
#include "tci2.hpp"

#include <thread>

#include <exception>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <string>
#include <unordered_set>

namespace t4a {

extern thread_local double g_chain_wait_seconds; // tci2_chain.hip

// =================================================================================================
// small device kernels: packing of factor matrices into site tensors (column-major [left, site, right])
// =================================================================================================
namespace {

// core[l,s,r] = mat[(l*S+s) + ldm*r] if (l*S+s) < mrows && r < mcols else 0   (tensorci2.rs:1957-1973, :996-1011)
__global__ void __launch_bounds__(256) pack_left_core_kernel(const double* __restrict__ mat, int ldm, int mrows,
                                                             int mcols, double* __restrict__ core, int L, int S, int R)
{
    const size_t total = (size_t)L * S * R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % L);
        const int s = (int)((e / L) % S);
        const int r = (int)(e / ((size_t)L * S));
        const int row = l * S + s;
        core[e] = (row < mrows && r < mcols) ? mat[(size_t)r * ldm + row] : 0.0;
    }
}

// core[l,s,r] = mat[l + ldm*(s*R+r)] if l < mrows && (s*R+r) < mcols else 0   (tensorci2.rs:1983-1999, :1020-1036)
__global__ void __launch_bounds__(256) pack_right_core_kernel(const double* __restrict__ mat, int ldm, int mrows,
                                                              int mcols, double* __restrict__ core, int L, int S, int R)
{
    const size_t total = (size_t)L * S * R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % L);
        const int s = (int)((e / L) % S);
        const int r = (int)(e / ((size_t)L * S));
        const int col = s * R + r;
        core[e] = (l < mrows && col < mcols) ? mat[(size_t)col * ldm + l] : 0.0;
    }
}

// core[l,s,r] = xt[r + ldx*(l*S+s)]   (tensorci2.rs:1167-1181)
__global__ void __launch_bounds__(256) pack_fill_core_kernel(const double* __restrict__ xt, int ldx,
                                                             double* __restrict__ core, int L, int S, int R,
                                                             const int* info)
{
    const size_t total = (size_t)L * S * R;
    const bool zero = info && *info == -1; // numerically zero pivot matrix: zero core (tensorci2.rs:1154-1157)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % L);
        const int s = (int)((e / L) % S);
        const int r = (int)(e / ((size_t)L * S));
        core[e] = zero ? 0.0 : xt[(size_t)(l * S + s) * ldx + r];
    }
}

// all cores of one fill_site_tensors in a single launch (blockIdx.y = site job; PackJob: kernels.hpp)
__global__ void __launch_bounds__(256) pack_fill_batched_kernel(const PackJob* __restrict__ jobs)
{
    const PackJob j = jobs[blockIdx.y];
    const size_t total = (size_t)j.L * j.S * j.R;
    const bool zero = j.info && *j.info == -1; // numerically zero pivot matrix: zero core (tensorci2.rs:1154-1157)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % j.L);
        const int s = (int)((e / j.L) % j.S);
        const int r = (int)(e / ((size_t)j.L * j.S));
        if (j.last)
            j.core[e] = j.src[(size_t)r * j.ld + (l * j.S + s)]; // Pi1 itself (tensorci2.rs:1109-1128), R == 1
        else
            j.core[e] = zero ? 0.0 : j.src[(size_t)(l * j.S + s) * j.ld + r]; // (tensorci2.rs:1167-1181)
    }
}

inline unsigned blocks_for(size_t total)
{
    size_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b == 0) b = 1;
    return (unsigned)b;
}

inline uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

std::string key_of(const uint32_t* v, size_t w) { return std::string(reinterpret_cast<const char*>(v), w * sizeof(uint32_t)); }

std::vector<size_t> non_empty_or_first(const std::vector<int>& perm, int rank) // tensorci2.rs:1813-1819
{
    std::vector<size_t> v;
    for (int i = 0; i < rank; ++i) v.push_back((size_t)perm[i]);
    if (v.empty()) v.push_back(0);
    return v;
}

} // namespace

bool IndexSet::contains(const uint32_t* v) const
{
    if (width == 0) return count > 0;
    for (size_t k = 0; k < count; ++k)
        if (std::memcmp(at(k), v, width * sizeof(uint32_t)) == 0) return true;
    return false;
}

void TCI2Options::validate() const // tensorci2.rs:140-149
{
    auto nonneg_finite = [](const char* n, double v) {
        if (!(v >= 0.0) || !std::isfinite(v))
            throw Error(T4A_GPU_INVALID_ARGUMENT, std::string(n) + " must be finite and non-negative");
    };
    nonneg_finite("tolerance", tolerance);
    nonneg_finite("tol_margin_global_search", tol_margin_global_search);
    if (max_iter == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_iter must be positive");
    if (ncheck_history == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "ncheck_history must be positive");
}

// =================================================================================================
Tci2::Tci2(const std::vector<size_t>& dims) : n_(dims.size()), local_dims(dims) // tensorci2.rs:380-404
{
    if (dims.size() < 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
    for (size_t s = 0; s < dims.size(); ++s)
        if (dims[s] == 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension at site " + std::to_string(s) + " must be positive");
    i_set.resize(n_);
    j_set.resize(n_);
    for (size_t p = 0; p < n_; ++p) {
        i_set[p].width = p;
        j_set[p].width = n_ - p - 1;
    }
    cores.resize(n_);
    for (size_t p = 0; p < n_; ++p) {
        cores[p].l = 0;
        cores[p].s = dims[p];
        cores[p].r = 0;
    }
    bond_errors.assign(n_ - 1, 0.0);
    offset_.resize(n_);
    total_ = 0;
    for (size_t s = 0; s < n_; ++s) {
        offset_[s] = total_;
        total_ += dims[s];
    }
    last_sweep_shapes.assign(n_ - 1, {0, 0, 0});
    d_maxbits_.reserve(2);
    ev_pi_.init();
    ev_fill_.init();
}

Tci2::~Tci2()
{
    chain_abort();
    for (hipEvent_t e : chain_.t0) (void)hipEventDestroy(e);
    for (hipEvent_t e : chain_.t1) (void)hipEventDestroy(e);
    if (chain_.group_ev) (void)hipEventDestroy(chain_.group_ev);
    if (export_event_) (void)hipEventDestroy(export_event_);
    if (import_event_) (void)hipEventDestroy(import_event_);
    if (import_stream_) pool::stream_put(import_stream_, 2); // (synchronises it)
    if (fill_graph_exec_) (void)hipGraphExecDestroy(fill_graph_exec_);
    if (fill_stream_) pool::stream_put(fill_stream_, 2); // (synchronises it)
    // every buffer of this handle (and of its engine) was only ever used on the three streams that are idle now: the blocks
    // go back to the process-wide cache without a device-wide synchronisation, which would wait for the other handles' chains
    if (hipStreamSynchronize(eng.stream()) == hipSuccess) idle_scope_.arm();
}

void Tci2::set_builtin(int fid, int n_acc, const double* params, const uint64_t* weights)
{
    if (fid < 0 || fid >= T4A_FN_COUNT) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown built-in function id");
    if (n_acc < 1 || n_acc > T4A_FN_MAX_ACC) throw Error(T4A_GPU_INVALID_ARGUMENT, "n_acc out of range");
    // after a bond chain the pinned mirror is the master copy of the index sets, with accumulators of the OLD weights laid out
    // for the old n_acc: decode the digit tables first (as set_callback does), so that everything below re-accumulates from
    // digits with the new weights
    sync_digits();
    invalidate_fill_cache();
    fn_dev_.fid = fid;
    fn_dev_.n_acc = n_acc;
    std::memcpy(fn_dev_.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
    weights_.assign(weights, weights + (size_t)n_acc * total_);
    fn_kind_ = FnKind::Builtin;
    chain_.weights_valid = false; // (the accumulators in the device tables belong to the old weights)
    chain_.tables_valid = false;
    chain_.snap_serial[0] = chain_.snap_serial[1] = ~0ull;
    // the functor's weights and the site table go to the device HERE, with the function they belong to (round 5: the first bond chain of a
    // handle used to upload them — two copies and a stream synchronisation, ~35 us inside the first iteration of every solve)
    {
        hipStream_t st = eng.stream();
        chain_.weights.reserve(weights_.size());
        chain_.siteinfo.reserve(2 * n_);
        std::vector<int> si(2 * n_);
        for (size_t p = 0; p < n_; ++p) {
            si[p] = (int)local_dims[p];
            si[n_ + p] = (int)offset_[p];
        }
        T4A_HIP(hipMemcpyAsync(chain_.weights.get(), weights_.data(), weights_.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(chain_.siteinfo.get(), si.data(), si.size() * sizeof(int), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
        chain_.weights_valid = true;
    }
}

void Tci2::set_callback(t4a_gpu_batch_eval_fn cb, void* ctx)
{
    if (!cb) throw Error(T4A_GPU_NULL_POINTER, "callback is null");
    sync_digits();
    cb_ = cb;
    cb_ctx_ = ctx;
    fn_kind_ = FnKind::Callback;
}

void Tci2::require_fn() const
{
    if (fn_kind_ == FnKind::None)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "no function set: call t4a_gpu_tci2_set_builtin_function or _set_callback");
}

size_t Tci2::rank() const // tensorci2.rs:600-614
{
    size_t r = 0;
    for (size_t p = 1; p < n_; ++p) r = std::max(r, i_set[p].count);
    return r;
}

std::vector<size_t> Tci2::link_dims() const
{
    std::vector<size_t> v;
    for (size_t p = 1; p < n_; ++p) v.push_back(i_set[p].count);
    return v;
}

double Tci2::max_bond_error() const // tensorci2.rs:630
{
    double m = 0.0;
    for (double e : bond_errors) m = std::fmax(m, e);
    return m;
}

void Tci2::invalidate_site_tensors() // tensorci2.rs:724-728
{
    for (size_t p = 0; p < n_; ++p) {
        cores[p].l = 0;
        cores[p].s = local_dims[p];
        cores[p].r = 0;
    }
}

void Tci2::add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots) // tensorci2.rs:668-711
{
    for (const auto& p : pivots) {
        if (p.size() != n_)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Pivot length (" + std::to_string(p.size()) +
                                                      ") must match number of sites (" + std::to_string(n_) + ")");
        for (size_t s = 0; s < n_; ++s)
            if (p[s] >= local_dims[s])
                throw Error(T4A_GPU_INVALID_ARGUMENT, "pivot value " + std::to_string(p[s]) +
                                                          " is out of bounds at site " + std::to_string(s));
    }
    if (!pivots.empty()) sync_digits();
    for (const auto& pivot : pivots) {
        for (size_t p = 0; p < n_; ++p) {
            const uint32_t* pre = pivot.data();
            const uint32_t* suf = pivot.data() + p + 1;
            if (!i_set[p].contains(pre)) {
                i_set[p].push(pre);
                chain_.tables_valid = false;
            }
            if (!j_set[p].contains(suf)) {
                j_set[p].push(suf);
                chain_.tables_valid = false;
            }
        }
    }
    invalidate_site_tensors();
}

IndexSet Tci2::kronecker_i(size_t p) const // tensorci2.rs:1224-1234
{
    IndexSet r;
    r.width = p + 1;
    const IndexSet& src = i_set[p];
    const size_t d = local_dims[p];
    r.count = src.count * d;
    r.d.resize(r.count * r.width);
    uint32_t* out = r.d.data();
    for (size_t k = 0; k < src.count; ++k) {
        const uint32_t* in = src.at(k);
        for (size_t li = 0; li < d; ++li) {
            if (src.width) std::memcpy(out, in, src.width * sizeof(uint32_t));
            out[src.width] = (uint32_t)li;
            out += r.width;
        }
    }
    return r;
}

IndexSet Tci2::kronecker_j(size_t p) const // tensorci2.rs:1236-1246
{
    IndexSet r;
    r.width = n_ - p;
    const IndexSet& src = j_set[p];
    const size_t d = local_dims[p];
    r.count = src.count * d;
    r.d.resize(r.count * r.width);
    uint32_t* out = r.d.data();
    for (size_t li = 0; li < d; ++li)
        for (size_t k = 0; k < src.count; ++k) {
            out[0] = (uint32_t)li;
            if (src.width) std::memcpy(out + 1, src.at(k), src.width * sizeof(uint32_t));
            out += r.width;
        }
    return r;
}

// order-preserving union (tensorci2.rs:1837-1846); hashing replaces the reference's O(M*chi) `contains`
void Tci2::union_extras(IndexSet& base, const IndexSet& extras)
{
    if (extras.count == 0) return;
    if (base.width == 0) {
        if (base.count == 0) base.count = 1;
        return;
    }
    // open-addressing table of (hash, index+1); digits are compared on a hash hit, so the result is exactly the
    // reference's `contains`-based union
    const size_t w = base.width;
    auto hash_of = [w](const uint32_t* v) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (size_t i = 0; i < w; ++i) {
            h ^= v[i];
            h *= 0x100000001b3ull;
        }
        return h ^ (h >> 29);
    };
    size_t cap = 16;
    while (cap < 2 * (base.count + extras.count)) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0);
    base.d.reserve((base.count + extras.count) * w);
    auto insert = [&](size_t idx) {
        size_t pos = hash_of(base.at(idx)) & (cap - 1);
        while (slot[pos]) pos = (pos + 1) & (cap - 1);
        slot[pos] = (uint32_t)idx + 1;
    };
    for (size_t k = 0; k < base.count; ++k) insert(k);
    for (size_t k = 0; k < extras.count; ++k) {
        const uint32_t* e = extras.at(k);
        size_t pos = hash_of(e) & (cap - 1);
        bool found = false;
        while (slot[pos]) {
            if (std::memcmp(base.at(slot[pos] - 1), e, w * sizeof(uint32_t)) == 0) {
                found = true;
                break;
            }
            pos = (pos + 1) & (cap - 1);
        }
        if (!found) {
            base.push(e);
            slot[pos] = (uint32_t)base.count; // index of the new entry + 1
        }
    }
}

void Tci2::accumulate(const IndexSet& set, size_t first_site, std::vector<uint64_t>& acc) const
{
    const int K = fn_dev_.n_acc;
    acc.assign(set.count * (size_t)K, 0);
    for (size_t e = 0; e < set.count; ++e) {
        const uint32_t* v = set.at(e);
        for (int k = 0; k < K; ++k) {
            uint64_t a = 0;
            const uint64_t* w = weights_.data() + (size_t)k * total_;
            for (size_t s = 0; s < set.width; ++s) a += w[offset_[first_site + s] + v[s]];
            acc[e * K + k] = a;
        }
    }
}

// Accumulators of both index sets -> pinned arena -> device (one asynchronous copy on the main stream).
void Tci2::stage_accumulators(const IndexSet& a, size_t a0, const IndexSet& b, size_t b0, const std::vector<uint64_t>* acc_a,
                              const std::vector<uint64_t>* acc_b, const uint64_t** d_ra, const uint64_t** d_rb, bool in_place)
{
    hipStream_t st = eng.stream();
    const size_t K = (size_t)fn_dev_.n_acc;
    std::vector<uint64_t> ra_own, rb_own;
    if (!acc_a) accumulate(a, a0, ra_own);
    if (!acc_b) accumulate(b, b0, rb_own);
    const std::vector<uint64_t>& ra = acc_a ? *acc_a : ra_own;
    const std::vector<uint64_t>& rb = acc_b ? *acc_b : rb_own;
    if (ra.size() != a.count * K || rb.size() != b.count * K)
        throw Error(T4A_GPU_INTERNAL_ERROR, "prepared accumulators have the wrong size");
    // pinned staging arena: entries stay valid until the next stream sync
    const size_t need = ra.size() + rb.size();
    if (acc_used_ + need > h_acc_.cap) {
        T4A_HIP(hipStreamSynchronize(st));
        acc_used_ = 0;
        h_acc_.reserve(std::max(need * 2, (size_t)1 << 16));
    }
    uint64_t* ha = h_acc_.get() + acc_used_;
    acc_used_ += need;
    std::memcpy(ha, ra.data(), ra.size() * sizeof(uint64_t));
    std::memcpy(ha + ra.size(), rb.data(), rb.size() * sizeof(uint64_t));
    if (in_place) { // pinned memory is device-visible under the same address
        *d_ra = ha;
        *d_rb = ha + ra.size();
        return;
    }
    d_rowacc_.reserve(need);
    static const bool dma_copy = diag_env("T4A_ACC_DMA") != nullptr;
    if (dma_copy) {
        T4A_HIP(hipMemcpyAsync(d_rowacc_.get(), ha, need * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    } else {
        stage_copy_launch(ha, d_rowacc_.get(), need, st);
        T4A_HIP(hipGetLastError());
    }
    *d_ra = d_rowacc_.get();
    *d_rb = d_rowacc_.get() + ra.size();
}

// out[ia + a.count*ib] = f(index with a's digits at sites [a0, a0+a.width) and b's at [b0, b0+b.width))
void Tci2::eval_matrix(const IndexSet& a, size_t a0, const IndexSet& b, size_t b0, double* d_out,
                       unsigned long long* d_maxbits, const std::vector<uint64_t>* acc_a,
                       const std::vector<uint64_t>* acc_b)
{
    require_fn();
    const size_t na = a.count, nb = b.count;
    if (na == 0 || nb == 0) return;
    if (a.width + b.width != n_) throw Error(T4A_GPU_INTERNAL_ERROR, "eval_matrix: index widths do not cover all sites");
    hipStream_t st = eng.stream();
    eng.prof.v[11] += (double)na * (double)nb;
    if (fn_kind_ == FnKind::Builtin) {
        const uint64_t *d_ra = nullptr, *d_rb = nullptr;
        stage_accumulators(a, a0, b, b0, acc_a, acc_b, &d_ra, &d_rb);
        pi_eval_launch(fn_dev_, d_ra, (int)na, d_rb, (int)nb, d_out, (int)na, false, d_maxbits, st);
        T4A_HIP(hipGetLastError());
    } else {
        // host batch callback: points in row-major order of (ia, ib) — `ia` outer, `ib` inner — exactly the
        // order the reference hands to batched_f (tensorci2.rs:1862-1869)
        const size_t npts = na * nb;
        cb_vals_.reserve(npts); // (pinned, grow-only: the previous upload out of it was synchronised below)
        double* const vals = cb_vals_.get();
        if (pi_shard.active()) {
            // column blocks over the ranks of the process group + one all-gather (pishard.hpp)
            pi_shard_evaluate(pi_shard, cb_, cb_ctx_, n_, a.d.data(), a.width, a0, na, b.d.data(), b.width, b0, nb, vals);
        } else {
            // the index buffer of the callback: grow-only and filled by a few host threads (round 5: measured on the native callback of
            // tools/bench_components.py — the backend's share of a cfg3 sweep through a callback was 0.27 s, most of it this buffer)
            if (cb_idx_cap_ < npts * n_) {
                cb_idx_cap_ = npts * n_ + npts * n_ / 4;
                cb_idx_.reset(new uint32_t[cb_idx_cap_]);
            }
            uint32_t* const idx = cb_idx_.get();
            auto fill_rows = [&](size_t ia0, size_t ia1) {
                for (size_t ia = ia0; ia < ia1; ++ia)
                    for (size_t ib = 0; ib < nb; ++ib) {
                        uint32_t* dst = idx + (ia * nb + ib) * n_;
                        std::memcpy(dst + a0, a.at(ia), a.width * sizeof(uint32_t));
                        std::memcpy(dst + b0, b.at(ib), b.width * sizeof(uint32_t));
                    }
            };
            const unsigned hw = std::thread::hardware_concurrency();
            const size_t nthr = (npts * n_ < ((size_t)1 << 21) || hw < 2) ? 1 : std::min<size_t>(std::min<size_t>(8, hw), na);
            if (nthr <= 1) {
                fill_rows(0, na);
            } else {
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nthr; ++t) pool.emplace_back(fill_rows, na * t / nthr, na * (t + 1) / nthr);
                fill_rows(0, na / nthr);
                for (auto& th : pool) th.join();
            }
            // opt-in (t4a_gpu_tci2_set_callback_threads): the point list of ONE candidate matrix split over host threads that call the
            // user function concurrently — outside the reference's contract (it calls f sequentially on one thread on purpose,
            // docs/design/adaptive-tci-interpolation.md:9-11), so only for callbacks that are thread safe.  Same points, same values.
            const size_t cthr = (callback_threads <= 1 || npts < ((size_t)1 << 14)) ? 1 : std::min<size_t>(callback_threads, npts >> 12);
            if (cthr <= 1) {
                const int64_t got = cb_(cb_ctx_, idx, n_, npts, vals);
                if (got < 0 || (size_t)got != npts)
                    throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned " + std::to_string(got) + " values for " +
                                                            std::to_string(npts) + " requested entries");
            } else {
                std::vector<int64_t> got(cthr, 0);
                auto call = [&](size_t t) {
                    const size_t p0 = npts * t / cthr, p1 = npts * (t + 1) / cthr;
                    got[t] = cb_(cb_ctx_, idx + p0 * n_, n_, p1 - p0, vals + p0) - (int64_t)(p1 - p0);
                };
                std::vector<std::thread> pool;
                for (size_t t = 1; t < cthr; ++t) pool.emplace_back(call, t);
                call(0);
                for (auto& th : pool) th.join();
                for (size_t t = 0; t < cthr; ++t)
                    if (got[t] != 0)
                        throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned a wrong number of values for block " + std::to_string(t) + " of " +
                                                                std::to_string(cthr) + " (callback threads)");
            }
        }
        d_vals_.reserve(npts);
        T4A_HIP(hipMemcpyAsync(d_vals_.get(), vals, npts * sizeof(double), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st)); // (the pinned buffer is reused by the next matrix)
        // vals is (na x nb) row-major == (nb x na) column-major; transpose into the column-major na x nb output
        transpose_launch(d_vals_.get(), (int)nb, (int)na, (int)nb, d_out, (int)na, st);
        if (d_maxbits) absmax_launch(d_out, npts, d_maxbits, st);
        T4A_HIP(hipGetLastError());
    }
}

std::vector<double> Tci2::eval_points_host(const std::vector<uint32_t>& idx, size_t n_pts)
{
    require_fn();
    std::vector<double> out(n_pts);
    if (n_pts == 0) return out;
    if (fn_kind_ == FnKind::Builtin) {
        for (size_t p = 0; p < n_pts; ++p) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < fn_dev_.n_acc; ++k) {
                const uint64_t* w = weights_.data() + (size_t)k * total_;
                for (size_t s = 0; s < n_; ++s) acc[k] += w[offset_[s] + idx[p * n_ + s]];
            }
            out[p] = t4a_fn_value(fn_dev_.fid, acc, fn_dev_.params);
        }
    } else {
        const int64_t got = cb_(cb_ctx_, idx.data(), n_, n_pts, out.data());
        if (got < 0 || (size_t)got != n_pts)
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned " + std::to_string(got) + " values for " +
                                                    std::to_string(n_pts) + " requested entries");
    }
    return out;
}

// kron(I_b, d_b) + extras (rows) or kron(d_{b+1}, J_{b+1}) + extras (columns) of bond `bond`, with accumulators
void Tci2::build_side(size_t bond, bool cols, const IndexSet& extra, SidePrep& out) const
{
    out.valid = false;
    out.bond = bond;
    out.cols = cols;
    out.set = cols ? kronecker_j(bond + 1) : kronecker_i(bond);
    union_extras(out.set, extra);
    out.acc.clear();
    if (fn_kind_ == FnKind::Builtin) accumulate(out.set, cols ? bond + 1 : 0, out.acc);
    out.valid = true;
}

void Tci2::invalidate_fill_cache()
{
    for (auto& f : fill_cache_) f.valid = false;
}

// Accumulators fill_site_tensors needs for site b (tensorci2.rs:1101-1145): J_b, kron(I_b, d_b) and I_{b+1}.
void Tci2::prepare_fill_site(size_t b)
{
    if (fn_kind_ != FnKind::Builtin || b >= n_) return;
    if (chain_.digits_stale) { // after a bond chain the accumulators are in the pinned mirror of the device tables: no digits needed
        prepare_fill_site_from_mirror(b);
        return;
    }
    if (fill_cache_.size() != n_) fill_cache_.assign(n_, FillAcc());
    if (shard_world > 1 && (b % shard_world) != shard_rank) return;
    FillAcc& f = fill_cache_[b];
    f.valid = false;
    if (i_set[b].count == 0 || j_set[b].count == 0) return;
    accumulate(j_set[b], b + 1, f.accJ);
    accumulate(kronecker_i(b), 0, f.accK);
    if (b + 1 < n_)
        accumulate(i_set[b + 1], 0, f.accI);
    else
        f.accI.clear();
    f.valid = true;
}

LuciResult Tci2::luci_on_sets(const IndexSet& is, const IndexSet& js, const RrLUOptions& o, bool need_factors,
                              const std::vector<uint64_t>* acc_rows, const std::vector<uint64_t>* acc_cols)
{
    const size_t M = is.count, N = js.count;
    hipStream_t st = eng.stream();
    static const bool no_fuse = diag_env("T4A_NO_FUSED_PI") != nullptr;
    if (fn_kind_ == FnKind::Builtin && !no_fuse && M > 0 && N > 0) {
        // built-in functor: only the accumulators travel; the rrLU kernel evaluates Pi into its registers
        FusedPi fp;
        fp.fn = fn_dev_;
        // small bonds (the single-workgroup plan with the fused candidate-matrix build takes them): the kernel reads the few
        // hundred bytes of accumulators straight from the pinned arena — no staging kernel in front of a 10 us factorisation
        static const bool no_in_place = diag_env("T4A_ACC_STAGE_ALWAYS") != nullptr;
        fp.host_resident = !no_in_place && (M * N <= (size_t)64 * 64);
        stage_accumulators(is, 0, js, is.width, acc_rows, acc_cols, &fp.d_rowacc, &fp.d_colacc, fp.host_resident);
        eng.prof.v[11] += (double)M * (double)N;
        LuciResult lu = eng.luci(nullptr, (int)M, (int)N, o, need_factors, false, &fp);
        acc_used_ = 0;
        if (lu.abs_max > max_sample_value) max_sample_value = lu.abs_max;
        return lu;
    }
    double* d_pi = eng.pi(M * N);
    static const bool host_prof = std::getenv("T4A_HOST_PROFILE") != nullptr;
    static double hp_eval = 0;
    static long hp_n = 0;
    const auto hp_t0 = std::chrono::steady_clock::now();
    if (eng.prof.enabled) T4A_HIP(hipEventRecord(ev_pi_.a, st));
    eval_matrix(is, 0, js, is.width, d_pi, nullptr, acc_rows, acc_cols);
    if (eng.prof.enabled) T4A_HIP(hipEventRecord(ev_pi_.b, st));
    if (host_prof) {
        hp_eval += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hp_t0).count();
        if (++hp_n % 580 == 0) std::fprintf(stderr, "[host profile] eval_matrix host part %.1f us per bond\n", 1e3 * hp_eval / hp_n);
    }
    LuciResult lu = eng.luci(d_pi, (int)M, (int)N, o, need_factors, false);
    acc_used_ = 0; // eng.luci synchronised the stream
    if (eng.prof.enabled) {
        float ms = 0.f;
        T4A_HIP(hipEventElapsedTime(&ms, ev_pi_.a, ev_pi_.b));
        eng.prof.v[2] += ms;
        eng.prof.v[3] += 1.0;
    }
    // update_max_sample_value over every entry of Π (tensorci2.rs:2009-2014)
    if (lu.abs_max > max_sample_value) max_sample_value = lu.abs_max;
    return lu;
}

// PivotSearchStrategy::Rook (tensorci2.rs:1904-1929): the candidate matrix is never materialised; the lazy
// block-rook kernel asks for full rows / columns of Pi, which are evaluated on demand (LazyPiEvaluator, :2035-2142).
LuciResult Tci2::rook_on_sets(const IndexSet& is, const IndexSet& js, const RrLUOptions& o)
{
    require_fn();
    const size_t M = is.count, N = js.count;
    hipStream_t st = eng.stream();
    RookSource src;
    src.M = (int)M;
    src.N = (int)N;
    if (fn_kind_ == FnKind::Builtin) {
        const size_t K = (size_t)fn_dev_.n_acc;
        std::vector<uint64_t> ra, rb;
        accumulate(is, 0, ra);
        accumulate(js, is.width, rb);
        d_rowacc_.reserve(ra.size() + rb.size());
        // (one copy out of pinned memory: two copies out of pageable vectors and a synchronisation cost ~60 us per bond on small problems;
        // the staging buffer is free again: every rook search ends with a stream synchronisation)
        h_rookacc_.reserve(ra.size() + rb.size());
        std::memcpy(h_rookacc_.get(), ra.data(), ra.size() * sizeof(uint64_t));
        std::memcpy(h_rookacc_.get() + ra.size(), rb.data(), rb.size() * sizeof(uint64_t));
        T4A_HIP(hipMemcpyAsync(d_rowacc_.get(), h_rookacc_.get(), (ra.size() + rb.size()) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        const uint64_t* d_ra = d_rowacc_.get();
        const uint64_t* d_rb = d_rowacc_.get() + ra.size();
        src.column = [=](int c, double* out) {
            pi_eval_launch(fn_dev_, d_ra, (int)M, d_rb + (size_t)c * K, 1, out, (int)M, false, nullptr, st);
        };
        src.row = [=](int r, double* out) {
            pi_eval_launch(fn_dev_, d_ra + (size_t)r * K, 1, d_rb, (int)N, out, 1, false, nullptr, st);
        };
        // (a built-in functor costs nothing to evaluate: the search runs device-resident on the materialised matrix, rook.hip)
        static const bool host_driven = std::getenv("T4A_ROOK_HOST") != nullptr; // (A/B and parity of the two search drivers)
        if (!host_driven)
            src.full = [=](double* out) { pi_eval_launch(fn_dev_, d_ra, (int)M, d_rb, (int)N, out, (int)M, false, nullptr, st); };
    } else {
        auto eval_points = [this, st, &is, &js](size_t r0, size_t nr, size_t c0, size_t nc, double* out) {
            const size_t npts = nr * nc;
            std::vector<uint32_t> idx(npts * n_);
            for (size_t a = 0; a < nr; ++a)
                for (size_t b = 0; b < nc; ++b) {
                    uint32_t* dst = idx.data() + (a * nc + b) * n_;
                    std::memcpy(dst, is.at(r0 + a), is.width * sizeof(uint32_t));
                    std::memcpy(dst + is.width, js.at(c0 + b), js.width * sizeof(uint32_t));
                }
            std::vector<double> vals(npts);
            const int64_t got = cb_(cb_ctx_, idx.data(), n_, npts, vals.data());
            if (got < 0 || (size_t)got != npts)
                throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned " + std::to_string(got) + " values for " +
                                                        std::to_string(npts) + " requested entries");
            T4A_HIP(hipMemcpyAsync(out, vals.data(), npts * sizeof(double), hipMemcpyHostToDevice, st));
            T4A_HIP(hipStreamSynchronize(st));
        };
        src.column = [=](int c, double* out) { eval_points(0, M, (size_t)c, 1, out); };
        src.row = [=](int r, double* out) { eval_points((size_t)r, 1, 0, N, out); };
    }
    double sampled = max_sample_value;
    LuciResult lu = rook_luci(eng, rook_work_, src, o, &sampled, &eng.prof.v[11]);
    acc_used_ = 0;
    max_sample_value = sampled; // tci.max_sample_value = evaluator.sampled_max() (tensorci2.rs:1926)
    return lu;
}

void Tci2::set_core_zero(size_t site, size_t l, size_t s, size_t r)
{
    DevCore& c = cores[site];
    c.buf.reserve(std::max<size_t>(l * s * r, 1));
    c.l = l;
    c.s = s;
    c.r = r;
    fill_launch(c.buf.get(), l * s * r, 0.0, eng.stream());
}

void Tci2::set_core_from_left(size_t site, size_t left_dim, size_t site_dim, const LuciResult& lu)
{
    const size_t nb = std::max<size_t>((size_t)lu.rank, 1);
    DevCore& c = cores[site];
    c.buf.reserve(left_dim * site_dim * nb);
    c.l = left_dim;
    c.s = site_dim;
    c.r = nb;
    const size_t total = c.size();
    hipLaunchKernelGGL(pack_left_core_kernel, dim3(blocks_for(total)), dim3(256), 0, eng.stream(), eng.left(), lu.M, lu.M,
                       lu.rank, c.buf.get(), (int)left_dim, (int)site_dim, (int)nb);
}

void Tci2::set_core_from_right(size_t site, size_t site_dim, size_t right_dim, const LuciResult& lu)
{
    const size_t nb = std::max<size_t>((size_t)lu.rank, 1);
    DevCore& c = cores[site];
    c.buf.reserve(nb * site_dim * right_dim);
    c.l = nb;
    c.s = site_dim;
    c.r = right_dim;
    const size_t total = c.size();
    const int ldm = lu.rank > 0 ? lu.rank : 1;
    hipLaunchKernelGGL(pack_right_core_kernel, dim3(blocks_for(total)), dim3(256), 0, eng.stream(), eng.right(), ldm,
                       lu.rank, lu.N, c.buf.get(), (int)nb, (int)site_dim, (int)right_dim);
}

void Tci2::update_pivot_errors(const std::vector<double>& e) // tensorci2.rs:801-808
{
    if (pivot_errors.size() < e.size()) pivot_errors.resize(e.size(), 0.0);
    for (size_t i = 0; i < e.size(); ++i) pivot_errors[i] = std::fmax(pivot_errors[i], e[i]);
}

// tensorci2.rs:1821-2007
void Tci2::update_pivots(size_t b, bool left_orthogonal, const TCI2Options& options, const IndexSet& extra_i,
                         const IndexSet& extra_j)
{
    static const bool host_prof = std::getenv("T4A_HOST_PROFILE") != nullptr;
    static double hp_sets = 0, hp_luci = 0, hp_post = 0;
    static long hp_n = 0;
    const auto hp_t0 = std::chrono::steady_clock::now();
    sync_digits();
    // one side may have been built while the previous bond's kernels were running (it does not depend on them)
    SidePrep ready;
    if (prep_.valid && prep_.bond == b) std::swap(ready, prep_);
    prep_.valid = false;
    const bool have_rows = ready.valid && !ready.cols, have_cols = ready.valid && ready.cols;
    IndexSet i_own, j_own;
    if (!have_rows) {
        i_own = kronecker_i(b);
        union_extras(i_own, extra_i);
    }
    if (!have_cols) {
        j_own = kronecker_j(b + 1);
        union_extras(j_own, extra_j);
    }
    const IndexSet& i_comb = have_rows ? ready.set : i_own;
    const IndexSet& j_comb = have_cols ? ready.set : j_own;
    const bool acc_ready = ready.valid && fn_kind_ == FnKind::Builtin;
    if (i_comb.count == 0 || j_comb.count == 0) return;
    if (prefetch_.wanted || prefetch_.fill_site >= 0 || prefetch_.flush_fill) {
        // host work that does not depend on this bond: the independent side of the NEXT bond and the fill
        // accumulators of a site that is already final; runs while this bond's kernels are in flight
        const Prefetch pf = prefetch_;
        prefetch_.wanted = false;
        prefetch_.fill_site = -1;
        prefetch_.flush_fill = false;
        eng.overlap_hook = [this, pf]() {
            if (pf.wanted) build_side(pf.bond, pf.cols, *pf.extra, prep_);
            if (pf.fill_site >= 0) prepare_fill_site((size_t)pf.fill_site);
            if (pf.flush_fill) flush_deferred_fill(); // the previous half-sweep's fill: its stream operations go out now
        };
    }
    const auto hp_t1 = std::chrono::steady_clock::now();

    const bool extras_used = extra_i.count != 0 || extra_j.count != 0;
    RrLUOptions lo;
    lo.max_bond_dim = options.max_bond_dim_or_max();
    lo.rel_tol = options.tolerance;
    lo.abs_tol = 0.0;
    lo.left_orthogonal = left_orthogonal;
    // the reference always builds the factors; they are only CONSUMED when no extras were merged
    // (tensorci2.rs:1942-1949), so the device skips the trsm/gemm otherwise.
    LuciResult lu = options.pivot_search == 0
                        ? luci_on_sets(i_comb, j_comb, lo, !extras_used, (acc_ready && have_rows) ? &ready.acc : nullptr,
                                       (acc_ready && have_cols) ? &ready.acc : nullptr)
                        : rook_on_sets(i_comb, j_comb, lo);
    if (eng.overlap_hook) { // not consumed (rook path / empty matrix): run it now so the next bond still finds it
        std::function<void()> hook;
        hook.swap(eng.overlap_hook);
        hook();
    }
    const auto hp_t2 = std::chrono::steady_clock::now();
    if (b < last_sweep_shapes.size()) last_sweep_shapes[b] = {i_comb.count, j_comb.count, (size_t)lu.rank};

    const std::vector<size_t> rows = non_empty_or_first(lu.row_perm, lu.rank);
    const std::vector<size_t> cols = non_empty_or_first(lu.col_perm, lu.rank);
    IndexSet ni, nj;
    ni.width = i_comb.width;
    nj.width = j_comb.width;
    for (size_t r : rows) ni.push(i_comb.at(r));
    for (size_t c : cols) nj.push(j_comb.at(c));
    i_set[b + 1] = ni;
    j_set[b] = nj;
    chain_.tables_valid = false; // (the host path moved the sets: the device tables of the bond chain are stale)
    if (host_prof) {
        const auto hp_t3 = std::chrono::steady_clock::now();
        hp_sets += std::chrono::duration<double, std::milli>(hp_t1 - hp_t0).count();
        hp_luci += std::chrono::duration<double, std::milli>(hp_t2 - hp_t1).count();
        hp_post += std::chrono::duration<double, std::milli>(hp_t3 - hp_t2).count();
        if (++hp_n % 580 == 0)
            std::fprintf(stderr, "[host profile] per bond: build sets %.1f us, luci_on_sets (host+device) %.1f us, post %.1f us\n",
                         1e3 * hp_sets / hp_n, 1e3 * hp_luci / hp_n, 1e3 * hp_post / hp_n);
    }

    if (extras_used) {
        if (!lu.pivot_errors.empty()) bond_errors[b] = lu.pivot_errors.back();
        return;
    }
    const size_t left_dim = (b == 0) ? 1 : i_set[b].count;
    set_core_from_left(b, left_dim, local_dims[b], lu);
    const size_t right_dim = (b + 1 == n_ - 1) ? 1 : j_set[b + 1].count;
    set_core_from_right(b + 1, local_dims[b + 1], right_dim, lu);
    if (!lu.pivot_errors.empty()) bond_errors[b] = lu.pivot_errors.back();
}

// tensorci2.rs:746-798
void Tci2::sweep2site(bool forward, const TCI2Options& options)
{
    options.validate();
    require_fn();
    fill_wait();
    invalidate_site_tensors();
    flush_pivot_errors();
    last_sweep_shapes.assign(n_ - 1, {0, 0, 0});
    prep_.valid = false;
    prefetch_.wanted = false;
    IndexSet ei, ej; // empty extras
    if (chain_enqueue(forward, options, -1, false)) { // built-in functor: the whole half-sweep in one go (tci2_chain.hip)
        chain_finish(options);
        fill_site_tensors();
        return;
    }
    sync_digits();
    if (forward) {
        for (size_t b = 0; b + 1 < n_; ++b) {
            ei.width = b + 1;
            ej.width = n_ - b - 1;
            update_pivots(b, true, options, ei, ej);
        }
    } else {
        for (size_t b = n_ - 1; b-- > 0;) {
            ei.width = b + 1;
            ej.width = n_ - b - 1;
            update_pivots(b, false, options, ei, ej);
        }
    }
    fill_site_tensors();
}

// tensorci2.rs:918-1050
void Tci2::sweep1site_at_bond(size_t b, bool forward, double rel_tol, double abs_tol, size_t max_bond_dim,
                              bool update_tensors)
{
    sync_digits();
    IndexSet is = forward ? kronecker_i(b) : i_set[b];
    IndexSet js = forward ? j_set[b] : kronecker_j(b);
    if (is.count == 0 || js.count == 0) return;
    RrLUOptions lo;
    lo.max_bond_dim = max_bond_dim;
    lo.rel_tol = rel_tol;
    lo.abs_tol = abs_tol;
    lo.left_orthogonal = forward;
    LuciResult lu = luci_on_sets(is, js, lo, update_tensors);
    const std::vector<size_t> rows = non_empty_or_first(lu.row_perm, lu.rank);
    const std::vector<size_t> cols = non_empty_or_first(lu.col_perm, lu.rank);
    IndexSet ni, nj;
    ni.width = is.width;
    nj.width = js.width;
    for (size_t r : rows) ni.push(is.at(r));
    for (size_t c : cols) nj.push(js.at(c));
    if (forward) {
        i_set[b + 1] = ni;
        j_set[b] = nj;
    } else {
        i_set[b] = ni;
        j_set[b - 1] = nj;
    }
    chain_.tables_valid = false;
    if (update_tensors) {
        if (forward) {
            const size_t left_dim = (b == 0) ? 1 : i_set[b].count;
            set_core_from_left(b, left_dim, local_dims[b], lu);
        } else {
            const size_t right_dim = (b == n_ - 1) ? 1 : j_set[b].count;
            set_core_from_right(b, local_dims[b], right_dim, lu);
        }
    }
    if (!lu.pivot_errors.empty()) {
        const size_t bond_idx = forward ? b : b - 1;
        bond_errors[bond_idx] = lu.pivot_errors.back();
    }
    update_pivot_errors(lu.pivot_errors);
}

// tensorci2.rs:865-915
void Tci2::sweep1site(bool forward, double rel_tol, double abs_tol, size_t max_bond_dim, bool update_tensors)
{
    if (!(rel_tol >= 0.0) || !std::isfinite(rel_tol)) throw Error(T4A_GPU_INVALID_ARGUMENT, "rel_tol must be finite and non-negative");
    if (!(abs_tol >= 0.0) || !std::isfinite(abs_tol)) throw Error(T4A_GPU_INVALID_ARGUMENT, "abs_tol must be finite and non-negative");
    if (max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be positive");
    require_fn();
    fill_wait();
    flush_pivot_errors();
    invalidate_site_tensors();
    // built-in functor: the whole sweep as one chain on the device (tci2_chain.hip; the independent side of every bond is the
    // table itself), the site tensors from the factored matrices it leaves behind
    bool chained = false;
    static const bool no_chain1 = std::getenv("T4A_NO_CHAIN_1SITE") != nullptr;
    if (!no_chain1 && n_ >= 2) {
        TCI2Options o1;
        o1.tolerance = rel_tol;
        o1.max_bond_dim = max_bond_dim == std::numeric_limits<size_t>::max() ? 0 : max_bond_dim;
        o1.pivot_search = 0;
        o1.nsearch = 0;
        prep_.valid = false;
        prefetch_.wanted = false;
        chain_.one_site = true;
        chain_.one_factors = update_tensors;
        chain_.abs_tol = abs_tol;
        try {
            chained = chain_enqueue(forward, o1, -1, false);
            if (chained) chain_finish(o1);
        } catch (...) {
            chain_abort(); // (nothing may stay in flight, no XCD reserved)
            chain_.one_site = false;
            throw;
        }
        chain_.one_site = false;
    }
    if (chained) {
        // (every bond went through the chain or through its per-bond tail)
    } else if (forward) {
        for (size_t b = 0; b + 1 < n_; ++b) sweep1site_at_bond(b, true, rel_tol, abs_tol, max_bond_dim, update_tensors);
    } else {
        for (size_t b = n_ - 1; b >= 1; --b) sweep1site_at_bond(b, false, rel_tol, abs_tol, max_bond_dim, update_tensors);
    }
    if (update_tensors) { // tensorci2.rs:902-912 + fill_tensor :813-850
        const size_t last = forward ? n_ - 1 : 0;
        const bool on_device = chained && forward && chain_.last_core_ok; // (evaluated from the tables behind the chain: chain_last_core_kernel)
        chain_.last_core_ok = false;
        const size_t A = i_set[last].count, S = local_dims[last], C = j_set[last].count; // (counts are current behind a chain, digits need not be)
        DevCore& c = cores[last];
        if (!on_device) c.buf.reserve(std::max<size_t>(A * S * C, 1));
        c.l = A;
        c.s = S;
        c.r = C;
        if (A * S * C > 0 && !on_device) {
            sync_digits();
            IndexSet rows = kronecker_i(last);
            const IndexSet& jl = j_set[last];
            double* d_m = eng.pi(A * S * C);
            eval_matrix(rows, 0, jl, rows.width, d_m, nullptr);
            hipLaunchKernelGGL(pack_left_core_kernel, dim3(blocks_for(c.size())), dim3(256), 0, eng.stream(), d_m,
                               (int)(A * S), (int)(A * S), (int)C, c.buf.get(), (int)A, (int)S, (int)C);
            T4A_HIP(hipStreamSynchronize(eng.stream()));
            acc_used_ = 0;
        }
    }
}

// tensorci2.rs:1201-1221
void Tci2::make_canonical(double rel_tol, double abs_tol, size_t max_bond_dim)
{
    if (!(rel_tol >= 0.0) || !std::isfinite(rel_tol)) throw Error(T4A_GPU_INVALID_ARGUMENT, "rel_tol must be finite and non-negative");
    if (!(abs_tol >= 0.0) || !std::isfinite(abs_tol)) throw Error(T4A_GPU_INVALID_ARGUMENT, "abs_tol must be finite and non-negative");
    if (max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be positive");
    sweep1site(true, 0.0, 0.0, std::numeric_limits<size_t>::max(), false);
    sweep1site(false, rel_tol, abs_tol, max_bond_dim, false);
    sweep1site(true, rel_tol, abs_tol, max_bond_dim, true);
}

// tensorci2.rs:1065-1186 — all sites are independent given the final I/J sets, so the evaluations, the
// partial-pivot LU factorisations and the triangular solves of every site are issued as batches.
// The whole fill runs on its own stream.  With `async` (optimize loop, nsearch == 0, built-in functor) the host
// does not wait: the next half-sweep's bond updates only need the index sets, so the fill overlaps with them
// (the rrLU chain leaves most CUs idle).  Errors (singular pivot matrix) surface at the next fill_wait().
void Tci2::fill_site_tensors() { fill_site_tensors_impl(false); }

// Issues the stream operations of one fill.  Fills of consecutive sweeps at saturated rank are operation-for-operation
// identical (same device addresses, shapes and pinned staging buffer), so the sequence is captured into a HIP graph the
// second time a signature is seen and replayed afterwards: one submission instead of ~25, which also keeps the
// runtime's submission path free for the latency-critical bond updates on the main stream.
void Tci2::issue_fill_ops(std::vector<std::function<void()>>& ops, const std::vector<uint64_t>& sig)
{
    hipStream_t st = fill_stream_;
    static const bool use_graph = std::getenv("T4A_NO_FILL_GRAPH") == nullptr;
    if (fill_timed_) T4A_HIP(hipEventRecord(ev_fill_.a, st));
    for (auto& f : fill_pre_ops_) f(); // (diagnosis switch T4A_FILL_GRAPH_NO_COPY: the upload in front of the graph)
    fill_pre_ops_.clear();
    static const bool dev_sync_first = diag_env("T4A_FILL_GRAPH_DEVSYNC") != nullptr; // (diagnosis)
    if (dev_sync_first) T4A_HIP(hipDeviceSynchronize());
    bool done = false;
    // (no graph replay on a handle whose cores are exported / imported through the LEGACY DEFAULT STREAM.  Round 4 saw GPU memory
    // faults in 25 - 75 % of `bench.py --mode site-shard` runs with replay on; round 5 bisected them (tools/r5_gpu_shardfault*.sh,
    // profiles/r05_fill_graph_fault_bisect.txt, five runs per arm): they need the replay AND an event protocol that runs through
    // stream 0 — torch's current stream in a process that never selected another — in BOTH directions (export: stream 0 waits for an
    // event recorded behind the replay; import: the import stream waits for an event recorded on stream 0).  Replacing either event
    // by a host wait, moving torch to a side stream, or a device-wide synchronisation in front of the replay: 0 faults of 5; the
    // pool, the bond chain, kernel serialisation, the order of the export copies and the copy nodes of the graph: no influence.
    // Every pointer the graph holds is part of the signature (checked again: tables, staging buffers, tickets, flags), so this is
    // the runtime's implicit ordering of stream 0 against the streams a graph launch runs on, not a stale pointer.  parallel.py and
    // bench.py therefore exchange on a side stream; a caller that hands in stream 0 gets direct issue, which is what the
    // round-4 stop-gap did for every shared handle.)
    // the root cause is NOT identified (the library's streams are non-blocking, so "implicit ordering with stream 0"
    // does not explain the faults, and no multi-GPU soak of the patch-farm path exists).  The default is therefore the round-4 guard —
    // no replay on ANY handle whose site tensors are exported / imported asynchronously; the relaxed guard (replay unless the legacy
    // stream or a blocking stream took part) is an opt-in: t4a_gpu_tci2_set_chain bit 4.  The gain it buys is ~0.4 % of a sweep.
    static const bool graph_shared = diag_env("T4A_FILL_GRAPH_SHARED") != nullptr; // (diagnosis: replay even then)
    const bool shared_ok = !cores_shared_async_ || (fill_graph_relaxed && !cores_shared_legacy_stream_) || graph_shared;
    if (use_graph && !fill_graph_broken_ && shared_ok) {
        if (fill_graph_exec_ && sig == fill_graph_sig_) {
            T4A_HIP(hipGraphLaunch(fill_graph_exec_, st));
            ++fill_stats_[1];
            done = true;
        } else if (sig == fill_last_sig_) { // second time in a row: worth capturing
            if (fill_graph_exec_) {
                (void)hipGraphExecDestroy(fill_graph_exec_);
                fill_graph_exec_ = nullptr;
            }
            hipGraph_t graph = nullptr;
            pool::capture_begin(); // (no device-wide synchronisation of another handle's thread may fall into the capture)
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                try {
                    for (auto& f : ops) f();
                } catch (...) {
                    ok = false;
                }
                if (hipStreamEndCapture(st, &graph) != hipSuccess || !graph) ok = false;
            }
            pool::capture_end();
            if (ok && hipGraphInstantiate(&fill_graph_exec_, graph, nullptr, nullptr, 0) != hipSuccess) {
                fill_graph_exec_ = nullptr;
                ok = false;
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (ok) {
                fill_graph_sig_ = sig;
                T4A_HIP(hipGraphLaunch(fill_graph_exec_, st));
                ++fill_stats_[2];
                done = true;
            } else {
                (void)hipGetLastError();
                fill_graph_broken_ = true; // never try again on this handle; fall through to the direct issue
            }
        }
    }
    fill_last_sig_ = sig;
    ++fill_stats_[0];
    if (!done)
        for (auto& f : ops) f();
    if (fill_timed_) T4A_HIP(hipEventRecord(ev_fill_.b, st));
}

void Tci2::flush_deferred_fill()
{
    if (fill_deferred_.empty()) return;
    std::vector<std::function<void()>> ops;
    ops.swap(fill_deferred_);
    issue_fill_ops(ops, fill_deferred_sig_);
    fill_inflight_ = true;
}

void Tci2::fill_wait()
{
    flush_deferred_fill();
    if (import_inflight_) { // cores of the other ranks' sites (site-sharded fill) still on their way into this handle
        import_inflight_ = false;
        T4A_HIP(hipStreamSynchronize(import_stream_));
    }
    if (!fill_inflight_) return;
    fill_inflight_ = false;
    T4A_HIP(hipStreamSynchronize(fill_stream_));
    T4A_HIP(hipGetLastError());
    if (eng.prof.enabled && fill_timed_) {
        float ms = 0.f;
        T4A_HIP(hipEventElapsedTime(&ms, ev_fill_.a, ev_fill_.b));
        eng.prof.v[4] += ms;
        eng.prof.v[5] += 1.0;
    }
    fill_timed_ = false;
    for (size_t k = 0; k < fill_solved_sites_.size(); ++k)
        if (h_fillinfo_.get()[fill_solved_sites_[k]] > 0) {
            const size_t site = fill_solved_sites_[k];
            fill_solved_sites_.clear();
            throw Error(T4A_GPU_INTERNAL_ERROR, "one-site interpolation solve failed: singular pivot matrix at site " +
                                                    std::to_string(site));
        }
    fill_solved_sites_.clear();
}

void Tci2::fill_site_tensors_impl(bool async)
{
    require_fn();
    // accumulators built ahead are only trusted when optimize() vouches for them (same half-sweep, sets final)
    const bool trust_cache = fill_cache_trusted_;
    fill_cache_trusted_ = false;
    static const bool host_prof_fill = std::getenv("T4A_HOST_PROFILE") != nullptr;
    const auto hpf_t0 = std::chrono::steady_clock::now();
    static double hpf_sec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto hpf_mark = [&](int k, std::chrono::steady_clock::time_point& t) {
        if (!host_prof_fill) return;
        const auto now = std::chrono::steady_clock::now();
        hpf_sec[k] += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    };
    auto hpf_t = hpf_t0;
    if (!trust_cache) invalidate_fill_cache();
    if (fn_kind_ != FnKind::Builtin) sync_digits();
    fill_wait(); // the scratch arenas of the previous fill are free again
    if (!fill_stream_) fill_stream_ = pool::stream_get(2); // lowest priority, recycled through the process-wide cache
    const bool builtin = fn_kind_ == FnKind::Builtin;
    static const bool sync_fill = std::getenv("T4A_SYNC_FILL") != nullptr; // measurement switch: no overlap at all
    if (!builtin || sync_fill) async = false;
    // deferred mode (optimize only): everything up to the upload buffer is prepared now, the ~25 stream operations are
    // issued later from an overlap hook, when the host would otherwise wait for a long bond-update kernel
    static const bool defer_env = diag_env("T4A_FILL_DEFER") != nullptr; // measured: no net gain (the fill then
    // overlaps the long mid-chain kernels and slows them down by as much as the host time it hides)
    const bool defer = async && builtin && fill_defer_requested_ && defer_env;
    fill_defer_requested_ = false;
    std::vector<std::function<void()>> ops; // the stream operations of this fill, in order (no event records)
    std::vector<uint64_t> sig;              // everything those operations depend on: equal signature <=> same graph
    auto dev = [&](std::function<void()> f) { ops.push_back(std::move(f)); };
    auto sg = [&](uint64_t v) { sig.push_back(v); };
    // everything of this fill is ordered after the work already enqueued on the main stream (the per-bond path writes site
    // tensors there).  After a bond chain nothing on the main stream concerns the fill — and the NEXT chain may already be
    // running on it: no wait then.
    const bool no_main_sync = fill_no_main_sync_ && fn_kind_ == FnKind::Builtin;
    fill_no_main_sync_ = false;
    if (!no_main_sync) T4A_HIP(hipStreamSynchronize(eng.stream()));
    hipStream_t st = fill_stream_;
    struct SiteJob {
        size_t b;
        size_t ni, nj, np;
        size_t offA, offB; // offsets (doubles) into d_fillA_ / d_fillB_
        size_t accJ, accK, accI; // offsets (u64) into the accumulator arena: J_b, kron_i(b), I_{b+1}
        bool last;
    };
    std::vector<SiteJob> jobs;
    size_t totA = 0, totB = 0;
    for (size_t b = 0; b < n_; ++b) {
        if (shard_world > 1 && (b % shard_world) != shard_rank) continue;
        const size_t ni = i_set[b].count * local_dims[b];
        const size_t nj = j_set[b].count;
        if (ni == 0 || nj == 0) { // tensorci2.rs:1074-1092
            const size_t left_dim = (b == 0) ? 1 : std::max<size_t>(i_set[b].count, 1);
            const size_t right_dim = (b == n_ - 1) ? 1 : std::max<size_t>(i_set[b + 1].count, 1);
            DevCore& c = cores[b];
            c.buf.reserve(std::max<size_t>(left_dim * local_dims[b] * right_dim, 1));
            c.l = left_dim;
            c.s = local_dims[b];
            c.r = right_dim;
            {
                double* ptr = c.buf.get();
                const size_t cnt = c.size();
                sg(0xF111ull);
                sg((uint64_t)(uintptr_t)ptr);
                sg((uint64_t)cnt);
                dev([=]() { fill_launch(ptr, cnt, 0.0, st); });
            }
            continue;
        }
        SiteJob j;
        j.b = b;
        j.ni = ni;
        j.nj = nj;
        j.last = (b == n_ - 1);
        j.np = j.last ? 0 : i_set[b + 1].count;
        if (!j.last && j.np != nj)
            throw Error(T4A_GPU_INTERNAL_ERROR, "one-site interpolation solve failed: pivot matrix at bond " +
                                                    std::to_string(b) + " is not square (" + std::to_string(j.np) + " x " +
                                                    std::to_string(nj) + ")");
        j.offA = totA;
        j.offB = totB;
        j.accJ = j.accK = j.accI = 0;
        totA += j.last ? 0 : nj * j.np;
        totB += nj * ni;
        jobs.push_back(j);
    }
    hpf_mark(0, hpf_t);
    if (jobs.empty()) {
        for (auto& f : ops) f();
        T4A_HIP(hipStreamSynchronize(st));
        return;
    }
    if (!builtin) { // host-callback mode stays immediate (its evaluations are synchronous anyway)
        for (auto& f : ops) f();
        ops.clear();
    }
    d_fillA_.reserve(std::max<size_t>(totA, 1));
    d_fillB_.reserve(totB);
    // max|P| bits (n_ u64) and solve status (n_ ints) share one allocation -> one memset
    // max|P| bits (n_ u64), solve status (n_ ints) and the work tickets of the trailing updates share one allocation -> one memset
    const size_t fm_words = n_ + (n_ + 1) / 2 + (LU_MAX_PANEL_STEPS + 1) / 2;
    d_fillmax_.reserve(fm_words);
    unsigned long long* d_max = d_fillmax_.get();
    int* d_info = reinterpret_cast<int*>(d_fillmax_.get() + n_);
    unsigned* d_tickets = reinterpret_cast<unsigned*>(d_fillmax_.get() + n_ + (n_ + 1) / 2);
    // beside a bond chain the fill keeps off the chain's XCD where it could not be placed anyway (lu_update_kernel)
    const int avoid_xcc = (no_main_sync && !xcd_disabled()) ? eng.xcc() : -1;
    h_fillinfo_.reserve(n_);
    {
        unsigned long long* ptr = d_fillmax_.get();
        const size_t bytes = fm_words * sizeof(unsigned long long);
        fill_timed_ = eng.prof.enabled;
        sg((uint64_t)(uintptr_t)ptr);
        sg((uint64_t)bytes);
        dev([=]() { T4A_HIP(hipMemsetAsync(ptr, 0, bytes, st)); });
        if (!builtin) {
            if (fill_timed_) T4A_HIP(hipEventRecord(ev_fill_.a, st));
            for (auto& f : ops) f();
            ops.clear();
        }
    }

    // core shapes are known up front: allocate them now so that every device address below is final
    for (const SiteJob& j : jobs) {
        const size_t left_dim = (j.b == 0) ? 1 : i_set[j.b].count;
        const size_t S = local_dims[j.b];
        DevCore& c = cores[j.b];
        c.l = left_dim;
        c.s = S;
        c.r = j.last ? 1 : j.np;
        c.buf.reserve(std::max<size_t>(c.size(), 1));
    }

    // solve descriptors (device addresses only): built before the upload so that ONE host-to-device copy carries
    // the accumulators, the evaluation jobs, the LU / triangular-solve problems and the packing jobs
    std::vector<LuProblem> lups;
    std::vector<TrsmProblem> trl, tru;
    std::vector<PackJob> packs;
    size_t piv_total = 0;
    for (const SiteJob& j : jobs)
        if (!j.last) piv_total += j.np;
    d_fillpiv_.reserve(std::max<size_t>(piv_total, 1));
    size_t piv_off = 0;
    int max_n = 0, max_nrhs = 0;
    size_t max_core = 1;
    fill_solved_sites_.clear();
    double flops = 0.0;
    for (const SiteJob& j : jobs) {
        const DevCore& c = cores[j.b];
        PackJob pk;
        pk.src = d_fillB_.get() + j.offB;
        pk.core = c.buf.get();
        pk.L = (int)c.l;
        pk.S = (int)c.s;
        pk.R = (int)c.r;
        pk.last = j.last ? 1 : 0;
        pk.ld = j.last ? (int)j.ni : (int)j.nj;
        pk.info = j.last ? nullptr : (const int*)(d_info + j.b);
        pk.pad_ = 0;
        packs.push_back(pk);
        max_core = std::max(max_core, c.size());
        if (j.last) continue;
        LuProblem lp;
        lp.A = d_fillA_.get() + j.offA;
        lp.lda = (int)j.nj;
        lp.n = (int)j.nj;
        lp.piv = d_fillpiv_.get() + piv_off;
        lp.info = d_info + j.b;
        lp.B = d_fillB_.get() + j.offB;
        lp.ldb = (int)j.nj;
        lp.nrhs = (int)j.ni;
        lp.pmax_bits = d_max + j.b;
        piv_off += j.np;
        lups.push_back(lp);
        TrsmProblem t;
        t.T = lp.A;
        t.ldt = lp.lda;
        t.n = lp.n;
        t.B = lp.B;
        t.ldb = lp.ldb;
        t.nrhs = lp.nrhs;
        t.lower = 1;
        t.unit_diag = 1;
        t.skip_flag = lp.info;
        trl.push_back(t);
        t.lower = 0;
        t.unit_diag = 0;
        tru.push_back(t);
        fill_solved_sites_.push_back(j.b);
        max_n = std::max(max_n, lp.n);
        max_nrhs = std::max(max_nrhs, lp.nrhs);
        const double n = (double)j.np;
        flops += (2.0 / 3.0) * n * n * n + 2.0 * n * n * (double)j.ni;
    }
    hpf_mark(1, hpf_t);
    const size_t np_ = lups.size();
    const size_t bytes_lu = np_ * sizeof(LuProblem), bytes_tr = 2 * np_ * sizeof(TrsmProblem);
    const size_t bytes_pk = packs.size() * sizeof(PackJob);
    auto up8 = [](size_t v) { return (v + 7) / 8 * 8; };

    // (1) evaluations.  B_b = Pi1^T (nj x ni), A_b = P^T (nj x np) — evaluated directly in transposed form
    //     (solve(P^T, Pi1^T), tensorci2.rs:1160-1162).
    const LuProblem* d_lups = nullptr;
    const TrsmProblem* d_trs = nullptr;
    const PackJob* d_packs = nullptr;
    const PiJob* d_pis = nullptr;
    // small problems (BASELINE configs[1], the first iterations of every run): evaluation, solve and packing of all sites in ONE
    // launch (fill_small_kernel) instead of five dependent ones — bitwise the same cores
    static const bool no_small_fill = diag_env("T4A_NO_SMALL_FILL") != nullptr;
    bool small_fill = builtin && !no_small_fill;
#ifdef T4A_TEST_HOOKS
    // libt4a_gpu_testhooks.so only: the general five-launch path for small problems too, so that a test can compare the two bit by bit
    // (tests/test_gpu_tci2.py::test_small_problem_fill_in_one_launch_is_bitwise_the_general_path)
    if (std::getenv("T4A_TEST_NO_SMALL_FILL")) small_fill = false;
#endif
    for (const SiteJob& j : jobs)
        if (j.nj > (size_t)FILL_SMALL_MAX_N || j.ni > (size_t)FILL_SMALL_MAX_RHS) small_fill = false;
    if (builtin) {
        std::vector<uint64_t> acc_all;
        const size_t Kacc = (size_t)fn_dev_.n_acc;
        for (SiteJob& j : jobs) {
            const bool cached = fill_cache_.size() == n_ && fill_cache_[j.b].valid &&
                                fill_cache_[j.b].accJ.size() == j.nj * Kacc && fill_cache_[j.b].accK.size() == j.ni * Kacc &&
                                (j.last || fill_cache_[j.b].accI.size() == j.np * Kacc);
            if (!cached) prepare_fill_site(j.b); // sites that became final only with the last bonds of the sweep
            const FillAcc& f = fill_cache_[j.b];
            j.accJ = acc_all.size();
            acc_all.insert(acc_all.end(), f.accJ.begin(), f.accJ.end());
            j.accK = acc_all.size();
            acc_all.insert(acc_all.end(), f.accK.begin(), f.accK.end());
            if (!j.last) {
                j.accI = acc_all.size();
                acc_all.insert(acc_all.end(), f.accI.begin(), f.accI.end());
            }
        }
        invalidate_fill_cache();
        hpf_mark(2, hpf_t);
        const size_t bytes_acc = acc_all.size() * sizeof(uint64_t);
        const size_t n_pi = 2 * jobs.size();
        const size_t off_pi = up8(bytes_acc), off_lu = up8(off_pi + n_pi * sizeof(PiJob));
        const size_t off_tr = up8(off_lu + bytes_lu), off_pk = up8(off_tr + bytes_tr);
        const size_t total_bytes = up8(off_pk + bytes_pk);
        h_fillacc_.reserve(total_bytes / 8);
        d_fillacc_.reserve(total_bytes / 8);
        hpf_mark(5, hpf_t);
        char* hb = reinterpret_cast<char*>(h_fillacc_.get());
        char* db = reinterpret_cast<char*>(d_fillacc_.get());
        const uint64_t* da = d_fillacc_.get();
        std::memcpy(hb, acc_all.data(), bytes_acc);
        hpf_mark(6, hpf_t);
        std::vector<PiJob> pis;
        int max_M = 0, max_N = 0;
        for (const SiteJob& j : jobs) {
            eng.prof.v[11] += (double)j.ni * j.nj + (double)j.np * j.nj;
            PiJob q;
            q.pad_ = 0;
            if (j.last) { // last site stores Pi1 itself (:1109-1128): ni x nj
                q.rowacc = da + j.accK;
                q.M = (int)j.ni;
                q.colacc = da + j.accJ;
                q.N = (int)j.nj;
                q.out = d_fillB_.get() + j.offB;
                q.ld = (int)j.ni;
                q.max_abs_bits = nullptr;
                pis.push_back(q);
            } else {
                q.rowacc = da + j.accJ;
                q.M = (int)j.nj;
                q.colacc = da + j.accK;
                q.N = (int)j.ni;
                q.out = d_fillB_.get() + j.offB;
                q.ld = (int)j.nj;
                q.max_abs_bits = nullptr;
                pis.push_back(q);
                q.colacc = da + j.accI;
                q.N = (int)j.np;
                q.out = d_fillA_.get() + j.offA;
                q.max_abs_bits = d_max + j.b;
                pis.push_back(q);
            }
        }
        for (const PiJob& q : pis) {
            max_M = std::max(max_M, q.M);
            max_N = std::max(max_N, q.N);
        }
        std::memcpy(hb + off_pi, pis.data(), pis.size() * sizeof(PiJob));
        if (np_) {
            std::memcpy(hb + off_lu, lups.data(), bytes_lu);
            std::memcpy(hb + off_tr, trl.data(), np_ * sizeof(TrsmProblem));
            std::memcpy(hb + off_tr + np_ * sizeof(TrsmProblem), tru.data(), np_ * sizeof(TrsmProblem));
        }
        std::memcpy(hb + off_pk, packs.data(), bytes_pk);
        hpf_mark(7, hpf_t);
        {
            const FnDevice fn = fn_dev_;
            const PiJob* dj = reinterpret_cast<const PiJob*>(db + off_pi);
            const int npi = (int)pis.size();
            sg((uint64_t)(uintptr_t)db);
            sg((uint64_t)(uintptr_t)hb);
            sg((uint64_t)total_bytes);
            sg((uint64_t)npi);
            sg(((uint64_t)(uint32_t)max_M << 32) | (uint32_t)max_N);
            sg((uint64_t)fn.fid * 16 + (uint64_t)fn.n_acc);
            for (int q = 0; q < T4A_FN_MAX_PARAMS; ++q) {
                uint64_t bits;
                std::memcpy(&bits, &fn.params[q], sizeof(bits));
                sg(bits);
            }
            static const bool copies_outside = diag_env("T4A_FILL_GRAPH_NO_COPY") != nullptr; // (diagnosis: the upload is issued directly, in front of the graph)
            if (copies_outside) {
                fill_pre_ops_.push_back([=]() { T4A_HIP(hipMemcpyAsync(db, hb, total_bytes, hipMemcpyHostToDevice, st)); });
                dev([=]() { pi_eval_batched_launch(fn, dj, npi, max_M, max_N, st); });
            } else {
                sg((uint64_t)small_fill);
                const bool small = small_fill;
                dev([=]() {
                    T4A_HIP(hipMemcpyAsync(db, hb, total_bytes, hipMemcpyHostToDevice, st));
                    if (!small) pi_eval_batched_launch(fn, dj, npi, max_M, max_N, st);
                });
            }
        }
        d_lups = reinterpret_cast<const LuProblem*>(db + off_lu);
        d_trs = reinterpret_cast<const TrsmProblem*>(db + off_tr);
        d_packs = reinterpret_cast<const PackJob*>(db + off_pk);
        d_pis = reinterpret_cast<const PiJob*>(db + off_pi);
    } else {
        // host callback: evaluated synchronously through eval_matrix (main stream), then continue on `st`
        for (const SiteJob& j : jobs) {
            IndexSet ik = kronecker_i(j.b);
            const IndexSet& jb = j_set[j.b];
            if (j.last) {
                eval_matrix(ik, 0, jb, ik.width, d_fillB_.get() + j.offB, nullptr);
            } else {
                eval_matrix(jb, j.b + 1, ik, 0, d_fillB_.get() + j.offB, nullptr);
                eval_matrix(jb, j.b + 1, i_set[j.b + 1], 0, d_fillA_.get() + j.offA, d_max + j.b);
            }
        }
        T4A_HIP(hipStreamSynchronize(eng.stream()));
        acc_used_ = 0;
        const size_t off_tr = up8(bytes_lu), off_pk = up8(off_tr + bytes_tr), total_bytes = up8(off_pk + bytes_pk);
        h_fillacc_.reserve(total_bytes / 8 + 1);
        d_fillacc_.reserve(total_bytes / 8 + 1);
        char* hb = reinterpret_cast<char*>(h_fillacc_.get());
        char* db = reinterpret_cast<char*>(d_fillacc_.get());
        if (np_) {
            std::memcpy(hb, lups.data(), bytes_lu);
            std::memcpy(hb + off_tr, trl.data(), np_ * sizeof(TrsmProblem));
            std::memcpy(hb + off_tr + np_ * sizeof(TrsmProblem), tru.data(), np_ * sizeof(TrsmProblem));
        }
        std::memcpy(hb + off_pk, packs.data(), bytes_pk);
        T4A_HIP(hipMemcpyAsync(db, hb, total_bytes, hipMemcpyHostToDevice, st));
        d_lups = reinterpret_cast<const LuProblem*>(db);
        d_trs = reinterpret_cast<const TrsmProblem*>(db + off_tr);
        d_packs = reinterpret_cast<const PackJob*>(db + off_pk);
    }

    // (2) batched solve; the zero-pivot-matrix guard (:1154-1157) is evaluated on the device: lu_kernel reads
    //     max|P| and flags info = -1, the solves skip flagged problems and the packing writes a zero core
    {
        const int npr = (int)np_;
        const unsigned gx = blocks_for(max_core) > 64 ? 64 : blocks_for(max_core), gy = (unsigned)packs.size();
        int* hinfo = h_fillinfo_.get();
        const size_t info_bytes = n_ * sizeof(int);
        sg((uint64_t)npr);
        sg((uint64_t)(int64_t)avoid_xcc);
        sg(((uint64_t)(uint32_t)max_n << 32) | (uint32_t)max_nrhs);
        sg(((uint64_t)gx << 32) | gy);
        sg((uint64_t)(uintptr_t)d_lups);
        sg((uint64_t)(uintptr_t)d_trs);
        sg((uint64_t)(uintptr_t)d_packs);
        sg((uint64_t)(uintptr_t)d_info);
        sg((uint64_t)(uintptr_t)hinfo);
        const FnDevice fn = fn_dev_;
        const int n_site_jobs = (int)jobs.size(), last_site = jobs.back().last ? 1 : 0;
        const bool small = small_fill;
        sg((uint64_t)(uintptr_t)d_pis);
        sg(((uint64_t)n_site_jobs << 1) | (uint64_t)last_site);
        dev([=]() {
            if (small) {
                fill_small_launch(fn, d_pis, d_lups, d_packs, n_site_jobs, last_site, st);
                T4A_HIP(hipMemcpyAsync(hinfo, d_info, info_bytes, hipMemcpyDeviceToHost, st));
                return;
            }
            if (npr) {
                // blocked LU with the unit-lower forward substitution of the right-hand sides folded in; beyond its size
                // limit the unblocked kernel + explicit forward solve (bitwise the same result)
                // (round 5) one fused solve behind the LU of the pivot matrices; outside its size range the two-step path
                if (!lu_solve_blocked_launch(d_lups, npr, max_n, max_nrhs, st, avoid_xcc, d_tickets)) {
                    if (!lu_forward_blocked_launch(d_lups, npr, max_n, max_nrhs, st, avoid_xcc, d_tickets)) {
                        lu_batched_launch(d_lups, npr, max_n, st);
                        trsm_left_batched_launch(d_trs, npr, max_n, max_nrhs, st);
                    }
                    trsm_left_batched_launch(d_trs + npr, npr, max_n, max_nrhs, st);
                }
            }
            // (3) pack all cores in one launch
            hipLaunchKernelGGL(pack_fill_batched_kernel, dim3(gx, gy), dim3(256), 0, st, d_packs);
            T4A_HIP(hipMemcpyAsync(hinfo, d_info, info_bytes, hipMemcpyDeviceToHost, st));
        });
    }
    hpf_mark(3, hpf_t);
    eng.prof.v[10] += flops;
    if (!builtin) {
        for (auto& f : ops) f();
        if (fill_timed_) T4A_HIP(hipEventRecord(ev_fill_.b, st));
        fill_inflight_ = true;
    } else if (defer) {
        fill_deferred_ = std::move(ops);
        fill_deferred_sig_ = std::move(sig);
        fill_inflight_ = false;
    } else {
        issue_fill_ops(ops, sig);
        fill_inflight_ = true;
    }
    if (host_prof_fill && !defer) {
        static double acc_ms = 0;
        static long calls = 0;
        acc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hpf_t0).count();
        hpf_mark(4, hpf_t);
        if (++calls % 20 == 0)
            std::fprintf(stderr, "[host profile] fill_site_tensors host part %.1f us per call (wait+jobs %.1f, descriptors %.1f, "
                                 "accumulators %.1f, staging %.1f, issue %.1f)\n",
                         1e3 * acc_ms / calls, hpf_sec[0] / calls, hpf_sec[1] / calls, hpf_sec[2] / calls, hpf_sec[3] / calls,
                         hpf_sec[4] / calls);
        if (calls % 20 == 0)
            std::fprintf(stderr, "[host profile]   staging in detail: reserve %.1f, accumulator copy %.1f, job tables %.1f us per call\n",
                         hpf_sec[5] / calls, hpf_sec[6] / calls, hpf_sec[7] / calls);
    }
    if (!async) fill_wait();
}

// the legacy default stream, or a blocking stream (which orders against it exactly like it): no graph replay beside these
static bool stream_orders_with_legacy(hipStream_t s)
{
    if (s == nullptr) return true;
    unsigned flags = 0;
    if (hipStreamGetFlags(s, &flags) != hipSuccess) {
        (void)hipGetLastError();
        return true; // (unknown: the careful answer)
    }
    return (flags & hipStreamNonBlocking) == 0;
}

// Copies every site tensor to dst + site * stride (doubles) WITHOUT blocking the host: the copies are ordered after a
// fill that is still in flight (same stream) and `consumer` waits for them through an event.
void Tci2::export_site_tensors_async(double* d_dst, size_t stride, hipStream_t consumer)
{
    cores_shared_async_ = true;
    if (stream_orders_with_legacy(consumer)) cores_shared_legacy_stream_ = true; // (see issue_fill_ops)
    hipStream_t st = fill_inflight_ ? fill_stream_ : eng.stream();
    for (size_t s = 0; s < n_; ++s) {
        const DevCore& c = cores[s];
        if (c.size() > stride) throw Error(T4A_GPU_BUFFER_TOO_SMALL, "export_site_tensors: stride smaller than a site tensor");
        if (c.size())
            T4A_HIP(hipMemcpyAsync(d_dst + s * stride, c.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    if (!export_event_) T4A_HIP(hipEventCreateWithFlags(&export_event_, hipEventDisableTiming));
    T4A_HIP(hipEventRecord(export_event_, st));
    T4A_HIP(hipStreamWaitEvent(consumer, export_event_, 0));
}

// Site-sharded fill (BASELINE.json configs[3]): the sites s = shard_rank + shard_world * k of this rank go to
// d_dst + k * stride on the stream of the fill that may still be in flight; `consumer` (the stream of the all-gather) waits.
void Tci2::export_site_shard_async(double* d_dst, size_t stride, hipStream_t consumer)
{
    cores_shared_async_ = true;
    if (stream_orders_with_legacy(consumer)) cores_shared_legacy_stream_ = true; // (see issue_fill_ops)
    hipStream_t st = fill_inflight_ ? fill_stream_ : eng.stream();
    static const bool export_sync = diag_env("T4A_EXPORT_SYNC") != nullptr; // (diagnosis: the fill has completed before the copies are enqueued)
    if (export_sync) T4A_HIP(hipStreamSynchronize(st));
    size_t k = 0;
    for (size_t s = shard_rank; s < n_; s += shard_world, ++k) {
        const DevCore& c = cores[s];
        if (c.size() > stride) throw Error(T4A_GPU_BUFFER_TOO_SMALL, "export_site_shard: stride smaller than a site tensor");
        if (c.size())
            T4A_HIP(hipMemcpyAsync(d_dst + k * stride, c.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    static const bool export_hostsync = diag_env("T4A_EXPORT_HOSTSYNC") != nullptr; // (diagnosis: no event, the host waits for the copies)
    if (export_hostsync) {
        T4A_HIP(hipStreamSynchronize(st));
        return;
    }
    if (!export_event_) T4A_HIP(hipEventCreateWithFlags(&export_event_, hipEventDisableTiming));
    T4A_HIP(hipEventRecord(export_event_, st));
    T4A_HIP(hipStreamWaitEvent(consumer, export_event_, 0));
}

// The other ranks' cores out of the gathered buffer [world][per_rank][stride]: site s of rank r = s % world sits at
// (r * per_rank + s / world) * stride.  Shapes follow from the (replicated) index sets AS THEY ARE NOW: call this in the same
// half-sweep as the export on the other ranks (parallel.ShardedCoreExchange does), before the next bond update changes a
// bond dimension.  The copies run on the handle's import stream after everything `producer` (the all-gather's stream) has
// enqueued so far; nothing blocks the host, the local fill still in flight is not waited for (it writes other sites), and
// the first reader of a core (fill_wait) waits for the import.
void Tci2::import_site_shard_async(const double* d_src, size_t stride, size_t per_rank, hipStream_t producer)
{
    cores_shared_async_ = true;
    if (stream_orders_with_legacy(producer)) cores_shared_legacy_stream_ = true; // (see issue_fill_ops)
    if (!import_stream_) import_stream_ = pool::stream_get(2);
    // at most one import in flight: the one of the previous half-sweep is long done (a whole chain of bond updates ago),
    // and with it every read of the receive buffer that the caller is about to reuse
    T4A_HIP(hipStreamSynchronize(import_stream_));
    static const bool import_hostsync = diag_env("T4A_IMPORT_HOSTSYNC") != nullptr; // (diagnosis: the host waits for the producer, no event)
    if (import_hostsync) {
        T4A_HIP(hipStreamSynchronize(producer));
    } else {
        if (!import_event_) T4A_HIP(hipEventCreateWithFlags(&import_event_, hipEventDisableTiming));
        T4A_HIP(hipEventRecord(import_event_, producer));
        T4A_HIP(hipStreamWaitEvent(import_stream_, import_event_, 0));
    }
    for (size_t s = 0; s < n_; ++s) {
        const size_t r = s % shard_world;
        if (r == shard_rank) continue;
        DevCore& c = cores[s];
        const size_t l = (s == 0) ? 1 : std::max<size_t>(i_set[s].count, 1);
        const size_t rr = (s + 1 == n_) ? 1 : std::max<size_t>(i_set[s + 1].count, 1);
        const size_t count = l * local_dims[s] * rr;
        if (count > stride) throw Error(T4A_GPU_BUFFER_TOO_SMALL, "import_site_shard: stride smaller than a site tensor");
        c.buf.reserve(std::max<size_t>(count, 1));
        c.l = l;
        c.s = local_dims[s];
        c.r = rr;
        T4A_HIP(hipMemcpyAsync(c.buf.get(), d_src + (r * per_rank + s / shard_world) * stride, count * sizeof(double),
                               hipMemcpyDeviceToDevice, import_stream_));
    }
    import_inflight_ = true;
}

// =================================================================================================
// TT evaluation / sum
// =================================================================================================
std::vector<double> Tci2::site_tensor_host(size_t site, size_t dims3[3])
{
    fill_wait();
    const DevCore& c = cores[site];
    dims3[0] = c.l;
    dims3[1] = c.s;
    dims3[2] = c.r;
    std::vector<double> h(c.size());
    if (!h.empty()) {
        T4A_HIP(hipMemcpyAsync(h.data(), c.buf.get(), h.size() * sizeof(double), hipMemcpyDeviceToHost, eng.stream()));
        T4A_HIP(hipStreamSynchronize(eng.stream()));
    }
    return h;
}

static void check_tt_chain(const std::vector<DevCore>& cores) // SimpleTensorTrain::new, simplett/src/tensortrain.rs:97
{
    if (cores.front().l != 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "First tensor must have left dimension 1");
    if (cores.back().r != 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "Last tensor must have right dimension 1");
    for (size_t i = 0; i + 1 < cores.size(); ++i)
        if (cores[i].r != cores[i + 1].l || cores[i].r == 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "tensor train bond dimension mismatch at bond " + std::to_string(i));
}

std::vector<double> Tci2::evaluate(const uint32_t* idx, size_t n_pts)
{
    fill_wait();
    check_tt_chain(cores);
    std::vector<double> out(n_pts);
    if (n_pts == 0) return out;
    for (size_t p = 0; p < n_pts; ++p)
        for (size_t s = 0; s < n_; ++s)
            if (idx[p * n_ + s] >= local_dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate: index out of bounds");
    hipStream_t st = eng.stream();
    std::vector<TtCoreDesc> desc(n_);
    int max_bond = 1;
    for (size_t s = 0; s < n_; ++s) {
        desc[s].data = cores[s].buf.get();
        desc[s].l = (int)cores[s].l;
        desc[s].d = (int)cores[s].s;
        desc[s].r = (int)cores[s].r;
        max_bond = std::max(max_bond, std::max(desc[s].l, desc[s].r));
    }
    d_coredesc_.reserve(n_);
    d_idx_.reserve(n_pts * n_);
    d_vals_.reserve(n_pts);
    T4A_HIP(hipMemcpyAsync(d_coredesc_.get(), desc.data(), n_ * sizeof(TtCoreDesc), hipMemcpyHostToDevice, st));
    T4A_HIP(hipMemcpyAsync(d_idx_.get(), idx, n_pts * n_ * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    T4A_HIP(hipStreamSynchronize(st));
    tt_evaluate_launch(d_coredesc_.get(), (int)n_, max_bond, d_idx_.get(), (int)n_pts, d_vals_.get(), st);
    T4A_HIP(hipMemcpyAsync(out.data(), d_vals_.get(), n_pts * sizeof(double), hipMemcpyDeviceToHost, st));
    T4A_HIP(hipStreamSynchronize(st));
    T4A_HIP(hipGetLastError());
    return out;
}

double Tci2::sum() // simplett/src/traits.rs:231-275 (host-side: O(n chi^2 d), not on the hot path)
{
    check_tt_chain(cores);
    std::vector<double> cur;
    for (size_t site = 0; site < n_; ++site) {
        size_t d3[3];
        std::vector<double> t = site_tensor_host(site, d3);
        const size_t L = d3[0], S = d3[1], R = d3[2];
        if (site == 0) {
            cur.assign(R, 0.0);
            for (size_t s = 0; s < S; ++s)
                for (size_t r = 0; r < R; ++r) cur[r] = cur[r] + t[0 + L * (s + S * r)];
            continue;
        }
        std::vector<double> site_sum(L * R, 0.0);
        for (size_t l = 0; l < L; ++l)
            for (size_t s = 0; s < S; ++s)
                for (size_t r = 0; r < R; ++r) site_sum[l * R + r] = site_sum[l * R + r] + t[l + L * (s + S * r)];
        std::vector<double> next(R, 0.0);
        for (size_t r = 0; r < R; ++r) {
            double sum = 0.0;
            for (size_t l = 0; l < L; ++l) sum = sum + cur[l] * site_sum[l * R + r];
            next[r] = sum;
        }
        cur.swap(next);
    }
    return cur[0];
}

// =================================================================================================
// global pivot search + optimisation loop
// =================================================================================================
// DefaultGlobalPivotFinder::find_global_pivots (tensorci/src/globalpivot.rs:160-219).  All candidate points
// of all searches are independent (the reference resets the coordinate after each 1-D scan), so f and the
// TT are evaluated in two batches.  RNG: rand 0.9 StdRng + random_range(0..d), restated in stdrng.hpp (initial points are drawn
// search by search, site by site, exactly as globalpivot.rs:174-180 does).
std::vector<std::vector<uint32_t>> Tci2::find_global_pivots(double abs_tol, const TCI2Options& o, StdRng& rng)
{
    std::vector<std::vector<uint32_t>> found;
    if (o.nsearch == 0) return found;
    std::vector<std::vector<uint32_t>> initial(o.nsearch, std::vector<uint32_t>(n_));
    for (auto& p : initial)
        for (size_t s = 0; s < n_; ++s) p[s] = (uint32_t)rng.random_range(local_dims[s]);
    std::vector<uint32_t> idx;
    for (const auto& point : initial)
        for (size_t p = 0; p < n_; ++p)
            for (size_t v = 0; v < local_dims[p]; ++v) {
                const size_t base = idx.size();
                idx.insert(idx.end(), point.begin(), point.end());
                idx[base + p] = (uint32_t)v;
            }
    const size_t npts = idx.size() / n_;
    std::vector<double> fv = eval_points_host(idx, npts);
    std::vector<double> tv;
    try {
        tv = evaluate(idx.data(), npts);
    } catch (const Error&) {
        tv.assign(npts, 0.0); // `.unwrap_or(T::zero())`
    }
    size_t q = 0;
    for (const auto& point : initial) {
        double best_error = 0.0;
        std::vector<uint32_t> best_point = point;
        for (size_t p = 0; p < n_; ++p)
            for (size_t v = 0; v < local_dims[p]; ++v, ++q) {
                const double diff = fv[q] - tv[q];
                const double err = std::sqrt(diff * diff);
                if (err > best_error) {
                    best_error = err;
                    best_point.assign(idx.begin() + q * n_, idx.begin() + (q + 1) * n_);
                }
            }
        if (best_error > abs_tol * o.tol_margin_global_search) found.push_back(best_point);
    }
    if (found.size() > o.max_nglobal_pivot) found.resize(o.max_nglobal_pivot);
    return found;
}

static bool convergence_criterion(const std::vector<size_t>& ranks, const std::vector<double>& errors,
                                  const std::vector<size_t>& nglobal, double tolerance, size_t max_bond_dim,
                                  size_t ncheck_history, int& out) // tensorci2.rs:1407-1437
{
    if (errors.size() < ncheck_history) return false;
    const size_t n = errors.size();
    bool errors_converged = true, no_global = true, at_max = true;
    size_t min_rank = std::numeric_limits<size_t>::max();
    for (size_t i = n - ncheck_history; i < n; ++i) {
        if (!(errors[i] < tolerance)) errors_converged = false;
        if (nglobal[i] != 0) no_global = false;
        if (!(ranks[i] >= max_bond_dim)) at_max = false;
        min_rank = std::min(min_rank, ranks[i]);
    }
    const bool rank_stable = (min_rank == ranks[n - 1]);
    if (at_max) {
        out = T4A_GPU_TCI2_MAX_BOND_DIMENSION;
        return true;
    }
    if (errors_converged && no_global && rank_stable) {
        out = T4A_GPU_TCI2_CONVERGED;
        return true;
    }
    return false;
}

// optimize_with_finder (tensorci2.rs:1626-1802)
// T4A_OPT_PROF=1: host time of the pieces of an iteration (microseconds, accumulated; printed by optimize())
static double g_opt_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
struct OptSeg {
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit OptSeg(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~OptSeg() { g_opt_seg[k] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

// optimize_with_finder (tensorci2.rs:1626-1802) as a resumable run: opt_begin, then per iteration opt_iter_start (prelude + the
// half-sweep enqueued as a bond chain, without waiting) and opt_iter_finish (the rest of the iteration), then opt_end.  optimize()
// drives one handle; optimize_group() drives up to eight in lock-step from one host thread — every handle's chain runs on its
// own XCD, so their bond updates overlap on the device while the host only ever waits for the slowest.
void Tci2::opt_begin(OptRun& r)
{
    const TCI2Options& options = r.options;

    options.validate();
    require_fn();
    if (rank() == 0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 state must contain at least one pivot before optimization");
    // pipelined mode (keep_site_tensors): a fill left in flight by the previous call keeps running beside the bond
    // updates below (they only touch the index sets); it is completed by the next fill or the first reader of a core
    if (!keep_site_tensors) fill_wait();
    ranks_hist.clear();
    errors_hist.clear();
    r.nglobal_hist.clear();
    termination = T4A_GPU_TCI2_MAX_ITERATIONS;
    r.rng.reseed(options.has_seed ? options.seed : 0x1234567ull); // (no seed: OS entropy in the reference — any stream will do)
    r.pending_fill = false;
    r.iter = 0;
    r.done = false;
    r.small_complete = false;
    // small problems: the whole loop in ONE launch (tci2_small.hip); it may finish the call, run the first iterations and hand the
    // rest back, or decline
    if (small_engine_run(r) && r.small_complete) return;
    // bounded rank, built-in functor: site tensors and fill workspaces get their final size now (a buffer that grows goes
    // through the process-wide cache, which waits for the whole device: once per iteration and buffer while ranks grow)
    if (fn_kind_ == FnKind::Builtin && options.max_bond_dim != 0 && options.max_bond_dim <= 1024) {
        const size_t chi = options.max_bond_dim;
        // what a bond can reach at all: min(chi, product of the local dimensions on either side) — the ends of a train stay small
        std::vector<size_t> lb(n_ + 1, 1), rb(n_ + 1, 1);
        for (size_t b = 0; b < n_; ++b) lb[b + 1] = std::min(chi, lb[b] * local_dims[b]);
        for (size_t b = n_; b-- > 0;) rb[b] = std::min(chi, rb[b + 1] * local_dims[b]);
        static const bool old_presize = diag_env("T4A_OLD_PRESIZE") != nullptr; // (debug: every bond at chi)
        auto bond = [&](size_t b) { return old_presize ? chi : std::min(lb[b], rb[b]); }; // bond b sits left of site b
        size_t totA = 0, totB = 0, tot_cores = 0;
        for (size_t b = 0; b < n_; ++b) {
            if (shard_world > 1 && (b % shard_world) != shard_rank) continue;
            const size_t l = bond(b), rr = bond(b + 1);
            tot_cores += l * local_dims[b] * rr;
            totA += rr * rr;
            totB += rr * l * local_dims[b];
        }
        // ... and only while the whole reservation stays moderate (chi = 1024 over 100 sites is 4 GB per handle,
        // times eight handles of a group, for a run that may stay at rank 10): beyond that the buffers grow on demand as before
        const bool presize = (tot_cores + totA + totB) * sizeof(double) <= ((size_t)1 << 30);
        for (size_t b = 0; presize && b < n_; ++b) {
            if (shard_world > 1 && (b % shard_world) != shard_rank) continue;
            cores[b].buf.reserve(std::max<size_t>(bond(b) * local_dims[b] * bond(b + 1), 1));
        }
        if (presize) {
        d_fillA_.reserve(std::max<size_t>(totA, 1));
        d_fillB_.reserve(std::max<size_t>(totB, 1));
        d_fillpiv_.reserve(std::max<size_t>(n_ * chi, 1));
        // the upload arena of a fill (accumulators of J_b, kron(I_b, d_b), I_{b+1} per site, then the job tables): grown once
        // per iteration it cost every fill a device-wide wait in the middle of the next chain
        size_t acc_words = 0;
        for (size_t b = 0; b < n_; ++b) {
            if (shard_world > 1 && (b % shard_world) != shard_rank) continue;
            acc_words += chi * (2 + local_dims[b]) * (size_t)fn_dev_.n_acc;
        }
        const size_t desc_bytes = n_ * (2 * sizeof(PiJob) + sizeof(LuProblem) + 2 * sizeof(TrsmProblem) + 64) + 256;
        h_fillacc_.reserve(acc_words + desc_bytes / 8 + 64);
        d_fillacc_.reserve(acc_words + desc_bytes / 8 + 64);
        d_fillmax_.reserve(n_ + (n_ + 1) / 2 + (LU_MAX_PANEL_STEPS + 1) / 2);
        h_fillinfo_.reserve(n_);
        }
    }
}

bool Tci2::opt_iter_start(OptRun& r, bool defer_launch)
{
    const TCI2Options& options = r.options;
    if (r.done || r.iter >= options.max_iter) return false;
    const size_t iter = r.iter;
    (void)iter;
        r.norm = (options.normalize_error && max_sample_value > 0.0) ? max_sample_value : 1.0;
        r.abs_tol = options.tolerance * r.norm;
        bool is_forward = true;
        if (options.sweep_strategy == 1) is_forward = false;
        else if (options.sweep_strategy == 2) is_forward = (iter % 2 == 0);

        // extras of this iteration: the sets at the start of the previous one (:1675-1685); then the sets as they are now join
        // the history (:1686-1689).  After a bond chain only codes and counts are current: the entry carries those.
        long ext_idx = (!options.strictly_nested && !history.empty()) ? (long)history.size() - 1 : -1;
        {
            OptSeg seg_(0);
            HistEntry cur;
            cur.serial = ++chain_.hist_serial;
            if (chain_.digits_stale) {
                cur.digits_valid = false;
                cur.cap = chain_.cap;
                cur.is.resize(n_);
                cur.js.resize(n_);
                for (size_t p = 0; p < n_; ++p) {
                    cur.is[p].width = i_set[p].width;
                    cur.is[p].count = i_set[p].count;
                    cur.js[p].width = j_set[p].width;
                    cur.js[p].count = j_set[p].count;
                }
                const ChainTab mi = chain_mirror(chain_.mcur, 0), mj = chain_mirror(chain_.mcur, 1);
                cur.code.resize(2 * n_ * chain_.cap);
                std::memcpy(cur.code.data(), mi.code, n_ * chain_.cap * sizeof(uint64_t));
                std::memcpy(cur.code.data() + n_ * chain_.cap, mj.code, n_ * chain_.cap * sizeof(uint64_t));
            } else {
                cur.is = i_set;
                cur.js = j_set;
            }
            history.push_back(std::move(cur));
            // only the most recent snapshot is ever read (:1677-1681): cap the memory held by older ones
            if (history.size() > 2) {
                history.erase(history.begin());
                if (ext_idx >= 0) --ext_idx;
            }
        }
        {
            OptSeg seg_(1);
            invalidate_site_tensors();
            flush_pivot_errors();
            last_sweep_shapes.assign(n_ - 1, {0, 0, 0});
            prep_.valid = false;
            invalidate_fill_cache();
        }
        const bool fill_ahead = fn_kind_ == FnKind::Builtin && options.pivot_search == 0;
        // a fill deferred by the previous iteration is issued from the hook of the 9th bond of this half-sweep (the
        // first kernels that are long enough to hide the host work); shorter chains: from the last bond
        const size_t nb_ = n_ - 1;
        static const int defer_k = diag_env("T4A_FILL_DEFER") ? std::atoi(diag_env("T4A_FILL_DEFER")) : 8;
        const size_t want_k = defer_k > 0 ? (size_t)defer_k : 8;
        const size_t flush_k = nb_ > want_k + 1 ? want_k : nb_ - 1;
        const size_t flush_at_fwd = flush_k, flush_at_bwd = nb_ - 1 - flush_k;
        flush_deferred_fill();
        // built-in functor, full pivot search: the whole half-sweep is enqueued at once (tci2_chain.hip) ...
        r.is_forward = is_forward;
        r.ext_idx = ext_idx;
        r.fill_ahead = fill_ahead;
        r.flush_at_fwd = flush_at_fwd;
        r.flush_at_bwd = flush_at_bwd;
        {
            OptSeg seg_(2);
            r.chained = chain_enqueue(is_forward, options, ext_idx, true, !defer_launch);
        }
    return true;
}

// ... and while the device works on the chain the host issues fill_site_tensors of the PREVIOUS iteration (its accumulators were
// taken from the mirror when that iteration finished; nothing of it touches the main stream).  Part of opt_iter_finish; a
// group calls it for every handle before it finishes the first one (the first chain_finish waits for the whole group's chain:
// whatever the host issues after that no longer overlaps it).
void Tci2::opt_iter_issue_pending_fill(OptRun& r)
{
    if (!r.pending_fill) return;
    r.pending_fill = false;
#ifdef T4A_TEST_HOOKS
    {   // libt4a_gpu_testhooks.so only (build.py; tests/test_gpu_chain.py loads it in a child process): T4A_TEST_THROW_IN_FILL=n makes
        // the n-th pending fill of the process fail while a chain is in flight.  The production library carries no fault injector.
        static const long inject_at = std::getenv("T4A_TEST_THROW_IN_FILL") ? std::atol(std::getenv("T4A_TEST_THROW_IN_FILL")) : 0;
        static std::atomic<long> issued{0};
        if (inject_at > 0 && ++issued == inject_at)
            throw Error(T4A_GPU_INTERNAL_ERROR, "injected failure while issuing fill_site_tensors (T4A_TEST_THROW_IN_FILL)");
    }
#endif
    for (size_t b = 0; b < n_; ++b) prepare_fill_site(b); // (from the mirror of the previous chain; the new one writes the other mirror)
    fill_cache_trusted_ = true;
    fill_no_main_sync_ = true;
    fill_site_tensors_impl(true);
    if (!keep_site_tensors) invalidate_site_tensors(); // (the end of that iteration invalidated them, tensorci2.rs:707-708)
}

void Tci2::opt_iter_finish(OptRun& r)
{
    const TCI2Options& options = r.options;
    const size_t iter = r.iter;
    const double norm = r.norm, abs_tol = r.abs_tol;
    const bool is_forward = r.is_forward, chained = r.chained, fill_ahead = r.fill_ahead;
    const long ext_idx = r.ext_idx;
    const size_t flush_at_fwd = r.flush_at_fwd, flush_at_bwd = r.flush_at_bwd;
    do { // (one pass; `break` = the convergence exit of the reference's loop)
        {
            OptSeg seg_(3);
            opt_iter_issue_pending_fill(r);
        }
        if (chained) {
            OptSeg seg_(4);
            chain_finish(options);
        } else {
            // ... otherwise bond by bond
            sync_digits();
            std::vector<IndexSet> no_i(n_), no_j(n_);
            for (size_t p = 0; p < n_; ++p) {
                no_i[p].width = p;
                no_j[p].width = n_ - p - 1;
            }
            if (ext_idx >= 0) hist_digits(history[(size_t)ext_idx]);
            const std::vector<IndexSet>& extra_i = ext_idx >= 0 ? history[(size_t)ext_idx].is : no_i;
            const std::vector<IndexSet>& extra_j = ext_idx >= 0 ? history[(size_t)ext_idx].js : no_j;
            if (is_forward) {
                for (size_t b = 0; b + 1 < n_; ++b) {
                    // bond b+1 reads J_{b+2}, which bond b does not touch: its column side can be built ahead
                    prefetch_.wanted = b + 2 < n_;
                    prefetch_.bond = b + 1;
                    prefetch_.cols = true;
                    prefetch_.extra = prefetch_.wanted ? &extra_j[b + 1] : nullptr;
                    // site b-1 only reads I_{b-1}, J_{b-1}, I_b: final since bond b-1 (forward bonds write I_{b+1}, J_b)
                    prefetch_.fill_site = (fill_ahead && b >= 1) ? (long)(b - 1) : -1;
                    prefetch_.flush_fill = (b == flush_at_fwd);
                    update_pivots(b, true, options, extra_i[b + 1], extra_j[b]);
                }
            } else {
                for (size_t b = n_ - 1; b-- > 0;) {
                    // bond b-1 reads I_{b-1}, which bond b does not touch: its row side can be built ahead
                    prefetch_.wanted = b > 0;
                    prefetch_.bond = b - 1;
                    prefetch_.cols = false;
                    prefetch_.extra = prefetch_.wanted ? &extra_i[b] : nullptr;
                    // site b+2 reads I_{b+2}, J_{b+2}, I_{b+3}: final since bond b+1 (backward bonds write I_{b+1}, J_b)
                    prefetch_.fill_site = (fill_ahead && b + 2 < n_) ? (long)(b + 2) : -1;
                    prefetch_.flush_fill = (b == flush_at_bwd);
                    update_pivots(b, false, options, extra_i[b + 1], extra_j[b]);
                }
            }
        }
        prefetch_.wanted = false;
        prefetch_.fill_site = -1;
        prep_.valid = false;
        // the cores are not needed by the next half-sweep unless the global pivot search evaluates the TT
        flush_deferred_fill(); // (normally already gone; guarantees the order of consecutive fills)
        const bool fill_async = options.nsearch == 0 && !options.strictly_nested;
        if (chained && chain_.digits_stale && fill_async) {
            // this fill's accumulators (out of the mirror), descriptors and launches go out after the NEXT iteration's chain
            // has been enqueued (or after the loop): the device is busy with that chain while the host prepares the fill
            r.pending_fill = true;
        } else {
            fill_cache_trusted_ = fill_ahead && !chained;
            fill_defer_requested_ = fill_ahead && !chained && iter + 1 < options.max_iter;
            fill_no_main_sync_ = chained && chain_.digits_stale;
            fill_site_tensors_impl(fill_async);
        }
        OptSeg seg_tail_(5);
        const double error = max_bond_error();
        errors_hist.push_back(error / norm);

        std::vector<std::vector<uint32_t>> gp = find_global_pivots(abs_tol, options, r.rng);
        // invalidates the site tensors even for an empty list (tensorci2.rs:707-708) unless the caller opted out
        if (!(gp.empty() && keep_site_tensors)) add_global_pivots(gp);
        r.nglobal_hist.push_back(gp.size());
        ranks_hist.push_back(rank());
        if (options.verbosity > 0)
            std::printf("iteration = %zu, rank = %zu, error = %.2e, maxsamplevalue = %.2e, nglobalpivot = %zu\n", iter + 1,
                        rank(), error / norm, max_sample_value, gp.size());
        int reason;
        if (convergence_criterion(ranks_hist, errors_hist, r.nglobal_hist, options.tolerance, options.max_bond_dim_or_max(),
                                  options.ncheck_history, reason)) {
            termination = reason;
            r.done = true;
            break;
        }
    } while (false);
    ++r.iter;
}

void Tci2::opt_end_issue_fill(OptRun& r)
{
    if (r.pending_fill) { // the last iteration's fill
        r.pending_fill = false;
        for (size_t b = 0; b < n_; ++b) prepare_fill_site(b);
        fill_cache_trusted_ = true;
        fill_no_main_sync_ = true;
        fill_site_tensors_impl(true);
        if (!keep_site_tensors) invalidate_site_tensors(); // (the end of that iteration invalidated them, tensorci2.rs:707-708)
    }
}

void Tci2::opt_end(OptRun& r)
{
    const TCI2Options& options = r.options;
    const bool final_sweep1site = r.final_sweep1site;
    opt_end_issue_fill(r);
    // the cores of the last iteration are complete (deferred solve errors surface here); in pipelined mode the wait
    // is left to the first reader (site_tensor*, evaluate, export_site_tensors_async, the next fill)
    flush_deferred_fill();
    if (!keep_site_tensors || final_sweep1site) fill_wait();
    if (final_sweep1site) { // :1781-1794
        const double norm = (options.normalize_error && max_sample_value > 0.0) ? max_sample_value : 1.0;
        const double abs_tol = options.tolerance * norm;
        sweep1site(true, 1e-14, abs_tol, options.max_bond_dim_or_max(), true);
    }
}

void Tci2::optimize(const TCI2Options& options, bool final_sweep1site)
{
    OptRun r;
    r.options = options;
    r.final_sweep1site = final_sweep1site;
    static const bool prof = std::getenv("T4A_OPT_PROF") != nullptr; // host time between two chains (stderr, once per call)
    try {
        opt_begin(r);
        if (!prof) {
            while (opt_iter_start(r)) opt_iter_finish(r);
        } else {
            double t_start = 0.0, t_finish = 0.0;
            const double wait0 = g_chain_wait_seconds;
            size_t iters = 0;
            for (;;) {
                const auto ta = std::chrono::steady_clock::now();
                if (!opt_iter_start(r)) break;
                const auto tb = std::chrono::steady_clock::now();
                opt_iter_finish(r);
                const auto tc = std::chrono::steady_clock::now();
                t_start += std::chrono::duration<double>(tb - ta).count();
                t_finish += std::chrono::duration<double>(tc - tb).count();
                ++iters;
            }
            std::fprintf(stderr, "[t4a] optimize segments (us per iteration): history snapshot %.1f, invalidate + flush %.1f, chain_enqueue %.1f | pending fill %.1f, chain_finish incl. the wait %.1f, error + global pivots + convergence %.1f\n",
                         g_opt_seg[0] / std::max<size_t>(iters, 1), g_opt_seg[1] / std::max<size_t>(iters, 1), g_opt_seg[2] / std::max<size_t>(iters, 1), g_opt_seg[3] / std::max<size_t>(iters, 1),
                         g_opt_seg[4] / std::max<size_t>(iters, 1), g_opt_seg[5] / std::max<size_t>(iters, 1));
            for (double& v : g_opt_seg) v = 0.0;
            const double wait = g_chain_wait_seconds - wait0;
            std::fprintf(stderr, "[t4a] optimize: %zu iterations: start (prepare + launch the chain) %.1f us, finish %.1f us of which waiting for the device %.1f us, per iteration\n",
                         iters, 1e6 * t_start / std::max<size_t>(iters, 1), 1e6 * t_finish / std::max<size_t>(iters, 1), 1e6 * wait / std::max<size_t>(iters, 1));
        }
        opt_end(r);
    } catch (...) {
        chain_abort(); // (a chain may be in flight: the error came from the fill of the previous iteration)
        throw;
    }
}

// Up to eight handles (one XCD each) optimised in lock-step by the calling thread: every iteration first enqueues all the bond
// chains, then finishes them one after the other.  Results are exactly those of optimize() on every handle.
void Tci2::optimize_group(const std::vector<Tci2*>& hs, const TCI2Options& options, bool final_sweep1site)
{
    if (hs.empty()) return;
    if (hs.size() > 8) throw Error(T4A_GPU_INVALID_ARGUMENT, "optimize_group: at most eight handles (one per XCD)");
    for (size_t i = 0; i < hs.size(); ++i) {
        if (!hs[i]) throw Error(T4A_GPU_NULL_POINTER, "optimize_group: null handle");
        for (size_t j = 0; j < i; ++j)
            if (hs[j] == hs[i]) throw Error(T4A_GPU_INVALID_ARGUMENT, "optimize_group: a handle appears twice");
        hs[i]->eng.set_xcc((int)i); // (distinct XCDs: a handle keeps its reservation from enqueue to finish)
    }
    std::vector<OptRun> runs(hs.size());
    try {
    for (size_t i = 0; i < hs.size(); ++i) {
        runs[i].options = options;
        runs[i].final_sweep1site = final_sweep1site;
        runs[i].allow_small = false; // (the group's point is eight chains side by side: one engine launch per handle would serialise them)
        hs[i]->opt_begin(runs[i]);
    }
    static const bool prof = std::getenv("T4A_GROUP_PROF") != nullptr; // host time per phase, printed once per call
    double t_start = 0.0, t_launch = 0.0, t_finish = 0.0, t_fill = 0.0;
    const double wait0 = g_chain_wait_seconds;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    size_t iters = 0;
    for (;;) {
        std::vector<size_t> active;
        std::vector<Tci2*> chained;
        const auto ta = now();
        for (size_t i = 0; i < hs.size(); ++i)
            if (hs[i]->opt_iter_start(runs[i], true)) {
                active.push_back(i);
                if (runs[i].chained) chained.push_back(hs[i]);
            }
        if (active.empty()) break;
        const auto tb = now();
        chain_group_launch(chained); // one chain of launches for all of them (or one each when they do not line up)
        const auto tc = now();
        // the chained handles first, the group's leader first of all: it holds the chip until its chain has completed, and a
        // handle on the per-bond path needs an XCD of its own
        for (size_t i : active)
            if (runs[i].chained) hs[i]->opt_iter_issue_pending_fill(runs[i]); // (all of them while the group's chain still runs)
        t_fill += secs(tc, now());
        for (size_t i : active)
            if (runs[i].chained) hs[i]->opt_iter_finish(runs[i]);
        for (size_t i : active)
            if (!runs[i].chained) hs[i]->opt_iter_finish(runs[i]);
        const auto td = now();
        t_start += secs(ta, tb);
        t_launch += secs(tb, tc);
        t_finish += secs(tc, td);
        ++iters;
    }
    if (prof)
        std::fprintf(stderr, "[t4a] optimize_group: %zu handles, %zu iterations: start %.2f ms, launch %.2f ms, finish %.2f ms (of which issuing the previous iteration's fills %.2f ms, waiting for the device %.2f ms)\n",
                     hs.size(), iters, 1e3 * t_start, 1e3 * t_launch, 1e3 * t_finish, 1e3 * t_fill, 1e3 * (g_chain_wait_seconds - wait0));
    // the last iteration's fills of ALL handles first (each on its own fill stream), then the waits: handle by handle the device
    // idled while the host issued the next handle's dozen launches
    for (size_t i = 0; i < hs.size(); ++i) hs[i]->opt_end_issue_fill(runs[i]);
    for (size_t i = 0; i < hs.size(); ++i) hs[i]->opt_end(runs[i]);
    } catch (...) {
        for (Tci2* h : hs) h->chain_abort(); // (the leader's abort waits for the group's stream and gives the chip back)
        throw;
    }
}

// fill_site_tensors on several handles: issue all, then complete all (t4a_gpu_tci2_fill_site_tensors_group)
void Tci2::fill_site_tensors_group(const std::vector<Tci2*>& hs)
{
    for (Tci2* h : hs)
        if (!h) throw Error(T4A_GPU_NULL_POINTER, "fill_site_tensors_group: null handle");
    std::exception_ptr first_error;
    for (Tci2* h : hs) {
        try {
            h->fill_site_tensors_impl(true); // asynchronous for built-in functors (a host callback fills synchronously)
        } catch (...) {
            if (!first_error) first_error = std::current_exception();
        }
    }
    for (Tci2* h : hs) {
        try {
            h->fill_wait();
        } catch (...) {
            if (!first_error) first_error = std::current_exception();
        }
    }
    if (first_error) std::rethrow_exception(first_error);
}

// crossinterpolate2 (tensorci2.rs:1513-1563)
void Tci2::crossinterpolate2(std::vector<std::vector<uint32_t>> pivots, const TCI2Options& options)
{
    options.validate();
    require_fn();
    if (pivots.empty()) pivots.push_back(std::vector<uint32_t>(n_, 0));
    add_global_pivots(pivots);
    std::vector<uint32_t> flat;
    for (const auto& p : pivots) flat.insert(flat.end(), p.begin(), p.end());
    std::vector<double> vals = eval_points_host(flat, pivots.size());
    for (double v : vals) {
        const double a = std::sqrt(v * v);
        if (a > max_sample_value) max_sample_value = a;
    }
    if (max_sample_value < 1e-30) throw Error(T4A_GPU_INVALID_ARGUMENT, "Initial pivots have zero function values");
    optimize(options, true);
}

} // namespace t4a
