// patching.hip — see patching.hpp.  Host queue logic + one device TCI2 per patch.
#include "patching.hpp"
#include "stdrng.hpp"

#include <algorithm>
#include <cmath>
#include <deque>
#include <set>

namespace t4a {

namespace {

constexpr double ZERO_SAMPLE_THRESHOLD = 1.0e-30; // adaptive_interpolation.rs:25

// T[l, s, r] = delta(l, r) * delta(s, value)   (IdxTensor::from_copy_selector, adaptive_interpolation.rs:614-660)
__global__ void __launch_bounds__(256) copy_selector_kernel(double* out, int B, int S, int value, double scale)
{
    const size_t total = (size_t)B * S * B;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % B);
        const int s = (int)((e / B) % S);
        const int r = (int)(e / ((size_t)B * S));
        out[e] = (l == r && s == value) ? scale : 0.0;
    }
}

using Rng = StdRng; // adaptive_interpolation.rs:164: ONE StdRng::seed_from_u64 stream consumed in FIFO patch order (stdrng.hpp)

using Pivot = std::vector<uint32_t>;
using Projector = std::map<size_t, size_t>;

struct Ctx {
    const std::vector<size_t>* dims;
    const FullFunction* f;
    std::vector<size_t> offset; // per site into the weight tables
    size_t total = 0;
};

std::vector<size_t> active_positions(size_t n, const Projector& pr) // :361-367
{
    std::vector<size_t> a;
    for (size_t p = 0; p < n; ++p)
        if (!pr.count(p)) a.push_back(p);
    return a;
}

Pivot expand_pivot(const Pivot& local, const std::vector<size_t>& active, const Projector& pr, size_t n) // :470-486
{
    Pivot full(n, 0);
    for (size_t k = 0; k < active.size() && k < local.size(); ++k) full[active[k]] = local[k];
    for (const auto& kv : pr) full[kv.first] = (uint32_t)kv.second;
    return full;
}

// values of the full function at a list of full pivots (host side: candidate screening, single-site patches)
std::vector<double> eval_full(const Ctx& c, const std::vector<Pivot>& pts)
{
    const size_t n = c.dims->size();
    std::vector<double> out(pts.size());
    if (pts.empty()) return out;
    if (c.f->builtin) {
        for (size_t p = 0; p < pts.size(); ++p) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < c.f->n_acc; ++k) {
                const uint64_t* w = c.f->weights.data() + (size_t)k * c.total;
                for (size_t s = 0; s < n; ++s) acc[k] += w[c.offset[s] + pts[p][s]];
            }
            out[p] = t4a_fn_value(c.f->fid, acc, c.f->params);
        }
    } else {
        std::vector<uint32_t> flat(pts.size() * n);
        for (size_t p = 0; p < pts.size(); ++p) std::copy(pts[p].begin(), pts[p].end(), flat.begin() + p * n);
        const int64_t got = c.f->cb(c.f->ctx, flat.data(), n, pts.size(), out.data());
        if (got < 0 || (size_t)got != pts.size())
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned " + std::to_string(got) + " values for " +
                                                    std::to_string(pts.size()) + " requested entries");
    }
    return out;
}

bool is_compatible(const Pivot& pivot, size_t n, const Projector& pr) // :445-458
{
    if (pivot.size() != n) return false;
    for (const auto& kv : pr)
        if (pivot[kv.first] != kv.second) return false;
    return true;
}

std::vector<Pivot> patch_candidates(const std::vector<size_t>& dims, const std::vector<size_t>& active,
                                    const Projector& pr, const std::vector<Pivot>& initial,
                                    const std::vector<Pivot>& recycled, size_t target, Rng& rng) // :369-443
{
    std::vector<Pivot> cand;
    std::set<Pivot> seen;
    auto take = [&](const Pivot& full) {
        if (!is_compatible(full, dims.size(), pr)) return;
        Pivot local;
        for (size_t p : active) local.push_back(full[p]);
        if (seen.insert(local).second) cand.push_back(local);
    };
    for (const auto& v : initial) take(v);
    for (const auto& v : recycled) take(v);
    std::vector<size_t> ld;
    for (size_t p : active) ld.push_back(dims[p]);
    size_t point_count = 1;
    for (size_t d : ld) {
        if (d != 0 && point_count > std::numeric_limits<size_t>::max() / d)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "active patch point count exceeds usize");
        point_count *= d;
    }
    const size_t desired = std::min(std::max(target, cand.size()), point_count);
    if (desired > (std::numeric_limits<size_t>::max() - 100) / 20)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "initial-pivot search attempt count exceeds usize");
    const size_t attempts = desired * 20 + 100;
    for (size_t a = 0; a < attempts && cand.size() < desired; ++a) {
        Pivot pv;
        for (size_t d : ld) pv.push_back((uint32_t)rng.random_range(d));
        if (seen.insert(pv).second) cand.push_back(pv);
    }
    for (size_t flat = 0; flat < point_count && cand.size() < desired; ++flat) {
        Pivot pv;
        size_t f = flat;
        for (size_t d : ld) { // decode_col_major :460-468
            pv.push_back((uint32_t)(f % d));
            f /= d;
        }
        if (seen.insert(pv).second) cand.push_back(pv);
    }
    return cand;
}

void make_selector_core(DevCore& c, size_t bond, size_t dim, size_t value, double scale, hipStream_t st)
{
    c.l = bond;
    c.s = dim;
    c.r = bond;
    c.buf.reserve(std::max<size_t>(c.size(), 1));
    size_t blocks = (c.size() + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_selector_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c.buf.get(), (int)bond, (int)dim,
                       (int)value, scale);
}

void make_host_core(DevCore& c, size_t l, size_t s, size_t r, const std::vector<double>& data, hipStream_t st)
{
    c.l = l;
    c.s = s;
    c.r = r;
    c.buf.reserve(std::max<size_t>(c.size(), 1));
    if (c.size()) {
        T4A_HIP(hipMemcpyAsync(c.buf.get(), data.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st)); // `data` is pageable
    }
}

// rank_one_full_tt :662-709
SubDomain rank_one_patch(const std::vector<size_t>& dims, const Projector& pr, double scale, hipStream_t st)
{
    SubDomain sd;
    sd.projector = pr;
    sd.cores.resize(dims.size());
    for (size_t p = 0; p < dims.size(); ++p) {
        std::vector<double> data(dims[p], 0.0);
        const double ls = p == 0 ? scale : 1.0;
        auto it = pr.find(p);
        if (it != pr.end())
            data[it->second] = ls;
        else
            std::fill(data.begin(), data.end(), ls);
        make_host_core(sd.cores[p], 1, dims[p], 1, data, st);
    }
    return sd;
}

// embed_active_tt :514-612: active cores stay on the device, projected sites get copy-selector tensors
SubDomain embed_patch(const std::vector<const DevCore*>& active_cores, const std::vector<size_t>& dims,
                      const std::vector<size_t>& active, const Projector& pr, hipStream_t st)
{
    const size_t n = dims.size(), na = active.size();
    std::vector<size_t> link;
    for (size_t k = 0; k + 1 < na; ++k) link.push_back(active_cores[k]->r);
    std::vector<size_t> edge(n > 0 ? n - 1 : 0, 1);
    for (size_t e = 0; e + 1 < n; ++e) {
        size_t left = 0;
        for (size_t p : active)
            if (p <= e) ++left;
        edge[e] = (left == 0 || left == na) ? 1 : link[left - 1];
    }
    SubDomain sd;
    sd.projector = pr;
    sd.cores.resize(n);
    size_t next_active = 0;
    for (size_t p = 0; p < n; ++p) {
        const size_t l = p == 0 ? 1 : edge[p - 1], r = p + 1 == n ? 1 : edge[p];
        if (next_active < na && active[next_active] == p) {
            const DevCore& src = *active_cores[next_active++];
            if (src.l != l || src.r != r || src.s != dims[p])
                throw Error(T4A_GPU_INTERNAL_ERROR, "embedded core shape mismatch at site " + std::to_string(p));
            DevCore& c = sd.cores[p];
            c.l = l;
            c.s = dims[p];
            c.r = r;
            c.buf.reserve(std::max<size_t>(c.size(), 1));
            if (c.size())
                T4A_HIP(hipMemcpyAsync(c.buf.get(), src.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
        } else {
            auto it = pr.find(p);
            if (it == pr.end())
                throw Error(T4A_GPU_INVALID_ARGUMENT, "an embedded inactive site is missing from its projector");
            if (l != r) throw Error(T4A_GPU_INTERNAL_ERROR, "projected site requires equal carried bonds");
            make_selector_core(sd.cores[p], l, dims[p], it->second, 1.0, st);
        }
    }
    T4A_HIP(hipStreamSynchronize(st));
    return sd;
}

// global_diagonal_pivots :488-512
std::vector<Pivot> global_diagonal_pivots(const Tci2& tci, const std::vector<size_t>& active, const Projector& pr, size_t n)
{
    std::vector<Pivot> out;
    std::set<Pivot> seen;
    const_cast<Tci2&>(tci).sync_digits(); // (after a device-side bond chain the digit tables are decoded on demand)
    for (size_t b = 0; b + 1 < active.size(); ++b) {
        const IndexSet& is = tci.i_set[b + 1];
        const IndexSet& js = tci.j_set[b];
        const size_t cnt = std::min(is.count, js.count);
        for (size_t k = 0; k < cnt; ++k) {
            Pivot local(is.at(k), is.at(k) + is.width);
            local.insert(local.end(), js.at(k), js.at(k) + js.width);
            if (local.size() == active.size()) {
                Pivot full = expand_pivot(local, active, pr, n);
                if (seen.insert(full).second) out.push_back(full);
            }
        }
    }
    return out;
}

// host-callback trampoline: local pivots of a patch -> full pivots -> user callback
struct PatchCallback {
    const FullFunction* f;
    const std::vector<size_t>* active;
    Pivot base; // projected values at their positions, 0 elsewhere
    size_t n_full;
};
int64_t patch_trampoline(void* vctx, const uint32_t* idx, size_t n_local, size_t n_pts, double* out)
{
    const PatchCallback* pc = static_cast<const PatchCallback*>(vctx);
    std::vector<uint32_t> full(n_pts * pc->n_full);
    for (size_t p = 0; p < n_pts; ++p) {
        uint32_t* dst = full.data() + p * pc->n_full;
        std::copy(pc->base.begin(), pc->base.end(), dst);
        for (size_t k = 0; k < n_local; ++k) dst[(*pc->active)[k]] = idx[p * n_local + k];
    }
    return pc->f->cb(pc->f->ctx, full.data(), pc->n_full, n_pts, out);
}

std::vector<size_t> validate(const std::vector<size_t>& dims, const std::vector<Pivot>& pivots, const AdaptiveOptions& o) // :264-355
{
    auto bad = [](const char* m) { throw Error(T4A_GPU_INVALID_ARGUMENT, m); };
    if (dims.empty()) bad("site_indices must not be empty");
    for (size_t d : dims)
        if (d == 0) bad("site indices must have positive dimensions");
    if (o.n_initial_pivots == 0) bad("n_initial_pivots must be positive");
    if (!std::isfinite(o.tci.tolerance) || o.tci.tolerance < 0.0) bad("TCI tolerance must be finite and nonnegative");
    if (o.tci.max_iter == 0) bad("TCI max_iter must be positive");
    if (o.tci.ncheck_history == 0) bad("TCI ncheck_history must be positive");
    if (!std::isfinite(o.tci.tol_margin_global_search) || o.tci.tol_margin_global_search < 0.0)
        bad("TCI tol_margin_global_search must be finite and nonnegative");
    for (const auto& p : pivots) {
        if (p.size() != dims.size()) bad("every initial pivot must have one coordinate per site");
        for (size_t s = 0; s < p.size(); ++s)
            if (p[s] >= dims[s]) bad("an initial pivot coordinate is outside its site dimension");
    }
    std::vector<size_t> order = o.patch_order;
    if (order.empty())
        for (size_t p = 0; p < dims.size(); ++p) order.push_back(p);
    std::set<size_t> uniq(order.begin(), order.end());
    if (order.size() != dims.size() || uniq.size() != order.size() || *uniq.rbegin() >= dims.size())
        bad("patch_order must be an exact permutation of site_indices");
    return order;
}

} // namespace

std::unique_ptr<PartitionedTT> adaptive_interpolate(const std::vector<size_t>& dims, const FullFunction& f,
                                                    const std::vector<std::vector<uint32_t>>& initial_pivots,
                                                    const AdaptiveOptions& options)
{
    const std::vector<size_t> patch_order = validate(dims, initial_pivots, options);
    const size_t n = dims.size();
    if (!f.builtin && !f.cb) throw Error(T4A_GPU_NULL_POINTER, "no function given");
    Ctx ctx;
    ctx.dims = &dims;
    ctx.f = &f;
    ctx.offset.resize(n);
    for (size_t s = 0; s < n; ++s) {
        ctx.offset[s] = ctx.total;
        ctx.total += dims[s];
    }
    if (f.builtin && f.weights.size() != (size_t)f.n_acc * ctx.total)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "built-in function weight table has the wrong size");
    std::unique_ptr<PartitionedTT> result(new PartitionedTT(dims));
    hipStream_t st = result->eng.stream();
    Rng rng(options.tci.has_seed ? options.tci.seed : 0);

    struct Pending {
        Projector projector;
        std::vector<Pivot> recycled;
    };
    std::deque<Pending> pending;
    pending.push_back({});
    while (!pending.empty()) {
        Pending patch = std::move(pending.front());
        pending.pop_front();
        const std::vector<size_t> active = active_positions(n, patch.projector);
        if (active.empty()) { // :75-85
            const double v = eval_full(ctx, {expand_pivot({}, active, patch.projector, n)})[0];
            result->patches.push_back(rank_one_patch(dims, patch.projector, v, st));
            continue;
        }
        if (active.size() == 1) { // :87-116: exact single-site patch
            const size_t d = dims[active[0]];
            std::vector<Pivot> pts;
            for (size_t s = 0; s < d; ++s) pts.push_back(expand_pivot({(uint32_t)s}, active, patch.projector, n));
            const std::vector<double> vals = eval_full(ctx, pts);
            DevCore core;
            make_host_core(core, 1, d, 1, vals, st);
            result->patches.push_back(embed_patch({&core}, dims, active, patch.projector, st));
            continue;
        }
        const std::vector<Pivot> cand = patch_candidates(dims, active, patch.projector, initial_pivots, patch.recycled,
                                                         options.n_initial_pivots, rng);
        {
            std::vector<Pivot> full;
            for (const auto& c : cand) full.push_back(expand_pivot(c, active, patch.projector, n));
            const std::vector<double> vals = eval_full(ctx, full);
            bool all_zero = true;
            for (double v : vals)
                if (!(std::fabs(v) < ZERO_SAMPLE_THRESHOLD)) all_zero = false;
            if (all_zero) { // :141-149
                result->patches.push_back(rank_one_patch(dims, patch.projector, 0.0, st));
                continue;
            }
        }
        std::vector<size_t> local_dims;
        for (size_t p : active) local_dims.push_back(dims[p]);
        Tci2 tci(local_dims);
        PatchCallback pcb;
        if (f.builtin) {
            // restrict the integer weight tables to the active sites; the projected sites contribute a constant that is
            // folded into every entry of the first active site (the accumulators are plain wrapping sums)
            size_t ltotal = 0;
            for (size_t d : local_dims) ltotal += d;
            std::vector<uint64_t> w((size_t)f.n_acc * ltotal);
            for (int k = 0; k < f.n_acc; ++k) {
                const uint64_t* src = f.weights.data() + (size_t)k * ctx.total;
                uint64_t constant = 0;
                for (const auto& kv : patch.projector) constant += src[ctx.offset[kv.first] + kv.second];
                size_t lo = 0;
                for (size_t a = 0; a < active.size(); ++a) {
                    for (size_t v = 0; v < local_dims[a]; ++v)
                        w[(size_t)k * ltotal + lo + v] = src[ctx.offset[active[a]] + v] + (a == 0 ? constant : 0);
                    lo += local_dims[a];
                }
            }
            tci.set_builtin(f.fid, f.n_acc, f.params, w.data());
        } else {
            pcb.f = &f;
            pcb.active = &active;
            pcb.base = expand_pivot({}, active, patch.projector, n);
            pcb.n_full = n;
            tci.set_callback(&patch_trampoline, &pcb);
        }
        tci.crossinterpolate2(cand, options.tci);
        const double normalization = (options.tci.normalize_error && tci.max_sample_value > 0.0) ? tci.max_sample_value : 1.0;
        const double final_error = tci.errors_hist.empty() ? tci.max_bond_error() / normalization : tci.errors_hist.back();
        if (tci.termination == T4A_GPU_TCI2_CONVERGED && final_error <= options.tci.tolerance) { // :357-359
            tci.fill_wait();
            std::vector<const DevCore*> ac;
            for (size_t a = 0; a < active.size(); ++a) ac.push_back(&tci.cores[a]);
            T4A_HIP(hipStreamSynchronize(tci.eng.stream()));
            result->patches.push_back(embed_patch(ac, dims, active, patch.projector, st));
            continue;
        }
        size_t split = n;
        for (size_t p : patch_order)
            if (!patch.projector.count(p)) {
                split = p;
                break;
            }
        if (split == n) throw Error(T4A_GPU_INVALID_ARGUMENT, "a nonconverged patch has no remaining split index");
        std::vector<Pivot> recycled;
        if (options.recycle_pivots) recycled = global_diagonal_pivots(tci, active, patch.projector, n);
        for (size_t v = 0; v < dims[split]; ++v) {
            Pending child;
            child.projector = patch.projector;
            child.projector[split] = v;
            child.recycled = recycled;
            pending.push_back(std::move(child));
        }
    }
    T4A_HIP(hipGetLastError());
    return result;
}

std::vector<double> PartitionedTT::evaluate(const uint32_t* idx, size_t n_pts)
{
    const size_t n = dims.size();
    std::vector<double> out(n_pts, 0.0);
    if (n_pts == 0) return out;
    for (size_t p = 0; p < n_pts; ++p)
        for (size_t s = 0; s < n; ++s)
            if (idx[p * n + s] >= dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate: index out of bounds");
    hipStream_t st = eng.stream();
    d_idx_.reserve(n_pts * n);
    d_vals_.reserve(n_pts);
    d_desc_.reserve(n);
    T4A_HIP(hipMemcpyAsync(d_idx_.get(), idx, n_pts * n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    std::vector<double> part(n_pts);
    std::vector<TtCoreDesc> desc(n);
    for (const SubDomain& sd : patches) {
        int max_bond = 1;
        for (size_t s = 0; s < n; ++s) {
            desc[s].data = sd.cores[s].buf.get();
            desc[s].l = (int)sd.cores[s].l;
            desc[s].d = (int)sd.cores[s].s;
            desc[s].r = (int)sd.cores[s].r;
            max_bond = std::max(max_bond, std::max(desc[s].l, desc[s].r));
        }
        T4A_HIP(hipMemcpyAsync(d_desc_.get(), desc.data(), n * sizeof(TtCoreDesc), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
        tt_evaluate_launch(d_desc_.get(), (int)n, max_bond, d_idx_.get(), (int)n_pts, d_vals_.get(), st);
        T4A_HIP(hipMemcpyAsync(part.data(), d_vals_.get(), n_pts * sizeof(double), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
        for (size_t p = 0; p < n_pts; ++p) out[p] = out[p] + part[p]; // patch order, like the oracle
    }
    T4A_HIP(hipGetLastError());
    return out;
}

} // namespace t4a
