// quantics.hip — quantics front end on the TreeTCI driver (see quantics.hpp).
#include "quantics.hpp"
#include "stdrng.hpp"

#include <algorithm>
#include <array>
#include <cmath>

namespace t4a {

QuanticsGrid::QuanticsGrid(const std::vector<size_t>& rs_, Unfolding u, bool disc, const std::vector<double>& lo,
                           const std::vector<double>& up, bool endpoint)
    : rs(rs_), lower(lo), upper(up), include_endpoint(endpoint), unfolding(u), discretized(disc)
{
    if (rs.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "a grid needs at least one variable");
    for (size_t r : rs)
        if (r == 0 || r > 62) throw Error(T4A_GPU_INVALID_ARGUMENT, "bits per variable must be in 1..62");
    if (lower.empty()) lower.assign(rs.size(), 0.0);
    if (upper.empty()) upper.assign(rs.size(), 1.0);
    if (lower.size() != rs.size() || upper.size() != rs.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, "bound length mismatch");
    if (disc)
        for (size_t d = 0; d < rs.size(); ++d)
            if (!(lower[d] < upper[d])) throw Error(T4A_GPU_INVALID_ARGUMENT, "lower bound must be below the upper bound");
    size_t max_r = 0;
    for (size_t r : rs) max_r = std::max(max_r, r);
    for (size_t level = 0; level < max_r; ++level) {
        std::vector<std::pair<size_t, size_t>> fused;
        for (size_t d = 0; d < rs.size(); ++d)
            if (level < rs[d]) {
                if (u == Unfolding::Interleaved)
                    sites.push_back({{d, level}});
                else
                    fused.push_back({d, level});
            }
        if (u == Unfolding::Fused) {
            if (fused.size() > 31) throw Error(T4A_GPU_INVALID_ARGUMENT, "too many variables fused into one site");
            sites.push_back(fused);
        }
    }
}

std::vector<size_t> QuanticsGrid::local_dimensions() const
{
    std::vector<size_t> d;
    for (const auto& s : sites) d.push_back((size_t)1 << s.size());
    return d;
}

std::vector<double> QuanticsGrid::grid_step() const
{
    std::vector<double> st(rs.size());
    for (size_t d = 0; d < rs.size(); ++d) {
        const double npts = (double)((uint64_t)1 << rs[d]);
        st[d] = include_endpoint ? (upper[d] - lower[d]) / (npts - 1.0) : (upper[d] - lower[d]) / npts;
    }
    return st;
}

void QuanticsGrid::grididx_to_quantics(const size_t* g, uint32_t* q) const
{
    for (size_t d = 0; d < rs.size(); ++d)
        if (g[d] >= ((size_t)1 << rs[d]))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Grid index " + std::to_string(g[d]) + " out of range for variable " + std::to_string(d));
    for (size_t s = 0; s < sites.size(); ++s) {
        uint32_t v = 0, p = 1;
        for (const auto& dl : sites[s]) {
            v += (uint32_t)((g[dl.first] >> (rs[dl.first] - 1 - dl.second)) & 1) * p;
            p *= 2;
        }
        q[s] = v;
    }
}

void QuanticsGrid::quantics_to_grididx(const uint32_t* q, size_t* g) const
{
    for (size_t d = 0; d < rs.size(); ++d) g[d] = 0;
    for (size_t s = 0; s < sites.size(); ++s) {
        if (q[s] >= ((uint32_t)1 << sites[s].size())) throw Error(T4A_GPU_INVALID_ARGUMENT, "quantics digit out of range");
        size_t v = q[s];
        for (const auto& dl : sites[s]) {
            g[dl.first] |= (v & 1) << (rs[dl.first] - 1 - dl.second);
            v >>= 1;
        }
    }
}

void QuanticsGrid::quantics_to_origcoord(const uint32_t* q, double* x) const
{
    std::vector<size_t> g(rs.size());
    quantics_to_grididx(q, g.data());
    const auto st = grid_step();
    for (size_t d = 0; d < rs.size(); ++d) x[d] = lower[d] + (double)g[d] * st[d];
}

TreeTciOptions QtciOptions::to_treetci_options() const
{
    TreeTciOptions o;
    o.tolerance = tolerance;
    o.max_iter = max_iter;
    o.max_bond_dim = max_bond_dim;
    o.normalize_error = normalize_error;
    o.enable_global_pivots = false;
    o.nsearch = 0;
    o.max_nglobal_pivot = 0;
    o.tol_margin_global_search = 10.0;
    return o;
}

void qtci_check_sizes(const std::vector<size_t>& sizes)
{
    std::vector<double> dims;
    for (size_t s : sizes) dims.push_back(std::log2((double)s));
    for (size_t k = 0; k + 1 < dims.size(); ++k)
        if (!(std::fabs(dims[k] - dims[k + 1]) < 1e-10))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "this method only supports grids with equal number of points in each direction");
    for (double d : dims)
        if (!(std::fabs(d - std::round(d)) < 1e-10))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "this method only supports grid sizes that are powers of 2");
}

bool qtci_check_xvals_uniform(const std::vector<std::vector<double>>& xvals)
{
    if (xvals.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "xvals must not be empty");
    for (const auto& x : xvals)
        if (x.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "xvals must not contain empty dimensions");
    for (size_t d = 0; d < xvals.size(); ++d) {
        for (double v : xvals[d])
            if (!std::isfinite(v)) throw Error(T4A_GPU_INVALID_ARGUMENT, "xvals[" + std::to_string(d) + "] must contain only finite values");
        for (size_t k = 0; k + 1 < xvals[d].size(); ++k)
            if (xvals[d][k] >= xvals[d][k + 1])
                throw Error(T4A_GPU_INVALID_ARGUMENT, "xvals[" + std::to_string(d) + "] must be strictly increasing without duplicates");
    }
    std::vector<size_t> sizes;
    for (const auto& x : xvals) sizes.push_back(x.size());
    qtci_check_sizes(sizes);
    bool uniform = true;
    for (const auto& x : xvals) {
        if (x.size() < 2) continue;
        const double step = x[1] - x[0];
        for (size_t k = 0; k + 1 < x.size(); ++k)
            if (!(std::fabs(x[k + 1] - x[k] - step) <= 1e-12)) uniform = false;
    }
    return uniform;
}

QuanticsTci::QuanticsTci(const QuanticsGrid& g, t4a_gpu_coord_eval_fn coord_cb, t4a_gpu_grididx_eval_fn grididx_cb, void* ctx,
                         std::vector<std::vector<double>> xvals)
    : grid(g), coord_cb_(coord_cb), grididx_cb_(grididx_cb), ctx_(ctx), xvals_(std::move(xvals))
{
    if ((coord_cb_ == nullptr) == (grididx_cb_ == nullptr)) throw Error(T4A_GPU_NULL_POINTER, "exactly one evaluation callback is required");
}

int64_t QuanticsTci::trampoline(void* ctx, const uint32_t* idx, size_t n_sites, size_t n_pts, double* out)
{
    try {
        return static_cast<QuanticsTci*>(ctx)->eval_batch(idx, n_sites, n_pts, out);
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return -1;
    }
}

// the batch adapter of quantics_tci.rs:224-232 with the per-point cache of :197-221; all cache misses of one batch go to
// the user in a single call
int64_t QuanticsTci::eval_batch(const uint32_t* idx, size_t n_sites, size_t n_pts, double* out)
{
    const size_t nv = grid.n_vars();
    std::vector<size_t> miss_point;
    std::vector<std::vector<uint32_t>> miss_key;
    std::unordered_map<std::vector<uint32_t>, size_t, KeyHash> pending; // key -> slot among the misses
    std::vector<long> slot(n_pts, -1);
    std::vector<uint32_t> key(n_sites);
    for (size_t p = 0; p < n_pts; ++p) {
        key.assign(idx + p * n_sites, idx + (p + 1) * n_sites);
        auto it = cache.find(key);
        if (it != cache.end()) {
            out[p] = it->second;
            continue;
        }
        auto ins = pending.emplace(key, miss_key.size());
        if (ins.second) miss_key.push_back(key);
        slot[p] = (long)ins.first->second;
    }
    const size_t n_miss = miss_key.size();
    if (n_miss) {
        std::vector<double> vals(n_miss);
        int64_t got;
        if (grididx_cb_) {
            std::vector<size_t> g(nv * n_miss);
            for (size_t m = 0; m < n_miss; ++m) grid.quantics_to_grididx(miss_key[m].data(), g.data() + m * nv);
            got = grididx_cb_(ctx_, g.data(), nv, n_miss, vals.data());
        } else {
            std::vector<double> x(nv * n_miss);
            if (xvals_.empty()) {
                for (size_t m = 0; m < n_miss; ++m) grid.quantics_to_origcoord(miss_key[m].data(), x.data() + m * nv);
            } else {
                std::vector<size_t> g(nv);
                for (size_t m = 0; m < n_miss; ++m) {
                    grid.quantics_to_grididx(miss_key[m].data(), g.data());
                    for (size_t d = 0; d < nv; ++d) x[m * nv + d] = xvals_[d][g[d]];
                }
            }
            got = coord_cb_(ctx_, x.data(), nv, n_miss, vals.data());
        }
        ++n_user_calls;
        n_user_points += n_miss;
        if (got < 0 || (size_t)got != n_miss) return -1;
        for (size_t m = 0; m < n_miss; ++m) cache.emplace(miss_key[m], vals[m]);
        for (size_t p = 0; p < n_pts; ++p)
            if (slot[p] >= 0) out[p] = vals[(size_t)slot[p]];
    }
    return (int64_t)n_pts;
}

void QuanticsTci::run(const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options)
{
    const std::vector<size_t> local_dims = grid.local_dimensions();
    const size_t n_sites = local_dims.size();
    std::vector<std::vector<uint32_t>> pivots;
    if (initial_pivots) {
        for (const auto& g : *initial_pivots) {
            if (g.size() != grid.n_vars()) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial pivot length must match the number of variables");
            std::vector<uint32_t> q(n_sites);
            try {
                grid.grididx_to_quantics(g.data(), q.data());
            } catch (const Error& e) {
                std::string s = "initial pivot [";
                for (size_t d = 0; d < g.size(); ++d) s += (d ? ", " : "") + std::to_string(g[d]);
                throw Error(T4A_GPU_INVALID_ARGUMENT, s + "] conversion failed: " + e.what());
            }
            pivots.push_back(q);
        }
    } else {
        pivots.push_back(std::vector<uint32_t>(n_sites, 0));
    }
    StdRng rng(options.has_seed ? options.seed : 0x13198A2E03707344ull); // (quantics_tci.rs:464: the thread rng there; any stream will do)
    for (size_t k = 0; k < options.n_random_init_pivot; ++k) {
        std::vector<uint32_t> p(n_sites);
        for (size_t s = 0; s < n_sites; ++s) p[s] = (uint32_t)rng.random_range(local_dims[s]);
        pivots.push_back(p);
    }
    if (pivots.empty()) pivots.push_back(std::vector<uint32_t>(n_sites, 0));

    std::vector<TreeEdge> edges;
    for (size_t k = 0; k + 1 < n_sites; ++k) edges.emplace_back(k, k + 1);
    if (n_sites < 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
    tci.reset(new TreeTci(local_dims, TreeGraph(n_sites, edges)));
    tci->set_callback(&QuanticsTci::trampoline, this);
    tci->add_global_pivots(pivots);
    // max_sample_value from the initial pivots (:268-278)
    std::vector<uint32_t> flat;
    for (const auto& p : pivots) flat.insert(flat.end(), p.begin(), p.end());
    std::vector<double> init(pivots.size());
    if (eval_batch(flat.data(), n_sites, pivots.size(), init.data()) != (int64_t)pivots.size())
        throw Error(T4A_GPU_CALLBACK_ERROR, "the evaluation callback returned a wrong number of values");
    double m = 0.0;
    for (double v : init) m = std::max(m, std::sqrt(v * v));
    tci->max_sample_value = m;
    if (tci->max_sample_value <= 0.0) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial pivots must not all evaluate to zero");
    tci->optimize(options.to_treetci_options());
    tci->materialize(0);
    const std::vector<DevCore> cores = tci->chain_cores();
    tt.reset(new TensorTrain(cores, tci->eng.stream()));
}

std::vector<double> QuanticsTci::evaluate(const size_t* grididx, size_t n_pts)
{
    if (!tt) throw Error(T4A_GPU_INVALID_ARGUMENT, "no interpolation has been run");
    const size_t nv = grid.n_vars(), ns = grid.n_sites();
    std::vector<uint32_t> q(ns * n_pts);
    for (size_t p = 0; p < n_pts; ++p) grid.grididx_to_quantics(grididx + p * nv, q.data() + p * ns);
    return tt->evaluate(q.data(), n_pts);
}

double QuanticsTci::sum()
{
    if (!tt) throw Error(T4A_GPU_INVALID_ARGUMENT, "no interpolation has been run");
    return tt->sum();
}

double QuanticsTci::integral() // :130-141
{
    const double s = sum();
    if (!grid.discretized) return s;
    double step = 1.0;
    for (double v : grid.grid_step()) step = step * v;
    return s * step;
}

// ------------------------------------------------------------------------------------------------ batched/mod.rs
std::unique_ptr<TensorTrain> combine_component_tts(std::vector<std::unique_ptr<TensorTrain>>& comps) // :193-318
{
    if (comps.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "no component tensor trains to combine");
    const size_t n = comps[0]->len();
    if (n == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "component tensor trains must have at least one site");
    for (size_t c = 0; c < comps.size(); ++c) {
        if (comps[c]->len() != n)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "component " + std::to_string(c) + " has " + std::to_string(comps[c]->len()) +
                                                      " sites, expected " + std::to_string(n));
        for (size_t s = 0; s < n; ++s)
            if (comps[c]->cores[s].s != comps[0]->cores[s].s)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "component " + std::to_string(c) + " site " + std::to_string(s) + " has a different site_dim");
    }
    // pure data movement: the blocks are placed on the host and the combined train is uploaded once
    std::vector<std::array<size_t, 3>> dims3;
    std::vector<double> data;
    for (size_t s = 0; s < n; ++s) {
        size_t total_l = 0, total_r = 0;
        for (auto& tt : comps) {
            total_l += tt->cores[s].l;
            total_r += tt->cores[s].r;
        }
        const size_t L = s == 0 ? 1 : total_l, S = comps[0]->cores[s].s, R = total_r;
        const size_t base = data.size();
        data.resize(base + L * S * R, 0.0);
        size_t lo = 0, ro = 0;
        for (auto& tt : comps) {
            const DevCore& c = tt->cores[s];
            const std::vector<double> h = tt->site_tensor_host(s);
            for (size_t r = 0; r < c.r; ++r)
                for (size_t x = 0; x < S; ++x)
                    for (size_t l = 0; l < c.l; ++l)
                        data[base + ((s == 0 ? 0 : lo) + l) + L * (x + S * (ro + r))] = h[l + c.l * (x + S * r)];
            lo += c.l;
            ro += c.r;
        }
        dims3.push_back({L, S, R});
    }
    size_t total_r = 0;
    for (auto& tt : comps) total_r += tt->cores[n - 1].r;
    const size_t base = data.size();
    data.resize(base + total_r * comps.size(), 0.0);
    size_t off = 0;
    for (size_t c = 0; c < comps.size(); ++c) {
        for (size_t i = 0; i < comps[c]->cores[n - 1].r; ++i) data[base + (off + i) + total_r * c] = 1.0;
        off += comps[c]->cores[n - 1].r;
    }
    dims3.push_back({total_r, comps.size(), 1});
    return std::unique_ptr<TensorTrain>(new TensorTrain(dims3, data.data()));
}

namespace {
struct BatchedShared {
    t4a_gpu_coord_eval_vec_fn f;
    void* ctx;
    size_t n_comp;
    size_t comp;
    struct KeyHash {
        size_t operator()(const std::vector<uint64_t>& v) const
        {
            uint64_t h = 0xcbf29ce484222325ull;
            for (uint64_t x : v) h = (h ^ x) * 0x100000001b3ull;
            return (size_t)h;
        }
    };
    std::unordered_map<std::vector<uint64_t>, std::vector<double>, KeyHash> cache; // coordinate bits -> all components
    size_t n_user_calls = 0, n_user_points = 0;
    std::string short_result;
};

int64_t batched_component_cb(void* vctx, const double* coords, size_t n_vars, size_t n_pts, double* out)
{
    BatchedShared& sh = *static_cast<BatchedShared*>(vctx);
    std::vector<size_t> miss;
    std::vector<std::vector<uint64_t>> keys(n_pts, std::vector<uint64_t>(n_vars));
    for (size_t p = 0; p < n_pts; ++p) {
        std::memcpy(keys[p].data(), coords + p * n_vars, n_vars * sizeof(double));
        if (!sh.cache.count(keys[p])) {
            sh.cache.emplace(keys[p], std::vector<double>()); // placeholder: identical coordinates inside one batch
            miss.push_back(p);
        }
    }
    if (!miss.empty()) {
        std::vector<double> x(miss.size() * n_vars), vals(miss.size() * sh.n_comp);
        for (size_t m = 0; m < miss.size(); ++m) std::memcpy(x.data() + m * n_vars, coords + miss[m] * n_vars, n_vars * sizeof(double));
        const int64_t got = sh.f(sh.ctx, x.data(), n_vars, miss.size(), sh.n_comp, vals.data());
        ++sh.n_user_calls;
        sh.n_user_points += miss.size();
        if (got < 0 || (size_t)got != miss.size() * sh.n_comp) {
            const size_t per_point = got < 0 ? 0 : (size_t)got / std::max<size_t>(miss.size(), 1);
            sh.short_result = "callback returned " + std::to_string(per_point) + " components, expected at least " +
                              std::to_string(sh.n_comp);
            return -1;
        }
        for (size_t m = 0; m < miss.size(); ++m)
            sh.cache[keys[miss[m]]].assign(vals.begin() + m * sh.n_comp, vals.begin() + (m + 1) * sh.n_comp);
    }
    for (size_t p = 0; p < n_pts; ++p) out[p] = sh.cache[keys[p]][sh.comp];
    return (int64_t)n_pts;
}
} // namespace

QuanticsBatchedResult quantics_batched(const QuanticsGrid& grid, t4a_gpu_coord_eval_vec_fn f, void* ctx,
                                       const std::vector<size_t>& output_dims,
                                       const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options)
{
    if (output_dims.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "output_dims must not be empty");
    size_t n_comp = 1;
    for (size_t d : output_dims) {
        if (d != 0 && n_comp > (size_t)-1 / d) throw Error(T4A_GPU_INVALID_ARGUMENT, "product of output_dims overflowed usize");
        n_comp *= d;
    }
    if (n_comp == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "product of output_dims must be positive, got 0");
    if (!grid.discretized) throw Error(T4A_GPU_INVALID_ARGUMENT, "a discretized grid is required");
    BatchedShared sh;
    sh.f = f;
    sh.ctx = ctx;
    sh.n_comp = n_comp;
    QuanticsBatchedResult res;
    res.output_dims = output_dims;
    std::vector<std::unique_ptr<TensorTrain>> comps;
    for (size_t comp = 0; comp < n_comp; ++comp) {
        sh.comp = comp;
        QuanticsTci q(grid, &batched_component_cb, nullptr, &sh, {});
        try {
            q.run(initial_pivots, options);
        } catch (const Error&) {
            if (!sh.short_result.empty()) throw Error(T4A_GPU_CALLBACK_ERROR, sh.short_result);
            throw;
        }
        const auto& t = *q.tci;
        if (res.ranks.size() < t.ranks_hist.size()) res.ranks.resize(t.ranks_hist.size(), 0);
        if (res.errors.size() < t.errors_hist.size()) res.errors.resize(t.errors_hist.size(), 0.0);
        for (size_t k = 0; k < t.ranks_hist.size(); ++k) res.ranks[k] = std::max(res.ranks[k], t.ranks_hist[k]);
        for (size_t k = 0; k < t.errors_hist.size(); ++k) res.errors[k] = std::max(res.errors[k], t.errors_hist[k]);
        comps.push_back(std::move(q.tt));
    }
    res.tt = combine_component_tts(comps);
    res.n_user_calls = sh.n_user_calls;
    res.n_user_points = sh.n_user_points;
    return res;
}

} // namespace t4a
