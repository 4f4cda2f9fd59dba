// oracle/t4a_oracle_quantics.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp).
// CPU restatement of the quantics front end, crates/tensor4all-quanticstci/src:
//   options.rs       QtciOptions :9-45, to_treetci_options :83-97 (global pivot search off, nsearch 0)
//   quantics_tci.rs  QuanticsTensorCI2 :53-173 (evaluate :118, sum :126, integral :130-141, cachedata :152),
//                    quanticscrossinterpolate :175-307 (memoised grid-point evaluation :197-221, batch adapter :224-232,
//                    initial pivots :235-253, TreeTCI on TreeTciGraph::linear_chain :256-283, to_treetn(center 0) + bridge to
//                    SimpleTensorTrain :284-292), quanticscrossinterpolate_from_arrays :309-432 (validation, uniform grids ->
//                    DiscretizedGrid with include_endpoint, otherwise coordinate lookup on an inherent grid),
//                    quanticscrossinterpolate_discrete :434-560.
//   batched/mod.rs   quanticscrossinterpolate_batched :50-191 (one scalar run per output component over a shared
//                    coordinate-keyed cache), combine_component_tts :193-318 (block-diagonal direct sum + selector site).
// The grid itself lives in the un-vendored crate `quanticsgrids` @ git rev 8214b72 (Cargo.toml:83).  Its published
// algorithm (QuanticsGrids.jl) is restated here: R_d bits per variable, most significant bit first; unfolding
// `Interleaved` = one binary site per (bit level, variable), levels outermost; `Fused` = one site per bit level whose
// value is sum_d bit_d * 2^d with the FIRST variable least significant; grid indices are 0-based; a discretized grid
// maps index g to lower + g * step with step = (upper - lower) / 2^R, or (upper - lower) / (2^R - 1) when the end point
// is included.  Pinned only through the quanticstci tests that observe it (tests/test_oracle_quantics.py): "parity
// unpinned" for anything those tests do not see (e.g. unequal R_d, MSB-aligned here).  The random initial pivots come
// from rand::rng() in the reference (non-reproducible by construction); splitmix64 here.
#pragma once

#include <cstring>
#include <memory>

#include "t4a_oracle_tree.hpp"

namespace t4a_oracle {

enum class Unfolding { Interleaved = 0, Fused = 1 };

struct QuanticsGrid {
    std::vector<size_t> rs;
    std::vector<double> lower, upper;
    bool include_endpoint = false;
    Unfolding unfolding = Unfolding::Interleaved;
    bool discretized = true; // false: InherentDiscreteGrid (integer coordinates)
    // site table: for every site the (variable, bit level) pairs it carries, first variable least significant
    std::vector<std::vector<std::pair<size_t, size_t>>> sites;

    QuanticsGrid() = default;
    QuanticsGrid(const std::vector<size_t>& rs_, Unfolding u, bool disc, const std::vector<double>& lo = {},
                 const std::vector<double>& up = {}, bool endpoint = false)
        : rs(rs_), lower(lo), upper(up), include_endpoint(endpoint), unfolding(u), discretized(disc)
    {
        if (rs.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "a grid needs at least one variable");
        for (size_t r : rs)
            if (r == 0 || r > 62) throw OracleError(ERR_INVALID_ARGUMENT, "bits per variable must be in 1..62");
        if (lower.empty()) lower.assign(rs.size(), 0.0);
        if (upper.empty()) upper.assign(rs.size(), 1.0);
        if (lower.size() != rs.size() || upper.size() != rs.size()) throw OracleError(ERR_INVALID_ARGUMENT, "bound length mismatch");
        if (disc)
            for (size_t d = 0; d < rs.size(); ++d)
                if (!(lower[d] < upper[d])) throw OracleError(ERR_INVALID_ARGUMENT, "lower bound must be below the upper bound");
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
            if (u == Unfolding::Fused) sites.push_back(fused);
        }
    }
    size_t n_vars() const { return rs.size(); }
    std::vector<size_t> local_dimensions() const
    {
        std::vector<size_t> d;
        for (const auto& s : sites) d.push_back((size_t)1 << s.size());
        return d;
    }
    std::vector<double> grid_step() const
    {
        std::vector<double> st(rs.size());
        for (size_t d = 0; d < rs.size(); ++d) {
            const double npts = (double)((uint64_t)1 << rs[d]);
            st[d] = include_endpoint ? (upper[d] - lower[d]) / (npts - 1.0) : (upper[d] - lower[d]) / npts;
        }
        return st;
    }
    MultiIndex grididx_to_quantics(const std::vector<size_t>& g) const
    {
        if (g.size() != rs.size()) throw OracleError(ERR_INVALID_ARGUMENT, "grid index length mismatch");
        for (size_t d = 0; d < rs.size(); ++d)
            if (g[d] >= ((size_t)1 << rs[d]))
                throw OracleError(ERR_INVALID_ARGUMENT, "Grid index " + std::to_string(g[d]) + " out of range for variable " + std::to_string(d));
        MultiIndex q(sites.size(), 0);
        for (size_t s = 0; s < sites.size(); ++s) {
            size_t v = 0, p = 1;
            for (const auto& dl : sites[s]) {
                const size_t bit = (g[dl.first] >> (rs[dl.first] - 1 - dl.second)) & 1;
                v += bit * p;
                p *= 2;
            }
            q[s] = v;
        }
        return q;
    }
    std::vector<size_t> quantics_to_grididx(const MultiIndex& q) const
    {
        if (q.size() != sites.size()) throw OracleError(ERR_INVALID_ARGUMENT, "quantics index length mismatch");
        std::vector<size_t> g(rs.size(), 0);
        for (size_t s = 0; s < sites.size(); ++s) {
            if (q[s] >= ((size_t)1 << sites[s].size())) throw OracleError(ERR_INVALID_ARGUMENT, "quantics digit out of range");
            size_t v = q[s];
            for (const auto& dl : sites[s]) {
                g[dl.first] |= (v & 1) << (rs[dl.first] - 1 - dl.second);
                v >>= 1;
            }
        }
        return g;
    }
    std::vector<double> quantics_to_origcoord(const MultiIndex& q) const
    {
        const auto g = quantics_to_grididx(q);
        const auto st = grid_step();
        std::vector<double> x(rs.size());
        for (size_t d = 0; d < rs.size(); ++d) x[d] = lower[d] + (double)g[d] * st[d];
        return x;
    }
};

struct QtciOptions { // options.rs:9-45
    double tolerance = 1e-8;
    size_t max_bond_dim = 0; // 0 == None
    size_t max_iter = 200;
    size_t n_random_init_pivot = 5;
    Unfolding unfolding = Unfolding::Interleaved;
    bool normalize_error = true;
    bool has_seed = false; // reference: rand::rng()
    uint64_t seed = 0;
    TreeTciOptions to_treetci_options() const // :83-97
    {
        TreeTciOptions o;
        o.tolerance = tolerance;
        o.max_iter = max_iter;
        o.has_max_bond_dim = max_bond_dim != 0;
        o.max_bond_dim = max_bond_dim;
        o.normalize_error = normalize_error;
        o.enable_global_pivots = false;
        o.nsearch = 0;
        o.max_nglobal_pivot = 0;
        o.tol_margin_global_search = 10.0;
        return o;
    }
};

using CoordFn = std::function<double(const std::vector<double>&)>;
using GridIdxFn = std::function<double(const std::vector<size_t>&)>;

struct QuanticsTensorCI2 { // quantics_tci.rs:53-173
    QuanticsGrid grid;
    SimpleTensorTrain tt;
    std::unique_ptr<TreeTCI2> tci;
    std::map<MultiIndex, double> cache;
    std::vector<size_t> ranks;
    std::vector<double> errors;

    double evaluate(const std::vector<size_t>& grididx) const { return tt.evaluate(grid.grididx_to_quantics(grididx)); }
    double sum() const { return tt.sum(); }
    double integral() const // :130-141
    {
        const double s = sum();
        if (!grid.discretized) return s;
        double step = 1.0;
        for (double v : grid.grid_step()) step = step * v;
        return s * step;
    }
};

namespace quantics_detail {

// tree network of a linear chain rooted at site 0 -> tensor train (tensor4all-treetn simplett bridge):
// site tensors are [d, bond to site k+1, bond to site k-1]
inline SimpleTensorTrain chain_network_to_tt(const TreeNetwork& net)
{
    const size_t n = net.tensors.size();
    std::vector<Tensor3> cores(n);
    for (size_t k = 0; k < n; ++k) {
        const TreeSiteTensor& t = net.tensors[k];
        const size_t d = t.dims[0];
        const size_t r = (k + 1 < n) ? t.dims[1] : 1;
        const size_t l = (k > 0) ? t.dims.back() : 1;
        Tensor3 c(l, d, r);
        for (size_t a = 0; a < l; ++a)
            for (size_t s = 0; s < d; ++s)
                for (size_t b = 0; b < r; ++b) c.at(a, s, b) = t.data[s + d * (b + r * a)];
        cores[k] = c;
    }
    return SimpleTensorTrain::make(cores);
}

inline QuanticsTensorCI2 run(const QuanticsGrid& grid, const std::function<double(const MultiIndex&)>& point_fn,
                             const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options)
{
    QuanticsTensorCI2 out;
    out.grid = grid;
    const std::vector<size_t> local_dims = grid.local_dimensions();
    const size_t n_sites = local_dims.size();
    auto* cache = &out.cache;
    TreeBatchFn batch_eval = [cache, point_fn](const std::vector<size_t>& data, size_t ns, size_t npts) {
        std::vector<double> res(npts);
        MultiIndex q(ns);
        for (size_t p = 0; p < npts; ++p) {
            for (size_t s = 0; s < ns; ++s) q[s] = data[s + ns * p];
            auto it = cache->find(q);
            if (it == cache->end()) it = cache->emplace(q, point_fn(q)).first;
            res[p] = it->second;
        }
        return res;
    };
    std::vector<MultiIndex> pivots;
    if (initial_pivots) {
        for (const auto& g : *initial_pivots) pivots.push_back(grid.grididx_to_quantics(g));
    } else {
        pivots.push_back(MultiIndex(n_sites, 0));
    }
    OracleStdRng rng(options.has_seed ? options.seed : 0x13198A2E03707344ull);
    for (size_t k = 0; k < options.n_random_init_pivot; ++k) {
        MultiIndex p(n_sites);
        for (size_t s = 0; s < n_sites; ++s) p[s] = rng.range(local_dims[s]);
        pivots.push_back(p);
    }
    if (pivots.empty()) pivots.push_back(MultiIndex(n_sites, 0));
    if (n_sites < 2) throw OracleError(ERR_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
    out.tci.reset(new TreeTCI2(local_dims, TreeGraph::linear_chain(n_sites)));
    out.tci->add_global_pivots(pivots);
    std::vector<size_t> flat;
    for (const auto& p : pivots) flat.insert(flat.end(), p.begin(), p.end());
    const std::vector<double> init = batch_eval(flat, n_sites, pivots.size());
    double m = 0.0;
    for (double v : init) m = std::max(m, std::sqrt(v * v));
    out.tci->max_sample_value = m;
    if (out.tci->max_sample_value <= 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "initial pivots must not all evaluate to zero");
    TreeOptimizeResult r = tree_optimize(*out.tci, batch_eval, options.to_treetci_options());
    out.ranks = r.ranks;
    out.errors = r.errors;
    TreeNetwork net = tree_materialize(*out.tci, batch_eval, 0);
    out.tt = chain_network_to_tt(net);
    return out;
}

inline void check_power_of_two_sizes(const std::vector<size_t>& sizes) // :454-472 / :355-375
{
    std::vector<double> dims;
    for (size_t s : sizes) dims.push_back(std::log2((double)s));
    for (size_t k = 0; k + 1 < dims.size(); ++k)
        if (!(std::fabs(dims[k] - dims[k + 1]) < 1e-10))
            throw OracleError(ERR_INVALID_ARGUMENT, "this method only supports grids with equal number of points in each direction");
    for (double d : dims)
        if (!(std::fabs(d - std::round(d)) < 1e-10))
            throw OracleError(ERR_INVALID_ARGUMENT, "this method only supports grid sizes that are powers of 2");
}

} // namespace quantics_detail

// quantics_tci.rs:175-307
inline QuanticsTensorCI2 quanticscrossinterpolate(const QuanticsGrid& grid, const CoordFn& f,
                                                  const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options)
{
    if (!grid.discretized) throw OracleError(ERR_INVALID_ARGUMENT, "a discretized grid is required");
    return quantics_detail::run(grid, [&grid, f](const MultiIndex& q) { return f(grid.quantics_to_origcoord(q)); }, initial_pivots, options);
}

// :434-560
inline QuanticsTensorCI2 quanticscrossinterpolate_discrete(const std::vector<size_t>& sizes, const GridIdxFn& f,
                                                           const std::vector<std::vector<size_t>>* initial_pivots,
                                                           const QtciOptions& options)
{
    if (sizes.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "this method requires at least one grid dimension, got an empty size");
    quantics_detail::check_power_of_two_sizes(sizes);
    const size_t r = (size_t)std::log2((double)sizes[0]);
    QuanticsGrid grid(std::vector<size_t>(sizes.size(), r), options.unfolding, false);
    QuanticsGrid g2 = grid;
    return quantics_detail::run(grid, [g2, f](const MultiIndex& q) { return f(g2.quantics_to_grididx(q)); }, initial_pivots, options);
}

// :309-432
inline QuanticsTensorCI2 quanticscrossinterpolate_from_arrays(const std::vector<std::vector<double>>& xvals, const CoordFn& f,
                                                              const std::vector<std::vector<size_t>>* initial_pivots,
                                                              const QtciOptions& options)
{
    if (xvals.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "xvals must not be empty");
    for (const auto& x : xvals)
        if (x.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "xvals must not contain empty dimensions");
    for (const auto& x : xvals) {
        for (double v : x)
            if (!std::isfinite(v)) throw OracleError(ERR_INVALID_ARGUMENT, "xvals must contain only finite values");
        for (size_t k = 0; k + 1 < x.size(); ++k)
            if (x[k] >= x[k + 1]) throw OracleError(ERR_INVALID_ARGUMENT, "xvals must be strictly increasing without duplicates");
    }
    std::vector<size_t> sizes;
    for (const auto& x : xvals) sizes.push_back(x.size());
    quantics_detail::check_power_of_two_sizes(sizes);
    bool uniform = true;
    for (const auto& x : xvals) {
        if (x.size() < 2) continue;
        const double step = x[1] - x[0];
        for (size_t k = 0; k + 1 < x.size(); ++k)
            if (!(std::fabs(x[k + 1] - x[k] - step) <= 1e-12)) uniform = false;
    }
    if (uniform) {
        std::vector<size_t> rs;
        std::vector<double> lo, up;
        for (const auto& x : xvals) {
            rs.push_back((size_t)std::log2((double)x.size()));
            lo.push_back(x.front());
            up.push_back(x.back());
        }
        QuanticsGrid grid(rs, options.unfolding, true, lo, up, true);
        return quanticscrossinterpolate(grid, f, initial_pivots, options);
    }
    auto coords = xvals;
    return quanticscrossinterpolate_discrete(sizes, [coords, f](const std::vector<size_t>& idx) {
        std::vector<double> x(idx.size());
        for (size_t d = 0; d < idx.size(); ++d) x[d] = coords[d][idx[d]];
        return f(x);
    }, initial_pivots, options);
}

// ---------------------------------------------------------------------------------------------
// batched/mod.rs — vector / tensor valued functions
// ---------------------------------------------------------------------------------------------
using CoordVecFn = std::function<std::vector<double>(const std::vector<double>&)>;

// :193-318
inline SimpleTensorTrain combine_component_tts(const std::vector<SimpleTensorTrain>& comps)
{
    if (comps.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "no component tensor trains to combine");
    const size_t n = comps[0].len();
    if (n == 0) throw OracleError(ERR_INVALID_ARGUMENT, "component tensor trains must have at least one site");
    for (const auto& tt : comps) {
        if (tt.len() != n) throw OracleError(ERR_INVALID_ARGUMENT, "components have different site counts");
        for (size_t s = 0; s < n; ++s)
            if (tt.tensors[s].s != comps[0].tensors[s].s) throw OracleError(ERR_INVALID_ARGUMENT, "components have different site dimensions");
    }
    std::vector<Tensor3> out;
    for (size_t s = 0; s < n; ++s) {
        size_t total_r = 0, total_l = 0;
        for (const auto& tt : comps) {
            total_r += tt.tensors[s].r;
            total_l += tt.tensors[s].l;
        }
        const size_t sd = comps[0].tensors[s].s;
        Tensor3 c(s == 0 ? 1 : total_l, sd, total_r);
        size_t lo = 0, ro = 0;
        for (const auto& tt : comps) {
            const Tensor3& t = tt.tensors[s];
            for (size_t l = 0; l < t.l; ++l)
                for (size_t x = 0; x < sd; ++x)
                    for (size_t r = 0; r < t.r; ++r) c.at((s == 0 ? 0 : lo) + l, x, ro + r) = t.at(l, x, r);
            lo += t.l;
            ro += t.r;
        }
        out.push_back(c);
    }
    size_t total_r = 0;
    for (const auto& tt : comps) total_r += tt.tensors[n - 1].r;
    Tensor3 sel(total_r, comps.size(), 1);
    size_t off = 0;
    for (size_t c = 0; c < comps.size(); ++c) {
        for (size_t i = 0; i < comps[c].tensors[n - 1].r; ++i) sel.at(off + i, c, 0) = 1.0;
        off += comps[c].tensors[n - 1].r;
    }
    out.push_back(sel);
    return SimpleTensorTrain::make(out);
}

struct QuanticsBatchedResult {
    SimpleTensorTrain tt;
    std::vector<size_t> output_dims;
    std::vector<size_t> ranks;
    std::vector<double> errors;
    size_t n_user_calls = 0;
};

// :50-191
inline QuanticsBatchedResult quanticscrossinterpolate_batched(const QuanticsGrid& grid, const CoordVecFn& f,
                                                              const std::vector<size_t>& output_dims,
                                                              const std::vector<std::vector<size_t>>* initial_pivots,
                                                              const QtciOptions& options)
{
    if (output_dims.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "output_dims must not be empty");
    size_t n_comp = 1;
    for (size_t d : output_dims) n_comp *= d;
    if (n_comp == 0) throw OracleError(ERR_INVALID_ARGUMENT, "product of output_dims must be positive");
    QuanticsBatchedResult res;
    res.output_dims = output_dims;
    std::map<std::vector<uint64_t>, std::vector<double>> cache;
    std::vector<SimpleTensorTrain> comps;
    for (size_t comp = 0; comp < n_comp; ++comp) {
        std::string short_result;
        CoordFn scalar = [&, comp](const std::vector<double>& x) {
            std::vector<uint64_t> key(x.size());
            for (size_t d = 0; d < x.size(); ++d) std::memcpy(&key[d], &x[d], sizeof(double));
            auto it = cache.find(key);
            if (it == cache.end()) {
                ++res.n_user_calls;
                it = cache.emplace(key, f(x)).first;
            }
            if (comp >= it->second.size()) {
                short_result = "callback returned " + std::to_string(it->second.size()) + " components, expected at least " +
                               std::to_string(comp + 1);
                return 0.0;
            }
            return it->second[comp];
        };
        std::unique_ptr<QuanticsTensorCI2> q;
        try {
            q.reset(new QuanticsTensorCI2(quanticscrossinterpolate(grid, scalar, initial_pivots, options)));
        } catch (...) {
            if (!short_result.empty()) throw OracleError(ERR_INVALID_ARGUMENT, short_result);
            throw;
        }
        if (!short_result.empty()) throw OracleError(ERR_INVALID_ARGUMENT, short_result);
        comps.push_back(q->tt);
        if (res.ranks.size() < q->ranks.size()) res.ranks.resize(q->ranks.size(), 0);
        if (res.errors.size() < q->errors.size()) res.errors.resize(q->errors.size(), 0.0);
        for (size_t k = 0; k < q->ranks.size(); ++k) res.ranks[k] = std::max(res.ranks[k], q->ranks[k]);
        for (size_t k = 0; k < q->errors.size(); ++k) res.errors[k] = std::max(res.errors[k], q->errors[k]);
    }
    res.tt = combine_component_tts(comps);
    return res;
}

} // namespace t4a_oracle
