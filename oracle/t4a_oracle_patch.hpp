// oracle/t4a_oracle_patch.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp).
// CPU restatement of the adaptive patching driver (BASELINE config 5):
//   crates/tensor4all-partitionedtt/src/adaptive_interpolation.rs
//     adaptiveinterpolate :58-262, validate_inputs :264-355, patch_is_accepted :357-359,
//     active_positions :361-367, patch_candidates :369-443, is_compatible_pivot :445-458,
//     decode_col_major :460-468, expand_pivot :470-486, global_diagonal_pivots :488-512,
//     embed_active_tt :514-612 (as plain (l, s, r) cores), projected_site_tensor :614-660, rank_one_full_tt :662-709.
// The index objects (DynIndex) of the reference are plain site positions here; a Projector is a sorted
// position -> value map.  Parity: the replenishment pivots come from rand 0.9 StdRng in the reference
// ("parity unpinned": splitmix64 here); runs whose patches all receive enough compatible initial / recycled pivots
// never draw a random number.
#pragma once

#include <deque>
#include <map>
#include <set>

#include "t4a_oracle_tt.hpp"

namespace t4a_oracle {

using Projector = std::map<size_t, size_t>; // site position -> fixed value

struct AdaptiveInterpolateOptions { // :27-52
    TCI2Options tci_options;
    std::vector<size_t> patch_order; // site positions; empty = natural order
    size_t n_initial_pivots = 5;
    bool recycle_pivots = false;
};

struct SubDomainTT {
    Projector projector;
    SimpleTensorTrain tt; // over ALL sites; projected sites carry delta tensors
};

namespace patch_detail {

constexpr double ZERO_SAMPLE_THRESHOLD = 1.0e-30;

inline std::vector<size_t> active_positions(size_t n, const Projector& pr)
{
    std::vector<size_t> a;
    for (size_t p = 0; p < n; ++p)
        if (!pr.count(p)) a.push_back(p);
    return a;
}

inline MultiIndex expand_pivot(const MultiIndex& local, const std::vector<size_t>& active, const Projector& pr, size_t n)
{
    MultiIndex full(n, 0);
    for (size_t k = 0; k < active.size() && k < local.size(); ++k) full[active[k]] = local[k];
    for (const auto& kv : pr) full[kv.first] = kv.second;
    return full;
}

inline bool is_compatible(const MultiIndex& pivot, size_t n, const Projector& pr)
{
    if (pivot.size() != n) return false;
    for (const auto& kv : pr)
        if (pivot[kv.first] != kv.second) return false;
    return true;
}

inline std::vector<MultiIndex> patch_candidates(const std::vector<size_t>& dims, const std::vector<size_t>& active,
                                                const Projector& pr, const std::vector<MultiIndex>& initial,
                                                const std::vector<MultiIndex>& recycled, size_t target, OracleStdRng& rng)
{
    std::vector<MultiIndex> cand;
    std::set<MultiIndex> seen;
    auto take = [&](const MultiIndex& full) {
        if (!is_compatible(full, dims.size(), pr)) return;
        MultiIndex local;
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
            throw OracleError(ERR_INVALID_ARGUMENT, "active patch point count exceeds usize");
        point_count *= d;
    }
    const size_t desired = std::min(std::max(target, cand.size()), point_count);
    const size_t attempts = desired * 20 + 100;
    for (size_t a = 0; a < attempts && cand.size() < desired; ++a) {
        MultiIndex pv;
        for (size_t d : ld) pv.push_back(rng.range(d));
        if (seen.insert(pv).second) cand.push_back(pv);
    }
    for (size_t flat = 0; flat < point_count && cand.size() < desired; ++flat) {
        MultiIndex pv;
        size_t f = flat;
        for (size_t d : ld) { // decode_col_major
            pv.push_back(f % d);
            f /= d;
        }
        if (seen.insert(pv).second) cand.push_back(pv);
    }
    return cand;
}

// rank_one_full_tt :662-709
inline SimpleTensorTrain rank_one_full_tt(const std::vector<size_t>& dims, const Projector& pr, double scale)
{
    std::vector<Tensor3> ts;
    for (size_t p = 0; p < dims.size(); ++p) {
        Tensor3 t(1, dims[p], 1);
        const double ls = p == 0 ? scale : 1.0;
        auto it = pr.find(p);
        if (it != pr.end())
            t.at(0, it->second, 0) = ls;
        else
            for (size_t s = 0; s < dims[p]; ++s) t.at(0, s, 0) = ls;
        ts.push_back(t);
    }
    return SimpleTensorTrain::make(ts);
}

// embed_active_tt :514-612
inline SimpleTensorTrain embed_active_tt(const std::vector<Tensor3>& active_cores, const std::vector<size_t>& dims,
                                         const std::vector<size_t>& active, const Projector& pr)
{
    const size_t n = dims.size(), na = active.size();
    std::vector<size_t> link;
    for (size_t k = 0; k + 1 < na; ++k) link.push_back(active_cores[k].r);
    std::vector<size_t> edge(n > 0 ? n - 1 : 0, 1);
    for (size_t e = 0; e + 1 < n; ++e) {
        size_t left = 0;
        for (size_t p : active)
            if (p <= e) ++left;
        edge[e] = (left == 0 || left == na) ? 1 : link[left - 1];
    }
    std::vector<Tensor3> ts;
    size_t next_active = 0;
    for (size_t p = 0; p < n; ++p) {
        const size_t l = p == 0 ? 1 : edge[p - 1], r = p + 1 == n ? 1 : edge[p];
        if (next_active < na && active[next_active] == p) {
            const Tensor3& c = active_cores[next_active++];
            if (c.l != l || c.r != r || c.s != dims[p]) throw OracleError(ERR_INTERNAL, "embedded core shape mismatch");
            ts.push_back(c);
        } else {
            auto it = pr.find(p);
            if (it == pr.end()) throw OracleError(ERR_INVALID_ARGUMENT, "an embedded inactive site is missing from its projector");
            if (l != r) throw OracleError(ERR_INTERNAL, "projected site requires equal carried bonds");
            Tensor3 t(l, dims[p], r);
            for (size_t b = 0; b < l; ++b) t.at(b, it->second, b) = 1.0; // from_copy_selector
            ts.push_back(t);
        }
    }
    return SimpleTensorTrain::make(ts);
}

// global_diagonal_pivots :488-512
inline std::vector<MultiIndex> global_diagonal_pivots(const TensorCI2& tci, const std::vector<size_t>& active,
                                                      const Projector& pr, size_t n)
{
    std::vector<MultiIndex> out;
    std::set<MultiIndex> seen;
    for (size_t b = 0; b + 1 < active.size(); ++b) {
        const auto& is = tci.i_set[b + 1];
        const auto& js = tci.j_set[b];
        for (size_t k = 0; k < is.size() && k < js.size(); ++k) {
            MultiIndex local = is[k];
            local.insert(local.end(), js[k].begin(), js[k].end());
            if (local.size() == active.size()) {
                MultiIndex full = expand_pivot(local, active, pr, n);
                if (seen.insert(full).second) out.push_back(full);
            }
        }
    }
    return out;
}

} // namespace patch_detail

// validate_inputs :264-355
inline std::vector<size_t> validate_adaptive_inputs(const std::vector<size_t>& dims, const std::vector<MultiIndex>& pivots,
                                                    const AdaptiveInterpolateOptions& o)
{
    auto bad = [](const char* m) { throw OracleError(ERR_INVALID_ARGUMENT, m); };
    if (dims.empty()) bad("site_indices must not be empty");
    for (size_t d : dims)
        if (d == 0) bad("site indices must have positive dimensions");
    if (o.n_initial_pivots == 0) bad("n_initial_pivots must be positive");
    if (!std::isfinite(o.tci_options.tolerance) || o.tci_options.tolerance < 0.0) bad("TCI tolerance must be finite and nonnegative");
    if (o.tci_options.max_iter == 0) bad("TCI max_iter must be positive");
    if (o.tci_options.ncheck_history == 0) bad("TCI ncheck_history must be positive");
    if (!std::isfinite(o.tci_options.tol_margin_global_search) || o.tci_options.tol_margin_global_search < 0.0)
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

// adaptiveinterpolate :58-262
inline std::vector<SubDomainTT> adaptiveinterpolate(const ScalarFn& f, const BatchFn* batched, const std::vector<size_t>& dims,
                                                    const std::vector<MultiIndex>& initial_pivots,
                                                    const AdaptiveInterpolateOptions& options)
{
    using namespace patch_detail;
    const std::vector<size_t> patch_order = validate_adaptive_inputs(dims, initial_pivots, options);
    const size_t n = dims.size();
    OracleStdRng rng(options.tci_options.has_seed ? options.tci_options.seed : 0); // adaptive_interpolation.rs:164
    struct Pending {
        Projector projector;
        std::vector<MultiIndex> recycled;
    };
    std::deque<Pending> pending;
    pending.push_back({});
    std::vector<SubDomainTT> accepted;
    while (!pending.empty()) {
        Pending patch = pending.front();
        pending.pop_front();
        const std::vector<size_t> active = active_positions(n, patch.projector);
        if (active.empty()) {
            const double v = f(expand_pivot({}, active, patch.projector, n));
            accepted.push_back({patch.projector, rank_one_full_tt(dims, patch.projector, v)});
            continue;
        }
        if (active.size() == 1) {
            const size_t d = dims[active[0]];
            Tensor3 core(1, d, 1);
            for (size_t s = 0; s < d; ++s) core.at(0, s, 0) = f(expand_pivot({s}, active, patch.projector, n));
            accepted.push_back({patch.projector, embed_active_tt({core}, dims, active, patch.projector)});
            continue;
        }
        std::vector<MultiIndex> cand = patch_candidates(dims, active, patch.projector, initial_pivots, patch.recycled,
                                                        options.n_initial_pivots, rng);
        bool all_zero = true;
        for (const auto& c : cand) {
            const double v = f(expand_pivot(c, active, patch.projector, n));
            if (!(std::fabs(v) < ZERO_SAMPLE_THRESHOLD)) all_zero = false;
        }
        if (all_zero) {
            accepted.push_back({patch.projector, rank_one_full_tt(dims, patch.projector, 0.0)});
            continue;
        }
        std::vector<size_t> local_dims;
        for (size_t p : active) local_dims.push_back(dims[p]);
        ScalarFn local_f = [&](const MultiIndex& pv) { return f(expand_pivot(pv, active, patch.projector, n)); };
        BatchFn local_b;
        if (batched && *batched)
            local_b = [&](const std::vector<MultiIndex>& pvs) {
                std::vector<MultiIndex> full;
                for (const auto& pv : pvs) full.push_back(expand_pivot(pv, active, patch.projector, n));
                return (*batched)(full);
            };
        TensorCI2 tci(local_dims);
        OptimizationResult res = crossinterpolate2(tci, local_f, (batched && *batched) ? &local_b : nullptr, cand,
                                                   options.tci_options);
        const double normalization =
            (options.tci_options.normalize_error && tci.max_sample_value > 0.0) ? tci.max_sample_value : 1.0;
        const double final_error = res.errors.empty() ? tci.max_bond_error() / normalization : res.errors.back();
        if (res.termination == Termination::Converged && final_error <= options.tci_options.tolerance) { // :357-359
            accepted.push_back({patch.projector, embed_active_tt(tci.site_tensors, dims, active, patch.projector)});
            continue;
        }
        size_t split = n;
        for (size_t p : patch_order)
            if (!patch.projector.count(p)) {
                split = p;
                break;
            }
        if (split == n) throw OracleError(ERR_INVALID_ARGUMENT, "a nonconverged patch has no remaining split index");
        std::vector<MultiIndex> recycled;
        if (options.recycle_pivots) recycled = global_diagonal_pivots(tci, active, patch.projector, n);
        for (size_t v = 0; v < dims[split]; ++v) {
            Pending child;
            child.projector = patch.projector;
            child.projector[split] = v;
            child.recycled = recycled;
            pending.push_back(child);
        }
    }
    return accepted;
}

} // namespace t4a_oracle
