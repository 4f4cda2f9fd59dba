// t4a_oracle_search.hpp — CPU restatement (TEST INFRASTRUCTURE ONLY: nothing under tensor4all-rs_amd/ may include this) of the
// post-hoc error estimation and the first-pivot search of tensor4all-tensorci:
//   * floating_zone          crates/tensor4all-tensorci/src/globalsearch.rs:163-243
//                            (walk: crates/tensor4all-core/src/floating_zone.rs:46-103, restated in t4a_oracle_aci.hpp)
//   * estimate_true_error    globalsearch.rs:70-118
//   * opt_first_pivot        crates/tensor4all-tensorci/src/optfirstpivot.rs:40-74
// Pinned by tests/test_oracle_search.py to the fixtures the reference holds: floating_zone -> pivot [3,3], error 8
// (globalsearch.rs:245-279) and error 9 (doc example :141-155), estimate_true_error best = ([3,3], 8) and sorted output
// (:30-46, :281-310), opt_first_pivot -> [3,3] from [0,0] and unchanged when already optimal (optfirstpivot.rs:80-101).
// Random starting points (rand 0.9 StdRng / thread rng) are "parity unpinned": splitmix64 (OracleRng) here.
#pragma once

#include "t4a_oracle_aci.hpp" // floating_zone_walk
#include "t4a_oracle_tt.hpp"

#include <algorithm>

namespace t4a_oracle {

// globalsearch.rs:163-243 (init == nullptr: random starting point)
inline std::pair<MultiIndex, double> floating_zone(const SimpleTensorTrain& tt, const ScalarFn& f, const std::vector<size_t>& local_dims,
                                                   const MultiIndex* init, uint64_t seed, double early_stop_tol)
{
    if (local_dims.size() != tt.len()) throw OracleError(ERR_INVALID_ARGUMENT, "local_dims length does not match tensor train length");
    for (size_t d : local_dims)
        if (d == 0) throw OracleError(ERR_INVALID_ARGUMENT, "local_dims must contain only positive dimensions");
    MultiIndex init_p;
    if (init) {
        if (init->size() != local_dims.size()) throw OracleError(ERR_INVALID_ARGUMENT, "initial pivot does not fit local_dims");
        for (size_t s = 0; s < local_dims.size(); ++s)
            if ((*init)[s] >= local_dims[s]) throw OracleError(ERR_INVALID_ARGUMENT, "initial pivot does not fit local_dims");
        init_p = *init;
    } else {
        OracleStdRng rng(seed);
        for (size_t d : local_dims) init_p.push_back(rng.range(d));
    }
    const size_t max_sweeps = local_dims.size() * 10; // :207-214
    TTCache cache(tt);
    return floating_zone_walk(local_dims, init_p, max_sweeps, early_stop_tol, [&](const std::vector<MultiIndex>& points) {
        const std::vector<double> tv = cache.evaluate_many(points, 0); // :222-224 (split = None)
        std::vector<double> errs(points.size());
        for (size_t p = 0; p < points.size(); ++p) {
            const double d = f(points[p]) - tv[p];
            errs[p] = std::sqrt(d * d); // sqrt(abs_sq(diff))
        }
        return errs;
    });
}

// globalsearch.rs:70-118
inline std::vector<std::pair<MultiIndex, double>> estimate_true_error(const SimpleTensorTrain& tt, const ScalarFn& f, size_t nsearch,
                                                                      const std::vector<MultiIndex>* initial_points, uint64_t seed)
{
    std::vector<size_t> site_dims;
    for (size_t i = 0; i < tt.len(); ++i) site_dims.push_back(tt.tensors[i].s);
    for (size_t d : site_dims)
        if (d == 0) throw OracleError(ERR_INVALID_ARGUMENT, "tensor train contains a zero-dimensional site");
    std::vector<MultiIndex> points;
    if (initial_points) {
        points = *initial_points;
    } else {
        OracleStdRng rng(seed);
        for (size_t k = 0; k < nsearch; ++k) {
            MultiIndex p;
            for (size_t d : site_dims) p.push_back(rng.range(d));
            points.push_back(std::move(p));
        }
    }
    std::vector<std::pair<MultiIndex, double>> out;
    for (const auto& p : points) out.push_back(floating_zone(tt, f, site_dims, &p, 0, std::numeric_limits<double>::max()));
    std::stable_sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.second > b.second; }); // :111-112
    out.erase(std::unique(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first == b.first; }), out.end()); // :115
    return out;
}

// optfirstpivot.rs:40-74 (point by point, as written)
inline MultiIndex opt_first_pivot(const ScalarFn& f, const std::vector<size_t>& local_dims, const MultiIndex& first_pivot, size_t max_sweep)
{
    const size_t n = local_dims.size();
    MultiIndex pivot = first_pivot;
    auto mag = [&](const MultiIndex& p) {
        const double v = f(p);
        return std::sqrt(v * v);
    };
    double val_f = mag(pivot);
    for (size_t sw = 0; sw < max_sweep; ++sw) {
        const double prev = val_f;
        for (size_t i = 0; i < n; ++i)
            for (size_t d = 0; d < local_dims[i]; ++d) {
                const size_t bak = pivot[i];
                pivot[i] = d;
                const double nv = mag(pivot);
                if (nv > val_f) val_f = nv;
                else pivot[i] = bak;
            }
        if (prev == val_f) break;
    }
    return pivot;
}

} // namespace t4a_oracle
