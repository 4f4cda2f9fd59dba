// globalsearch.hip — post-hoc error estimation and first-pivot optimisation of tensor4all-tensorci on top of the device
// tensor train: estimate_true_error / floating_zone (crates/tensor4all-tensorci/src/globalsearch.rs:70-243, walk:
// crates/tensor4all-core/src/floating_zone.rs:46-103) and opt_first_pivot (optfirstpivot.rs:40-74).
//
// The walks are host logic exactly as in the reference; what runs on the GPU is the batched tensor-train evaluation of every
// site scan (TensorTrain::evaluate_many = TTCache::evaluate_many, bit-identical to the CPU restatement).  The exact function is
// the caller's batch callback (t4a_gpu_batch_eval_fn): the reference calls f point by point, the values are the same.
// Random starting points: the reference draws `rng.random_range(0..d)` from the caller's rng (globalsearch.rs:93-99; the thread rng
// when init_p is None, :202); here rand 0.9 `StdRng::seed_from_u64(seed)` as restated in stdrng.hpp.
#include "globalsearch.hpp"
#include "stdrng.hpp"

#include <algorithm>
#include <cmath>

namespace t4a {

namespace {
void check_dims(const TensorTrain& tt, const std::vector<size_t>& local_dims)
{
    if (local_dims.size() != tt.len())
        throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims length " + std::to_string(local_dims.size()) + " does not match tensor train length " +
                                                  std::to_string(tt.len()));
    for (size_t d : local_dims)
        if (d == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims must contain only positive dimensions");
}
} // namespace

// globalsearch.rs:163-243
std::pair<std::vector<uint32_t>, double> floating_zone(TensorTrain& tt, const SearchFn& f, const std::vector<size_t>& local_dims,
                                                       const std::vector<uint32_t>* init_p, uint64_t seed, double early_stop_tol)
{
    check_dims(tt, local_dims);
    const size_t n = local_dims.size();
    std::vector<uint32_t> pivot(n);
    if (init_p) {
        if (init_p->size() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial pivot does not fit local_dims");
        for (size_t s = 0; s < n; ++s)
            if ((*init_p)[s] >= local_dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial pivot does not fit local_dims");
        pivot = *init_p;
    } else {
        StdRng st(seed);
        for (size_t s = 0; s < n; ++s) pivot[s] = (uint32_t)st.random_range(local_dims[s]);
    }
    if (n > std::numeric_limits<size_t>::max() / 10) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims sweep count overflowed usize");
    const size_t max_sweeps = n * 10;

    std::vector<uint32_t> pts;
    std::vector<double> tv, fv;
    // |f - tt| at a batch of points (columns of pts): one device evaluate_many + one callback per site scan
    auto errors_at = [&](size_t n_pts) {
        tv.assign(n_pts, 0.0);
        fv.assign(n_pts, 0.0);
        tt.evaluate_many(pts.data(), n_pts, 0, tv.data());
        f(pts.data(), n, n_pts, fv.data());
        for (size_t p = 0; p < n_pts; ++p) {
            const double d = fv[p] - tv[p];
            tv[p] = std::sqrt(d * d);
        }
    };
    // floating_zone_walk (tensor4all-core/src/floating_zone.rs:46-103)
    pts.assign(pivot.begin(), pivot.end());
    errors_at(1);
    double max_error = tv[0];
    for (size_t sw = 0; sw < max_sweeps; ++sw) {
        const double prev = max_error;
        for (size_t ipos = 0; ipos < n; ++ipos) {
            const size_t d = local_dims[ipos];
            pts.resize(n * d);
            for (size_t v = 0; v < d; ++v) {
                std::copy(pivot.begin(), pivot.end(), pts.begin() + v * n);
                pts[v * n + ipos] = (uint32_t)v;
            }
            errors_at(d);
            uint32_t best_idx = pivot[ipos];
            double best = 0.0;
            for (size_t v = 0; v < d; ++v)
                if (tv[v] > best) {
                    best = tv[v];
                    best_idx = (uint32_t)v;
                }
            pivot[ipos] = best_idx;
            max_error = std::fmax(max_error, best);
        }
        if (max_error == prev || max_error > early_stop_tol) break;
    }
    return {pivot, max_error};
}

// globalsearch.rs:70-118
std::vector<std::pair<std::vector<uint32_t>, double>> estimate_true_error(TensorTrain& tt, const SearchFn& f, size_t nsearch,
                                                                          const std::vector<std::vector<uint32_t>>* initial_points,
                                                                          uint64_t seed)
{
    const std::vector<size_t> site_dims = tt.site_dims();
    for (size_t d : site_dims)
        if (d == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "tensor train contains a zero-dimensional site");
    std::vector<std::vector<uint32_t>> points;
    if (initial_points) {
        points = *initial_points;
    } else {
        StdRng st(seed);
        for (size_t k = 0; k < nsearch; ++k) {
            std::vector<uint32_t> p(site_dims.size());
            for (size_t s = 0; s < site_dims.size(); ++s) p[s] = (uint32_t)st.random_range(site_dims[s]);
            points.push_back(std::move(p));
        }
    }
    std::vector<std::pair<std::vector<uint32_t>, double>> out;
    for (const auto& p : points) out.push_back(floating_zone(tt, f, site_dims, &p, 0, std::numeric_limits<double>::max()));
    // sort_by(b.1.partial_cmp(a.1)) is a stable sort by descending error; dedup_by removes CONSECUTIVE equal pivots
    std::stable_sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    out.erase(std::unique(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first == b.first; }), out.end());
    return out;
}

// optfirstpivot.rs:40-74.  The candidates of one site differ only in that site's coordinate, so they are evaluated as one
// batch; the accept rule (strict improvement, scanned in order) is applied to the values afterwards — the same trajectory.
std::vector<uint32_t> opt_first_pivot(const SearchFn& f, const std::vector<size_t>& local_dims, const std::vector<uint32_t>& first_pivot,
                                      size_t max_sweep)
{
    const size_t n = local_dims.size();
    if (first_pivot.size() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "first pivot does not fit local_dims");
    for (size_t s = 0; s < n; ++s)
        if (local_dims[s] == 0 || first_pivot[s] >= local_dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "first pivot does not fit local_dims");
    std::vector<uint32_t> pivot = first_pivot;
    std::vector<uint32_t> pts;
    std::vector<double> vals(1);
    f(pivot.data(), n, 1, vals.data());
    double val_f = std::sqrt(vals[0] * vals[0]);
    for (size_t sw = 0; sw < max_sweep; ++sw) {
        const double prev = val_f;
        for (size_t i = 0; i < n; ++i) {
            const size_t d = local_dims[i];
            pts.resize(n * d);
            for (size_t v = 0; v < d; ++v) {
                std::copy(pivot.begin(), pivot.end(), pts.begin() + v * n);
                pts[v * n + i] = (uint32_t)v;
            }
            vals.assign(d, 0.0);
            f(pts.data(), n, d, vals.data());
            for (size_t v = 0; v < d; ++v) {
                const double nv = std::sqrt(vals[v] * vals[v]);
                if (nv > val_f) {
                    val_f = nv;
                    pivot[i] = (uint32_t)v;
                }
            }
        }
        if (prev == val_f) break;
    }
    return pivot;
}

} // namespace t4a
