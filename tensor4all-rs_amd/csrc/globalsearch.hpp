// globalsearch.hpp — estimate_true_error / floating_zone / opt_first_pivot (tensor4all-tensorci/src/globalsearch.rs,
// optfirstpivot.rs) on top of the device tensor train.  See globalsearch.hip.
#pragma once

#include <functional>
#include <utility>
#include <vector>

#include "tt.hpp"

namespace t4a {

// exact function as a batch evaluator: idx is n_sites x n_pts column-major, one value per point (throws on failure)
using SearchFn = std::function<void(const uint32_t* idx, size_t n_sites, size_t n_pts, double* out)>;

std::pair<std::vector<uint32_t>, double> floating_zone(TensorTrain& tt, const SearchFn& f, const std::vector<size_t>& local_dims,
                                                       const std::vector<uint32_t>* init_p, uint64_t seed, double early_stop_tol);
std::vector<std::pair<std::vector<uint32_t>, double>> estimate_true_error(TensorTrain& tt, const SearchFn& f, size_t nsearch,
                                                                          const std::vector<std::vector<uint32_t>>* initial_points,
                                                                          uint64_t seed);
std::vector<uint32_t> opt_first_pivot(const SearchFn& f, const std::vector<size_t>& local_dims, const std::vector<uint32_t>& first_pivot,
                                      size_t max_sweep);

} // namespace t4a
