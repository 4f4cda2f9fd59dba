// quantics.hpp — host-side mirror of tensor4all-quanticstci (crates/tensor4all-quanticstci/src): QtciOptions
// (options.rs:9-97), QuanticsTensorCI2 (quantics_tci.rs:53-173), quanticscrossinterpolate (:175-307),
// quanticscrossinterpolate_from_arrays (:309-432), quanticscrossinterpolate_discrete (:434-560).
// The interpolation itself is the TreeTCI driver of tree.hpp on a linear chain; this layer owns the grid conversions
// (the un-vendored crate quanticsgrids @ 8214b72, restated from its published algorithm — conventions in the header of
// oracle/t4a_oracle_quantics.hpp), the memoised point evaluation and the bridge to a device-resident tensor train.
#pragma once

#include <memory>
#include <unordered_map>

#include "tree.hpp"
#include "tt.hpp"

namespace t4a {

enum class Unfolding : int { Interleaved = 0, Fused = 1 };

struct QuanticsGrid {
    std::vector<size_t> rs;
    std::vector<double> lower, upper;
    bool include_endpoint = false;
    Unfolding unfolding = Unfolding::Interleaved;
    bool discretized = true; // false: InherentDiscreteGrid
    // per site: (variable, bit level) pairs, first variable least significant inside a fused site
    std::vector<std::vector<std::pair<size_t, size_t>>> sites;

    QuanticsGrid() = default;
    QuanticsGrid(const std::vector<size_t>& rs, Unfolding u, bool discretized, const std::vector<double>& lower = {},
                 const std::vector<double>& upper = {}, bool include_endpoint = false);
    size_t n_vars() const { return rs.size(); }
    size_t n_sites() const { return sites.size(); }
    std::vector<size_t> local_dimensions() const;
    std::vector<double> grid_step() const;
    void grididx_to_quantics(const size_t* g, uint32_t* q) const;
    void quantics_to_grididx(const uint32_t* q, size_t* g) const;
    void quantics_to_origcoord(const uint32_t* q, double* x) const;
};

struct QtciOptions { // options.rs:9-45
    double tolerance = 1e-8;
    size_t max_bond_dim = 0; // 0 == None
    size_t max_iter = 200;
    size_t n_random_init_pivot = 5;
    Unfolding unfolding = Unfolding::Interleaved;
    bool normalize_error = true;
    bool has_seed = false; // reference: rand::rng(); a fixed stream here unless a seed is given
    uint64_t seed = 0;
    TreeTciOptions to_treetci_options() const; // :83-97
};

class QuanticsTci {
public:
    // exactly one of coord_cb / grididx_cb is set; `xvals` (optional) maps grid indices of an inherent grid to
    // coordinates before coord_cb is called (from_arrays on non-uniform coordinates)
    QuanticsTci(const QuanticsGrid& grid, t4a_gpu_coord_eval_fn coord_cb, t4a_gpu_grididx_eval_fn grididx_cb, void* ctx,
                std::vector<std::vector<double>> xvals);
    void run(const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options);

    std::vector<double> evaluate(const size_t* grididx, size_t n_pts); // grididx n_vars x n_pts col-major
    double sum();
    double integral();

    QuanticsGrid grid;
    std::unique_ptr<TreeTci> tci;
    std::unique_ptr<TensorTrain> tt;
    struct KeyHash {
        size_t operator()(const std::vector<uint32_t>& v) const
        {
            uint64_t h = 0xcbf29ce484222325ull;
            for (uint32_t x : v) h = (h ^ x) * 0x100000001b3ull;
            return (size_t)h;
        }
    };
    std::unordered_map<std::vector<uint32_t>, double, KeyHash> cache; // cachedata (:152)
    size_t n_user_calls = 0, n_user_points = 0;

private:
    static int64_t trampoline(void* ctx, const uint32_t* idx, size_t n_sites, size_t n_pts, double* out);
    int64_t eval_batch(const uint32_t* idx, size_t n_sites, size_t n_pts, double* out);
    t4a_gpu_coord_eval_fn coord_cb_;
    t4a_gpu_grididx_eval_fn grididx_cb_;
    void* ctx_;
    std::vector<std::vector<double>> xvals_;
};

// quanticscrossinterpolate_batched (batched/mod.rs:50-191): one scalar interpolation per output component over a shared
// coordinate-keyed cache (each coordinate reaches the user once, for all components), then combine_component_tts
// (:193-318): block-diagonal direct sum of the component trains plus a component selector site.
struct QuanticsBatchedResult {
    std::unique_ptr<TensorTrain> tt;
    std::vector<size_t> output_dims;
    std::vector<size_t> ranks;   // element-wise maximum over the components
    std::vector<double> errors;
    size_t n_user_calls = 0, n_user_points = 0;
};
QuanticsBatchedResult quantics_batched(const QuanticsGrid& grid, t4a_gpu_coord_eval_vec_fn f, void* ctx,
                                       const std::vector<size_t>& output_dims,
                                       const std::vector<std::vector<size_t>>* initial_pivots, const QtciOptions& options);
std::unique_ptr<TensorTrain> combine_component_tts(std::vector<std::unique_ptr<TensorTrain>>& comps);

// quanticscrossinterpolate_from_arrays / _discrete input checks (:322-375, :449-472)
void qtci_check_sizes(const std::vector<size_t>& sizes);
bool qtci_check_xvals_uniform(const std::vector<std::vector<double>>& xvals);

} // namespace t4a
