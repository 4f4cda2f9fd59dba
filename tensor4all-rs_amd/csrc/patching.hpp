// patching.hpp — adaptive patching driver on the gfx950 TCI2 engine (BASELINE config 5).
// Mirrors crates/tensor4all-partitionedtt/src/adaptive_interpolation.rs (adaptiveinterpolate :58-262): a FIFO queue
// of projected patches, one crossinterpolate2 per patch on the device, non-converged patches split along the next
// unprojected site of `patch_order`.  Index objects of the reference are plain site positions here.
#pragma once

#include <map>
#include <memory>

#include "tci2.hpp"

namespace t4a {

struct AdaptiveOptions { // adaptive_interpolation.rs:27-52
    TCI2Options tci;
    std::vector<size_t> patch_order; // empty = natural site order
    size_t n_initial_pivots = 5;
    bool recycle_pivots = false;
};

// the user function over the FULL index space
struct FullFunction {
    bool builtin = false;
    int fid = 0, n_acc = 0;
    double params[T4A_FN_MAX_PARAMS] = {0};
    std::vector<uint64_t> weights; // n_acc * sum(dims)
    t4a_gpu_batch_eval_fn cb = nullptr;
    void* ctx = nullptr;
};

struct SubDomain {
    std::map<size_t, size_t> projector; // site position -> fixed value
    std::vector<DevCore> cores;         // over ALL sites; projected sites carry delta ("copy selector") tensors
};

class PartitionedTT {
public:
    explicit PartitionedTT(std::vector<size_t> d) : dims(std::move(d)) {}
    std::vector<size_t> dims;
    std::vector<SubDomain> patches; // in acceptance (FIFO) order
    Engine eng;
    // sum over the patches of their tensor trains (every train vanishes outside its projector)
    std::vector<double> evaluate(const uint32_t* idx, size_t n_pts); // idx n_sites x n_pts col-major

private:
    DevBuf<TtCoreDesc> d_desc_;
    DevBuf<uint32_t> d_idx_;
    DevBuf<double> d_vals_;
};

std::unique_ptr<PartitionedTT> adaptive_interpolate(const std::vector<size_t>& dims, const FullFunction& f,
                                                    const std::vector<std::vector<uint32_t>>& initial_pivots,
                                                    const AdaptiveOptions& options);

} // namespace t4a
