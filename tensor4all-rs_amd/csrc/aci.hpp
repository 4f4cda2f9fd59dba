// aci.hpp — device mirror of tensor4all-aci (crates/tensor4all-aci/src): elementwise / elementwise_batched
// (elementwise.rs:107-218), ElementwiseProblem (state.rs:24-925: frames, local_update, add_global_pivots),
// LocalBlockEvaluator::materialize_local_matrix (local.rs:299-394), the global guard (global_guard.rs:49-181 on
// tensor4all-core's floating_zone_walk) and the initial guess (random_tt.rs:15-150).
//
// HBM layout: the input trains stay where their t4a_gpu_tt handles put them; per (input, site) one left and one right frame
// (column-major, rows = pivots / columns = input bond and vice versa); the solution cores; per bond update the left factors
// (R*s1 x m_k), the right factors (m_k x s2*C), the candidate matrix and the LUCI factors of the engine.  Every product is
// accumulated k-ascending with separately rounded multiply and add (the order the oracle restates), so the candidate matrix
// — and with it every rrLU pivot — is bit-identical to the CPU restatement.
#pragma once

#include <functional>
#include <memory>

#include "tt.hpp"

namespace t4a {

enum class AciOpKind : int { Callback = 0, Product = 1, Sum = 2 };
enum class AciTermination : int { Converged = 0, RankLimited = 1, MaxIterations = 2 }; // result.rs

// values[input + n_inputs * point] -> out[point] (batch.rs:33-217); throws to stop the sweep
using AciHostOp = std::function<void(const double* values, size_t n_inputs, size_t n_points, double* out)>;

struct AciOptions { // options.rs:37-168
    size_t max_iters = 20;
    size_t min_iters = 2;
    bool has_max_bond_dim = false;
    size_t max_bond_dim = 0;
    double tolerance = 1e-12;
    bool scale_tolerance = true;
    uint64_t rng_seed = 0;
    bool enable_global_guard = true;
    size_t nsearch_global_pivots = 5;
    size_t max_nglobal_pivot = 5;
    size_t nsweeps_global_search = 100;
    double tol_margin_global_search = 10.0;
    void validate() const; // validation.rs:4-46
};

struct AciFrame {
    DevBuf<double> buf;
    size_t nr = 0, nc = 0;
    bool present = false;
};

constexpr int ACI_MAX_INPUTS = 8;

class AciProblem {
public:
    // `inputs` are borrowed (they must outlive the problem); `guess` == nullptr: default random guess (random_tt.rs:15-39,
    // splitmix64 / Box-Muller stream instead of ChaCha8: "parity unpinned")
    AciProblem(const std::vector<TensorTrain*>& inputs, const TensorTrain* guess, const AciOptions& options, AciOpKind kind,
               AciHostOp host_op);

    size_t len() const { return sol_.size(); }
    size_t n_inputs() const { return inputs_.size(); }
    size_t rank() const;
    std::vector<size_t> link_dims() const;

    void local_update(size_t bond, bool left_orthogonal);
    size_t add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots);
    std::vector<std::vector<uint32_t>> find_global_pivots(uint64_t seed);
    // the sweep loop of elementwise_batched; histories in ranks / errors / nglobal_pivots afterwards
    void run();

    std::unique_ptr<TensorTrain> solution_tt(); // device-to-device copy of the current solution
    std::vector<double> frame_host(bool right, size_t input, size_t site, size_t* nr, size_t* nc);

    std::vector<double> pivot_errors, pivot_scales;
    std::vector<size_t> ranks, nglobal_pivots;
    std::vector<double> errors;
    AciTermination termination = AciTermination::MaxIterations;

private:
    void initialize_right_frames();
    // full left / right factor of input k at `site`: frame x core, core x frame (build_left_factor / build_right_factor)
    void left_factor(size_t k, size_t site, DevBuf<double>& out);
    void right_factor(size_t k, size_t site, DevBuf<double>& out);
    void select_left_frames(size_t site, const std::vector<int>& rows);  // update_left_frames from lf_
    void select_right_frames(size_t site, const std::vector<int>& cols); // update_right_frames from rf_
    void apply_op_host(const double* values, size_t n_points, double* out);
    void set_frame(AciFrame& f, const std::vector<double>& host, size_t nr, size_t nc);

    std::vector<TensorTrain*> inputs_;
    AciOptions opt_;
    AciOpKind kind_;
    AciHostOp host_op_;
    std::vector<DevCore> sol_;
    std::vector<std::vector<AciFrame>> lframes_, rframes_;
    std::vector<DevBuf<double>> lf_, rf_; // per input
    DevBuf<double> d_vals_, d_tmp_, d_env_;
    DevBuf<int> d_idx_;
    Engine eng_;
};

// single-site trains (elementwise.rs:220-254): one operator call over the site points
std::unique_ptr<TensorTrain> aci_one_site(const std::vector<TensorTrain*>& inputs, AciOpKind kind, const AciHostOp& host_op);

// tensor4all-treeaci local step (local_update.rs:35-262 materialize_and_factor_edge, from the candidate frames on): per input k the
// candidate values are row_frames[k]^T col_frames[k] (bond_k x row_count and bond_k x col_count, column-major: a candidate's frame
// vector is contiguous, as candidate_frames_for_edge hands them out), the operator combines the inputs point by point (batch layout
// values[input + n_inputs * (row + row_count * col)], batch.rs), the local matrix goes through MatrixLUCI.  Everything between the
// upload of the frames and the download of the factors runs on the device (aci_pi_kernel + the rrLU / factor kernels of the engine).
struct TreeAciLocalResult {
    size_t rank = 0;                      // factor rank (1 for a zero matrix: local_update.rs:230-238)
    std::vector<size_t> row_indices, col_indices;
    std::vector<double> pivot_errors;     // MatrixLuciFactors::pivot_errors (rank_luci + 1 entries)
    std::vector<double> left, right;      // row_count x rank, rank x col_count (column-major)
    double sampled_scale = 0.0;           // max |local value|
    std::vector<double> local_values;     // row_count x col_count
};
TreeAciLocalResult treeaci_local_update(Engine& eng, const std::vector<size_t>& bond_dims, const std::vector<const double*>& row_frames,
                                        const std::vector<const double*>& col_frames, size_t row_count, size_t col_count, AciOpKind kind,
                                        const AciHostOp& host_op, size_t max_bond_dim, double tolerance, bool scale_tolerance,
                                        bool left_orthogonal);

} // namespace t4a
