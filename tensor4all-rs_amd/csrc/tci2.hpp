// tci2.hpp — host-side mirror of tensor4all-tensorci's TensorCI2 driver
// (crates/tensor4all-tensorci/src/tensorci2.rs) on top of the gfx950 engine.
// Same names, argument meaning and error behaviour as the reference; the I/J index sets are the
// master copy on the host (flat digit tables), all matrices and site tensors live on the device.
#pragma once

#include <memory>
#include <vector>

#include "engine.hpp"
#include "rook.hpp"
#include "tt.hpp"

namespace t4a {

// A list of multi-indices of fixed width, flat: entry k = d[k*width .. (k+1)*width)
struct IndexSet {
    size_t width = 0;
    size_t count = 0;
    std::vector<uint32_t> d;
    const uint32_t* at(size_t k) const { return d.data() + k * width; }
    void push(const uint32_t* v)
    {
        d.insert(d.end(), v, v + width);
        ++count;
    }
    void clear()
    {
        d.clear();
        count = 0;
    }
    bool contains(const uint32_t* v) const;
};

struct TCI2Options { // tensorci2.rs:73-170
    double tolerance = 1e-8;
    size_t max_iter = 20;
    size_t max_bond_dim = 0; // 0 == None
    int pivot_search = 0;
    bool normalize_error = true;
    size_t verbosity = 0;
    size_t max_nglobal_pivot = 5;
    size_t nsearch = 5;
    int sweep_strategy = 2;
    size_t ncheck_history = 3;
    bool strictly_nested = false;
    double tol_margin_global_search = 10.0;
    bool has_seed = false;
    uint64_t seed = 0;
    size_t max_bond_dim_or_max() const
    {
        return max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
    }
    void validate() const;
};

struct FromTensorTrainOptions { // tensorci/src/conversion.rs:20-36
    double tolerance = 1e-12;
    size_t max_bond_dim = 0; // 0 == None
    size_t max_iter = 3;
};

class Tci2 {
public:
    explicit Tci2(const std::vector<size_t>& local_dims);
    ~Tci2();

    // function source
    void set_builtin(int fid, int n_acc, const double* params, const uint64_t* weights);
    void set_callback(t4a_gpu_batch_eval_fn cb, void* ctx);

    // reference API
    size_t len() const { return n_; }
    size_t rank() const;
    std::vector<size_t> link_dims() const;
    double max_bond_error() const;
    void add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots);
    void crossinterpolate2(std::vector<std::vector<uint32_t>> initial_pivots, const TCI2Options& options);
    void optimize(const TCI2Options& options, bool final_sweep1site);
    void sweep2site(bool forward, const TCI2Options& options);
    void sweep1site(bool forward, double rel_tol, double abs_tol, size_t max_bond_dim, bool update_tensors);
    void fill_site_tensors();
    void fill_site_tensors_impl(bool async);
    void fill_wait(); // completes an asynchronous fill (and reports its deferred errors)
    void export_site_tensors_async(double* d_dst, size_t stride, hipStream_t consumer);
    void export_site_shard_async(double* d_dst, size_t stride, hipStream_t consumer);
    void import_site_shard_async(const double* d_src, size_t stride, size_t per_rank, hipStream_t producer);
    void make_canonical(double rel_tol, double abs_tol, size_t max_bond_dim);
    void invalidate_site_tensors();
    void flush_pivot_errors() { pivot_errors.clear(); }

    std::vector<double> evaluate(const uint32_t* idx, size_t n_pts); // idx n_sites x n_pts col-major
    double sum();
    std::vector<double> site_tensor_host(size_t site, size_t dims3[3]);

    // TensorCI2::from_tensor_train (tensorci/src/conversion.rs:66-121): replaces the I/J sets, site tensors,
    // pivot errors and max_sample_value of this (freshly constructed) object.  `tt` is left untouched.
    void assign_from_tensor_train(const TensorTrain& tt, const FromTensorTrainOptions& options);

    // state (public like the accessors of the reference)
    size_t n_;
    std::vector<size_t> local_dims;
    std::vector<IndexSet> i_set, j_set;
    std::vector<std::vector<IndexSet>> i_set_history, j_set_history;
    std::vector<double> bond_errors, pivot_errors;
    double max_sample_value = 0.0;
    std::vector<DevCore> cores;
    std::vector<size_t> ranks_hist;
    std::vector<double> errors_hist;
    int termination = T4A_GPU_TCI2_MAX_ITERATIONS;
    std::vector<std::array<size_t, 3>> last_sweep_shapes;
    size_t shard_rank = 0, shard_world = 1;
    bool keep_site_tensors = false;
    Engine eng;

private:
    enum class FnKind { None, Builtin, Callback };
    FnKind fn_kind_ = FnKind::None;
    FnDevice fn_dev_{};
    std::vector<uint64_t> weights_; // n_acc * total
    std::vector<size_t> offset_;
    size_t total_ = 0;
    t4a_gpu_batch_eval_fn cb_ = nullptr;
    void* cb_ctx_ = nullptr;

    // index helpers
    IndexSet kronecker_i(size_t p) const;
    IndexSet kronecker_j(size_t p) const;
    static void union_extras(IndexSet& base, const IndexSet& extras);
    void accumulate(const IndexSet& set, size_t first_site, std::vector<uint64_t>& acc) const;

    // Evaluate f into a device matrix: out[ia + a.count*ib] = f(index with a's digits at sites
    // [a0, a0+a.width) and b's digits at [b0, b0+b.width)).  If d_maxbits != nullptr the kernel also
    // atomically maxes bits(sqrt(v*v)) into it.
    void eval_matrix(const IndexSet& a, size_t a0, const IndexSet& b, size_t b0, double* d_out,
                     unsigned long long* d_maxbits, const std::vector<uint64_t>* acc_a = nullptr,
                     const std::vector<uint64_t>* acc_b = nullptr);
    // in_place: hand out the pinned arena addresses themselves (no device copy): small bonds, read once by the fused kernel
    void stage_accumulators(const IndexSet& a, size_t a0, const IndexSet& b, size_t b0, const std::vector<uint64_t>* acc_a,
                            const std::vector<uint64_t>* acc_b, const uint64_t** d_ra, const uint64_t** d_rb, bool in_place = false);
    std::vector<double> eval_points_host(const std::vector<uint32_t>& idx, size_t n_pts);
    void require_fn() const;

    struct BondOut {
        LuciResult lu;
    };
    // One side (rows or columns) of the candidate matrix of a bond, built ahead of time: the combined index set
    // (Kronecker product + history extras) and, for built-in functions, its integer accumulators.
    struct SidePrep {
        bool valid = false;
        size_t bond = 0;
        bool cols = false;
        IndexSet set;
        std::vector<uint64_t> acc;
    };
    SidePrep prep_;                 // filled by the overlap hook while the previous bond's kernels run
    struct Prefetch {
        bool wanted = false;
        size_t bond = 0;
        bool cols = false;
        const IndexSet* extra = nullptr;
        long fill_site = -1;        // a site whose I/J sets are already final: its fill accumulators can be built now
        bool flush_fill = false;    // issue the stream operations of the previous half-sweep's (deferred) fill
    } prefetch_;
    // accumulators of fill_site_tensors (J_b, kron_i(b), I_{b+1}) built ahead of time, site by site, while the bond
    // updates of the same half-sweep are running; only valid inside optimize() between the bond loop and its fill
    struct FillAcc {
        bool valid = false;
        std::vector<uint64_t> accJ, accK, accI;
    };
    std::vector<FillAcc> fill_cache_;
    bool fill_cache_trusted_ = false;
    bool fill_defer_requested_ = false;
    std::vector<std::function<void()>> fill_deferred_; // stream operations of a prepared, not yet issued fill
    void flush_deferred_fill();
    void issue_fill_ops(std::vector<std::function<void()>>& ops, const std::vector<uint64_t>& sig);
    std::vector<uint64_t> fill_deferred_sig_, fill_last_sig_, fill_graph_sig_;
    hipGraphExec_t fill_graph_exec_ = nullptr;
    bool fill_graph_broken_ = false;
    void prepare_fill_site(size_t b);
    void invalidate_fill_cache();                    // set by the sweep loop: which side of which bond is independent of the current one
    void build_side(size_t bond, bool cols, const IndexSet& extra, SidePrep& out) const;
    LuciResult luci_on_sets(const IndexSet& is, const IndexSet& js, const RrLUOptions& o, bool need_factors,
                            const std::vector<uint64_t>* acc_rows = nullptr, const std::vector<uint64_t>* acc_cols = nullptr);
    LuciResult rook_on_sets(const IndexSet& is, const IndexSet& js, const RrLUOptions& o);
    RookWork rook_work_;
    void update_pivots(size_t b, bool left_orthogonal, const TCI2Options& options, const IndexSet& extra_i,
                       const IndexSet& extra_j);
    void sweep1site_at_bond(size_t b, bool forward, double rel_tol, double abs_tol, size_t max_bond_dim,
                            bool update_tensors);
    void set_core_from_left(size_t site, size_t left_dim, size_t site_dim, const LuciResult& lu);
    void set_core_from_right(size_t site, size_t site_dim, size_t right_dim, const LuciResult& lu);
    void set_core_zero(size_t site, size_t l, size_t s, size_t r);
    void update_pivot_errors(const std::vector<double>& e);
    std::vector<std::vector<uint32_t>> find_global_pivots(double abs_tol, const TCI2Options& o, uint64_t& rng_state);

    // device scratch
    DevBuf<uint64_t> d_rowacc_, d_colacc_;
    PinBuf<uint64_t> h_acc_;
    size_t acc_used_ = 0;
    DevBuf<unsigned long long> d_maxbits_;
    DevBuf<uint32_t> d_idx_;
    DevBuf<double> d_vals_;
    DevBuf<TtCoreDesc> d_coredesc_;
    // fill_site_tensors scratch
    DevBuf<double> d_fillA_, d_fillB_;
    DevBuf<int> d_fillpiv_, d_fillinfo_;
    DevBuf<LuProblem> d_lup_;
    DevBuf<TrsmProblem> d_trp_;
    DevBuf<unsigned long long> d_fillmax_;
    EventTimer ev_pi_, ev_fill_;
    hipStream_t fill_stream_ = nullptr, import_stream_ = nullptr;
    bool import_inflight_ = false;
    hipEvent_t export_event_ = nullptr, import_event_ = nullptr;
    bool fill_inflight_ = false, fill_timed_ = false;
    std::vector<size_t> fill_solved_sites_;
    PinBuf<int> h_fillinfo_;
    PinBuf<uint64_t> h_fillacc_;
    DevBuf<uint64_t> d_fillacc_;
    PinBuf<char> h_fillprob_;
};

} // namespace t4a
