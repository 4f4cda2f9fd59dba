// tci2.hpp — host-side mirror of tensor4all-tensorci's TensorCI2 driver
// (crates/tensor4all-tensorci/src/tensorci2.rs) on top of the gfx950 engine.
// Same names, argument meaning and error behaviour as the reference; the I/J index sets are the
// master copy on the host (flat digit tables), all matrices and site tensors live on the device.
#pragma once

#include <memory>
#include <vector>

#include "engine.hpp"
#include "stdrng.hpp"
#include "pishard.hpp"
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
    pool::IdleScope idle_scope_; // (first member: destroyed last, see ~Tci2)
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
    // the same for up to eight handles at once, in lock-step from the calling thread (one XCD per handle)
    static void optimize_group(const std::vector<Tci2*>& handles, const TCI2Options& options, bool final_sweep1site);
    // fill_site_tensors of several handles: all fills issued (each on its own fill stream), then all completed
    static void fill_site_tensors_group(const std::vector<Tci2*>& handles);
    struct OptRun { // one optimize() call in progress
        TCI2Options options;
        bool final_sweep1site = false;
        std::vector<size_t> nglobal_hist;
        StdRng rng;   // rand 0.9 StdRng::seed_from_u64 (tensorci2.rs:1653-1657); one stream for the whole optimisation
        bool pending_fill = false; // fill_site_tensors of the last iteration: accumulators / stream operations not yet issued
        size_t iter = 0;
        bool done = false;
        // the iteration between opt_iter_start and opt_iter_finish
        double norm = 1.0, abs_tol = 0.0;
        bool is_forward = true, chained = false, fill_ahead = false;
        long ext_idx = -1;
        size_t flush_at_fwd = 0, flush_at_bwd = 0;
        // small-problem engine (tci2_small.hip): offered every run unless the caller drives several handles in lock-step
        bool allow_small = true;
        bool small_complete = false; // the engine finished the whole call (iterations and, when asked for, the final 1-site sweep)
    };
    // the small-problem engine: the whole optimize loop of a small problem in ONE launch (kernels_small.hip)
    bool small_enabled = true;
    // [0] optimize calls it completed [1] iterations it ran [2] runs it handed back to the general path [3] calls that were not eligible
    std::array<uint64_t, 4> small_stats{{0, 0, 0, 0}};
    uint64_t small_last_clocks_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // last launch (SmallOutHeader::clocks)
    // candidate matrices up to this many rows / columns stay in the launch.  Default 16: measured (profiles/r06_small_engine_reach.txt), the
    // 32 x 32 tile (sixteen entries per lane) is slower than the general path — a rank-8 run completes in 2.1 ms inside the launch against
    // 1.64 ms outside — so runs that grow past rank ~5 are handed over early; t4a_gpu_tci2_set_chain bit 5 selects 32.
    int small_tile_max = 16;
    bool small_stamps = false; // the launch stamps its phases (diagnostic: t4a_gpu_tci2_set_chain bit 3)
    int small_last_reason_ = 0;
    bool small_engine_eligible(const TCI2Options& options) const;
    bool small_engine_run(OptRun& r);
    void opt_begin(OptRun& r);
    void opt_iter_issue_pending_fill(OptRun& r);
    void opt_end_issue_fill(OptRun& r); // the last iteration's fill, issued without waiting (first half of opt_end; a group issues all of them first)
    bool opt_iter_start(OptRun& r, bool defer_launch = false); // defer_launch: the chain is prepared, not launched (optimize_group)
    void opt_iter_finish(OptRun& r);
    void opt_end(OptRun& r);
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
    // the I/J sets (or the history) were changed behind the driver's back (C-ABI setters, conversion): the device-side
    // tables of the bond chain are re-uploaded before they are used again
    void mark_sets_changed()
    {
        chain_.tables_valid = false;
        chain_.digits_stale = false; // (the caller has just written digit tables: they are the master copy now)
    }
    // statistics of the device-side bond chain: [0] half-sweeps run as a chain [1] bonds run in chains [2] chains that fell back
    // to the per-bond path part-way [3] half-sweeps that were not eligible
    std::array<uint64_t, 5> chain_stats{{0, 0, 0, 0, 0}};
    // [0] chained sweeps (2-site half-sweeps and 1-site sweeps) that ran as ONE persistent workgroup (kernels_chain.hip,
    // chain_walk_kernel) [1] 1-site sweeps (sweep1site) that ran as a chain [2] 1-site sweeps that were not eligible and ran bond
    // by bond [3] chained 1-site sweeps that fell back to the per-bond path part-way
    std::array<uint64_t, 4> chain_stats_ext{{0, 0, 0, 0}};
    const uint64_t* fill_stats() const { return fill_stats_; }
    const RookWork& rook_work() const { return rook_work_; } // [fills issued asynchronously, graph replays, graph captures]
    bool chain_enabled = true;  // false: every half-sweep runs bond by bond (A/B measurements, tests)
    bool chain_verify = false;  // true: after every chain the device tables are read back and compared with the host's sets
    bool chain_event_timing = false; // profiling: rrLU launches of a chain are timed with HIP events around each launch instead of
                                     // the kernel's own time stamps (two more packets per bond on the stream: calibration runs only)

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
    // history of the index sets (tensorci2.rs:1675-1689: the sets at the start of every iteration; only the newest two are ever
    // read).  After a device-side bond chain the digit tables are not materialised: an entry then carries the codes of the sets.
    struct HistEntry {
        std::vector<IndexSet> is, js;   // widths and counts always valid, digit tables only when digits_valid
        bool digits_valid = true;
        std::vector<uint64_t> code;     // [2][n][cap] (I then J), copied from the pinned mirror of the device tables
        size_t cap = 0;
        uint64_t serial = 0;
    };
    std::vector<HistEntry> history;
    void clear_history()
    {
        history.clear();
        chain_.hist_serial += 16;
    }
    // I / J digit tables up to date (after a bond chain only codes and counts are current: decoded here on demand)
    void sync_digits();
    std::vector<double> bond_errors, pivot_errors;
    double max_sample_value = 0.0;
    std::vector<DevCore> cores;
    std::vector<size_t> ranks_hist;
    std::vector<double> errors_hist;
    int termination = T4A_GPU_TCI2_MAX_ITERATIONS;
    std::vector<std::array<size_t, 3>> last_sweep_shapes;
    size_t shard_rank = 0, shard_world = 1;
    PiShard pi_shard; // column-block shard of callback-evaluated candidate matrices over a process group (pishard.hpp)
    bool keep_site_tensors = false;
    size_t callback_threads = 1; // opt-in: host threads that evaluate one candidate matrix through the callback concurrently
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
    std::vector<std::function<void()>> fill_pre_ops_; // (diagnosis switch T4A_FILL_GRAPH_NO_COPY)
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
    std::vector<std::vector<uint32_t>> find_global_pivots(double abs_tol, const TCI2Options& o, StdRng& rng);

    // ---- device-side bond chain (tci2_chain.hip, kernels_chain.hip) ----
    struct ChainState {
        bool tables_valid = false;   // device tables == host master copy of I / J (digit tables or mirror)
        bool digits_stale = false;   // the digit tables of i_set / j_set are older than the mirror (their counts are current)
        size_t cap = 0;              // entries per site in every table
        int n_acc = 0;               // accumulators per entry the tables were laid out for
        uint64_t hist_serial = 0;    // serial number of the newest history entry (advanced by every push)
        uint64_t snap_serial[2] = {~0ull, ~0ull}; // which history entry a device snapshot slot holds
        int mcur = 0;                // which of the two pinned mirrors holds the current sets
        DevBuf<uint64_t> tab;        // [6 families: I, J, snapshot 0 (I, J), snapshot 1 (I, J)] codes + accumulators
        DevBuf<int> cnt;             // [6][n]
        PinBuf<uint64_t> mtab;       // [2 mirrors][2 families]: pinned host copies of I, J, written by the device
        PinBuf<int> mcnt;            // [2][2][n]
        DevBuf<uint64_t> weights;    // [K][total]
        DevBuf<int> siteinfo;        // [2][n]: local dimensions, weight offsets
        bool weights_valid = false;
        DevBuf<uint64_t> ind, dep;   // independent-side lists of all bonds; dependent-side list of the current bond
        DevBuf<int> ind_cnt, rowmap;
        DevBuf<double> pi, spec[2];  // candidate matrix of the first bond; speculative candidate matrices (alternating)
        DevBuf<char> blocks;         // per-bond result blocks
        PinBuf<char> hblocks;
        DevBuf<int> dims;            // per-bond {M, N, poison, lda}
        PinBuf<int> hdims;
        // a chain that has been enqueued and not yet finished
        bool inflight = false;
        bool forward = true;
        bool in_optimize = false;
        long ext_idx = -1;
        size_t chi = 0;
        std::vector<size_t> order;
        std::vector<ChainRrluPlan> plans;
        std::vector<unsigned> tokens;
        // a 1-site sweep as a chain (sweep1site, tensorci2.rs:865-1050): set by sweep1site() around chain_enqueue / chain_finish
        bool one_site = false;       // the independent side is the table itself, no extras; both tolerances apply
        bool one_factors = false;    // update_tensors: every bond's factored matrix is kept and its LUCI factor becomes the site tensor
        double abs_tol = 0.0;
        DevBuf<double> factors, urows; // [n_bonds][factors_stride] factored matrices; finished rows of U of the bond in flight
        size_t factors_stride = 0;
        bool cores_batched = false;  // the chain in flight writes the site tensors of its low-rank bonds itself (one launch behind it)
        bool last_core_launched = false, last_core_ok = false; // forward 1-site chain: the last site's tensor was evaluated behind the chain / the whole chain completed, it is valid
        DevBuf<unsigned long long> walk_dbg; // diagnostic phase times of the persistent half-sweep (T4A_WALK_DEBUG)
        bool walked = false;         // the chain in flight is a persistent half-sweep
        bool prep_dbg = false;       // ... or a launched chain whose preparation kernels stamp their phases (T4A_PREP_DEBUG)
        unsigned walk_token = 1;     // completion tokens of the persistent half-sweep (bond k of a walk: base + k)
        ChainBlock proto;
        bool timed = false, timed_events = false;
        std::vector<hipEvent_t> t0, t1; // per bond: around the rrLU launch (chain_event_timing)
        // a half-sweep that has been prepared (plans, buffers, tables, kernel constants) and waits for its launch: on its own
        // (chain_launch) or together with the chains of other handles (chain_group_launch)
        bool prepared = false;
        ChainCommon common;
        std::vector<size_t> dep_ub, ind_ub;
        bool use_extras = false;
        size_t ind_cap = 0, cap_side = 0;
        double tol = 0.0;
        hipStream_t wait_stream = nullptr; // the stream whose completion chain_finish waits for (a group chain: the leader's)
        int group_role = 0;                // 0: a chain of its own, 1: leader of a group chain (holds the chip), 2: member
        hipEvent_t group_ev = nullptr;     // orders the group's stream behind this handle's own
        DevBuf<ChainGroupSlot> gslots;     // leader: the per-handle constants of the group's half-sweep
        PinBuf<ChainGroupSlot> hgslots;
    } chain_;
    bool chain_usable(const TCI2Options& options) const;
    ChainTab chain_tab(int family) const;    // 0: I, 1: J, 2 + 2 s: snapshot s of I, 3 + 2 s: snapshot s of J
    ChainTab chain_mirror(int which, int family) const; // pinned mirror `which` of family 0 (I) / 1 (J)
    void chain_layout(size_t cap);
    void chain_upload_current();
    void chain_upload_hist(HistEntry& e, int slot);
    uint64_t code_of(const uint32_t* v, size_t first_site, size_t width, bool prefix) const;
    void decode_set(IndexSet& s, const uint64_t* codes, size_t first_site, bool prefix) const;
    void hist_digits(HistEntry& e);
    // enqueues the whole half-sweep (false: not eligible, nothing was done); chain_finish() waits for it and takes over the
    // results (falling back to update_pivots for the bonds after one that did not complete)
    // launch = false: everything but the launches (chain_.prepared); chain_launch() or chain_group_launch() must follow.
    bool chain_enqueue(bool forward, const TCI2Options& options, long ext_idx, bool in_optimize, bool launch = true);
    void chain_launch();
    // the prepared half-sweeps of several handles as ONE chain of launches (every kernel serves all of them, every handle's
    // rrLU on its own XCD); falls back to chain_launch() per handle when the chains do not line up
    static void chain_group_launch(const std::vector<Tci2*>& handles);
    void chain_finish(const TCI2Options& options);
    void chain_abort() noexcept; // after an exception between launch and finish: wait, release the XCD, drop the chain's results
    void prepare_fill_site_from_mirror(size_t b);
    bool fill_no_main_sync_ = false; // the next fill does not depend on work of the main stream (bond chain: no cores written there)

    // small-problem engine: input / result blocks (pinned), scratch site tensors
    PinBuf<char> small_in_, small_out_;
    DevBuf<double> small_scratch_;
    unsigned small_token_ = 0;
    // device scratch
    DevBuf<uint64_t> d_rowacc_, d_colacc_;
    PinBuf<uint64_t> h_acc_;
    size_t acc_used_ = 0;
    DevBuf<unsigned long long> d_maxbits_;
    DevBuf<uint32_t> d_idx_;
    DevBuf<double> d_vals_;
    // host-callback path: the index buffer handed to the callback and the values it returns, grow-only (a 700 x 700 candidate matrix of a
    // 30-site problem is 59 MB of indices: as a fresh std::vector per bond it was zero-filled and page-faulted in every time)
    std::unique_ptr<uint32_t[]> cb_idx_;
    size_t cb_idx_cap_ = 0;
    PinBuf<double> cb_vals_;
    DevBuf<TtCoreDesc> d_coredesc_;
    // fill_site_tensors scratch
    PinBuf<uint64_t> h_rookacc_; // rook_on_sets: pinned staging of the row / column accumulators
    DevBuf<double> d_fillA_, d_fillB_;
    DevBuf<int> d_fillpiv_, d_fillinfo_;
    DevBuf<LuProblem> d_lup_;
    DevBuf<TrsmProblem> d_trp_;
    DevBuf<unsigned long long> d_fillmax_;
    EventTimer ev_pi_, ev_fill_;
    hipStream_t fill_stream_ = nullptr, import_stream_ = nullptr;
    bool import_inflight_ = false;
    bool cores_shared_legacy_stream_ = false; // ... and stream 0 (the legacy default stream) was the consumer / producer of one: no graph replay of the fill (issue_fill_ops)
    uint64_t fill_stats_[3] = {0, 0, 0};       // fills issued through issue_fill_ops, graph replays, graph captures
    bool cores_shared_async_ = false; // an asynchronous export / import of cores was requested on this handle: fills are issued directly, not replayed from a graph (issue_fill_ops)
public:
    bool fill_graph_relaxed = false;  // opt-in: replay the fill graph on a handle with shared site tensors unless the legacy / a blocking stream took part
private:
    hipEvent_t export_event_ = nullptr, import_event_ = nullptr;
    bool fill_inflight_ = false, fill_timed_ = false;
    std::vector<size_t> fill_solved_sites_;
    PinBuf<int> h_fillinfo_;
    PinBuf<uint64_t> h_fillacc_;
    DevBuf<uint64_t> d_fillacc_;
    PinBuf<char> h_fillprob_;
};

} // namespace t4a
