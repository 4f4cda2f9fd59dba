// engine.hpp — device engine shared by the dense C-ABI entry points and the TCI2 driver:
// owns the HIP stream, work buffers and the "Π -> rrLU -> LUCI factors" pipeline
// (matrix_luci_factors_from_matrix, tensor4all-core/src/matrix_luci.rs:366-374).
#pragma once

#include <array>
#include <functional>
#include <thread>
#include <map>
#include <limits>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace t4a {

struct RrLUOptions { // core/src/matrixlu.rs:688-708
    size_t max_bond_dim = std::numeric_limits<size_t>::max();
    double rel_tol = 1e-14;
    double abs_tol = 0.0;
    bool left_orthogonal = true;
};

struct LuciResult {
    int M = 0, N = 0;
    int rank = 0;
    std::vector<int> row_perm, col_perm;   // full permutations (RrLU::row_permutation / col_permutation)
    std::vector<double> pivot_errors;      // rank + 1 entries (RrLU::pivot_errors, matrixlu.rs:361-365)
    double last_error = 0.0;
    double abs_max = 0.0;                  // max sqrt(v*v) over the input matrix
    bool has_factors = false;              // d_left (M x rank), d_right (rank x N) valid on the engine
};

// Candidate matrix given implicitly: entry (i, j) = fn(rowacc[i] + colacc[j]) (device accumulators [count][n_acc]).
// The register-resident rrLU kernel builds it straight into its registers; other paths materialise it first.
struct FusedPi {
    FnDevice fn;
    const uint64_t* d_rowacc;
    const uint64_t* d_colacc;
    bool host_resident = false; // the accumulators sit in pinned host memory (small bonds: read in place by the fused kernel)
};

struct Profile {
    double v[T4A_GPU_PROFILE_SLOTS] = {0};
    bool enabled = false;
};

// process-wide arbiter of the persistent multi-workgroup rrLU kernels (engine.hip)
bool xcd_disabled();
void xcd_disable();
int xcd_version(); // generation of the single-XCD rrLU kernel (2: kernels_rrlu_xcd2.hip; the first generation was retired in round 6)
void rrlu_xcd_launch_v(int version, const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream);
void rrlu_xcd_group_launch_v(int version, const RrluXcdPlan& plan, const RrluXcdGroupArgs& args, bool tie_row_major, hipStream_t stream);
int xcd_assign();
int xcd_plan_max_w();
struct XcdArbiter {
    class Lock {
    public:
        Lock() = default;
        Lock(const Lock&) = delete;
        Lock& operator=(const Lock&) = delete;
        ~Lock() { release(); }
        void acquire(int xcc); // xcc >= 0: that XCD; -1: the whole chip
        void release();
    private:
        int held_ = 0;
        std::thread::id owner_; // the thread the XCDs are booked under (a Lock may be released by another one)
    };
};

// rrLU launch of one bond of a device-side bond chain (tci2_chain.hip): planned for upper bounds, dimensions read on the device
struct ChainRrluPlan {
    int kind = 0;        // 1: single-workgroup register kernel   2: single-XCD kernel
    bool fused = false;  // kind 1: the candidate matrix is built in the registers from the accumulators (no matrix in memory)
    RrluRegPlan reg;
    RrluXcdPlan xcd;
    int kM = 0, kN = 0;  // upper bounds of the matrix the KERNEL sees (transposed for a right-orthogonal factorisation)
    int code = 0;        // profile code of the kernel instantiation (Engine::variant_stats_)
};
struct ChainBlock {      // packed result block of one bond: [dresult 2 f64][iresult 4 i32][pivot values][row perm][col perm]
    char* dev = nullptr;
    char* host = nullptr; // pinned mirror, same layout
    size_t off_piv = 32, off_rp = 0, off_cp = 0, off_ts = 0, bytes = 0; // off_ts: two u64 time stamps of the kernel (0: none)
};

class Engine {
public:
    Engine();
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    hipStream_t stream() const { return stream_; }
    int num_cus() const { return num_cus_; }
    void sync() { T4A_HIP(hipStreamSynchronize(stream_)); }

    // Device matrix buffers reused across calls.
    double* pi(size_t count)
    {
        d_pi_.reserve(count);
        return d_pi_.get();
    }
    double* lu_buf() { return d_lu_.get(); }
    double* left() { return d_left_.get(); }
    double* right() { return d_right_.get(); }
    void reserve_factors(size_t left_count, size_t right_count)
    {
        d_left_.reserve(left_count > 0 ? left_count : 1);
        d_right_.reserve(right_count > 0 ? right_count : 1);
    }

    // Runs rrLU on the M x N column-major matrix at d_a (device) and optionally builds the LUCI factors.
    // `want_lu_copy` additionally keeps the factored matrix (permuted coordinates) in lu_buf().
    // `fused` != nullptr: d_a is ignored (may be null) and the matrix is defined by the accumulators.
    LuciResult luci(const double* d_a, int M, int N, const RrLUOptions& opts, bool need_factors, bool want_lu_copy,
                    const FusedPi* fused = nullptr);

    // RrLU::left(true) / RrLU::right(true) (matrixlu.rs:263-326) of the factorisation kept by the last
    // luci(..., want_lu_copy = true): left() is M x rank, right() is rank x N afterwards.
    void lu_permuted_factors(const LuciResult& r, bool left_orth);

    // thin SVD (svd_backend, tensorbackend/src/backend.rs:709): d_u M x k, d_s k, d_vt k x N with k = min(M, N);
    // one-sided Jacobi.  Throws INVALID_ARGUMENT for non-finite input.
    void svd(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt);
    void svd_plain(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt); // the Jacobi iteration on the matrix as it is
    void svd_in_range(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt); // svd() behind the scaling of its input
    // thin QR (qr_backend, backend.rs:742): d_q M x k, d_r k x N; Householder.
    void qr(const double* d_a, int M, int N, double* d_q, double* d_r);

    // ---- bond chain: launches without a host round trip (dimensions in device memory) ----
    bool chain_plan(int kM, int kN, ChainRrluPlan* out) const;
    // before the first launch of a chain: mailbox capacity for every plan, the XCD (or the whole chip) reserved until chain_end()
    void chain_begin(const std::vector<ChainRrluPlan>& plans, size_t reserve_mailbox_words = 0);
    // rrLU of one bond: `left` as in RrLUOptions::left_orthogonal; d_a: the kernel's matrix (already transposed for !left; read
    // through d_rowmap with leading dimension d_dims[3] when d_rowmap != nullptr) unless the plan is fused (then `fused` holds
    // the accumulators of the KERNEL's rows and columns); d_dims: {M, N, poison, lda} of the MATRIX on the device.  Returns the
    // completion token the kernel writes to iresult[3] of the device block (and to int word 7 of the host mirror).
    // spec (single-XCD plans only, may be null): the candidate matrix of the NEXT bond for the launch's pass-through workgroups.
    unsigned chain_rrlu(const ChainRrluPlan& pl, bool left, const double* d_a, const int* d_rowmap, const FusedPi* fused, const int* d_dims,
                        size_t max_bond_dim, double rel_tol, double abs_tol, const ChainBlock& blk, const XcdSpecArgs* spec, double* d_aout = nullptr, double* d_urows = nullptr);
    void chain_end();
    // Group chain (tci2_chain.hip): several handles advance in lock step, one rrLU launch per bond for all of them, handle i on
    // XCD `slot` i.  chain_group_plan: the single-XCD plan every member uses for a bond (made for the largest upper bounds in
    // the group; no single-workgroup plans: a small bond costs a group launch, i.e. an eighth of it per handle).
    // chain_group_reserve: this member's mailbox for the plans (no lock: the group's leader reserves the chip with
    // chain_group_lock / chain_end).  chain_group_args: the argument block of this member for one bond (advances the member's
    // salt and ticket base exactly like a launch of its own); the caller launches the assembled blocks with rrlu_xcd_group_launch.
    static bool chain_group_plan(int kM, int kN, ChainRrluPlan* out);
    void chain_group_reserve(const std::vector<ChainRrluPlan>& plans, size_t reserve_mailbox_words, hipStream_t order_stream);
    void chain_group_lock() { chain_lock_.acquire(-1); }
    unsigned chain_group_args(const ChainRrluPlan& pl, bool left, const double* d_a, const int* d_dims, size_t max_bond_dim, double rel_tol,
                              double abs_tol, const ChainBlock& blk, int slot, RrluXcdArgs* out, hipStream_t order_stream);
    int xcc() const { return xcc_; }
    void set_xcc(int xcc) { xcc_ = xcc & 7; } // (optimize_group: handle i of a group works on XCD i)

    // Host work that does not depend on the running factorisation: executed once, after the kernels of the next luci()
    // call have been enqueued and before the host blocks on them (then cleared).
    std::function<void()> overlap_hook;

    Profile prof;
    // rrLU launch statistics per kernel instantiation: code -> {ms, launches, algorithmic bytes, pivot steps}
    std::map<int, std::array<double, 4>> variant_stats_;

    // scratch for the TCI2 driver
    DevBuf<double> d_tmp, d_tmp2;

private:
    void build_factors(const LuciResult& r, bool left_orth);
public:
    // factors_from_rrlu on a factored matrix that sits somewhere else (the per-bond buffers of a chained 1-site sweep): left() /
    // right() hold the factors afterwards, on the engine's stream.  d_rowperm / d_colperm: the full permutations on the device.
    void build_factors_from(const double* d_lu, const int* d_rowperm, const int* d_colperm, int M, int N, int rank, bool left_orth);
private:

    XcdArbiter::Lock chain_lock_;
    hipStream_t stream_ = nullptr;
    int num_cus_ = 0;
    unsigned rrlu_salt_ = 0;
    // single-XCD rrLU kernel: elected XCD, mailboxes, monotonic ticket counter
    int xcc_ = 0;
    bool xcd_retry_v1_ = false; // luci(): this call re-runs a factorisation a register kernel gave up on (non-finite values) on the chip-wide kernels
    unsigned xcd_salt_ = 0, xcd_ticket_base_ = 0, xcd_ticket_base_multi_ = 0;
    void xcd_take_tickets(const RrluXcdPlan& plan, RrluXcdArgs& a);
    DevBuf<unsigned long long> d_xkeys_; // mailbox of the single-XCD kernel: keys, then column slots
    DevBuf<unsigned> d_xticket_;
    DevBuf<double> d_xurows_;
    bool header_clean_ = false, keys_clean_ = false;
    unsigned done_token_ = 0;              // completion tokens of single-workgroup launches (never 0)
    DevBuf<uint64_t> d_accstage_;          // device copy of host-resident accumulators when a plan cannot read them in place
    char* header_ptr_ = nullptr;
    int key_parity_ = 0;
    DevBuf<double> d_pi_, d_lu_, d_left_, d_right_, d_w1_, d_w2_, d_at_;
    DevBuf<char> d_out_;
    PinBuf<char> h_out_;
    int* d_rowperm_ptr_ = nullptr;
    int* d_colperm_ptr_ = nullptr;
    std::vector<int> h_ints_;
    std::vector<double> h_dbls_;
    DevBuf<unsigned long long> d_keys_, d_cols_, d_rkeys_, d_rcols_, d_stamps_;
    DevBuf<TrsmProblem> d_trsm_;
    PinBuf<TrsmProblem> h_trsm_;
    DevBuf<int> d_gints_;      // HBM-resident rrLU workspace
    DevBuf<double> d_gdbls_;
    EventTimer ev_rrlu_, ev_fac_;
#ifdef T4A_RRLU_TRACE
    DevBuf<unsigned long long> d_trace_; // T4A_RRLU_TRACE_FILE (diagnostic builds only)
    int trace_dumps_ = 0;
#endif
    // SVD / QR workspaces
    DevBuf<double> d_sw_, d_sv_, d_su_, d_svs_, d_ssig_;
    DevBuf<double> d_pa_, d_pq_, d_pr_, d_pl_, d_pul_, d_pvl_, d_pu_; // QR-preconditioned SVD: A', Q, R, L = R^T, factors of L, U'
    size_t svd_sweeps_last_ = 0;
    DevBuf<int> d_sflags_;
    DevBuf<unsigned long long> d_sabs_; // largest magnitude of an SVD input (bit pattern)
    DevBuf<double> d_sscaled_;          // the input scaled by a power of two (only when its largest entry is outside 2^-200 .. 2^200)
};

// triangle extraction helper kernels (engine.hip)
void tri_extract_launch(const double* in, int ldi, int rows, int cols, int keep_lower, int unit_diag, double* out,
                        int ldo, hipStream_t stream);

} // namespace t4a
