// engine.hip — device pipeline "matrix -> rrLU -> LUCI factors"
// (matrix_luci_factors_from_matrix / factors_from_rrlu, tensor4all-core/src/matrix_luci.rs:256-290,366-374).
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>

namespace t4a {

namespace {
thread_local std::string g_last_error;

__global__ void __launch_bounds__(256) tri_extract_kernel(const double* __restrict__ in, int ldi, int rows, int cols,
                                                          int keep_lower, int unit_diag, double* __restrict__ out,
                                                          int ldo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    for (int j = blockIdx.y; j < cols; j += gridDim.y) {
        double v;
        if (i == j)
            v = unit_diag ? 1.0 : in[(size_t)j * ldi + i];
        else if ((keep_lower && i > j) || (!keep_lower && i < j))
            v = in[(size_t)j * ldi + i];
        else
            v = 0.0;
        out[(size_t)j * ldo + i] = v;
    }
}
} // namespace

// ---- process-wide arbiter of the persistent multi-workgroup rrLU kernels ----
// Their workgroups spin on each other's mailboxes, so all of them must be resident at once: two launches that compete for the
// same compute units would starve each other until the bounded spins give up.  A single-XCD launch owns one of the eight XCDs
// for its duration, a chip-wide launch (old register kernel, LDS kernel) owns all of them.
namespace {
constexpr int kXcdSharedMaxW = 28; // single-XCD plans up to this many workgroups leave room for other handles' pass-through workgroups
std::atomic<int> g_xcd_next{0};
std::atomic<int> g_live_engines{0};
std::atomic<bool> g_xcd_disabled{false};
} // namespace
bool xcd_disabled()
{
    static const bool env_off = std::getenv("T4A_NO_XCD") != nullptr;
    return env_off || g_xcd_disabled.load(std::memory_order_relaxed);
}
void xcd_disable()
{
    if (!g_xcd_disabled.exchange(true, std::memory_order_relaxed))
        std::fprintf(stderr, "[t4a] single-XCD rrLU path disabled for this process (workgroup placement did not hold or a launch "
                             "timed out); the chip-wide kernels take over\n");
}

// The single-XCD kernel is kernels_rrlu_xcd2.hip (the first generation was retired in round 6).  A launch that meets non-finite values
// gives up with code 2; the caller then runs the chip-wide kernels for that matrix (they implement the NaN-incumbent rule).
int xcd_version() { return 2; }
void rrlu_xcd_launch_v(int version, const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream)
{
    if (plan.wg == 2) rrlu_w1_launch(plan, args, stream);
    else if (plan.wg) rrlu_wg_launch(plan, args, stream);
    else if (plan.big()) rrlu_xcd2m_launch(plan, args, stream); // (second generation only: rrlu_xcd_make_plan hands such plans out with allow_big)
    else rrlu_xcd2_launch(plan, args, stream);
}
void rrlu_xcd_group_launch_v(int version, const RrluXcdPlan& plan, const RrluXcdGroupArgs& args, bool tie_row_major, hipStream_t stream)
{
    if (plan.wg == 2) rrlu_w1_group_launch(plan, args, tie_row_major, stream);
    else if (plan.wg) rrlu_wg_group_launch(plan, args, tie_row_major, stream);
    else rrlu_xcd2_group_launch(plan, args, tie_row_major, stream);
}

// Placement census (once per process, first Engine): the single-XCD kernel assumes that workgroup b of a grid lands on XCD
// b % 8 (tools/xcd_bench.hip).  On a partitioned device (CPX / NPS modes), under a CU mask or with fewer than 8 XCCs that does
// not hold and every launch would spin until its bounded polls give up; better to find out with a 10 us kernel.
namespace {
__global__ void xcd_census_kernel(unsigned* counts)
{
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        atomicAdd(&counts[v & 0xF], 1u);
        if ((v & 0xF) != (blockIdx.x & 7u)) atomicAdd(&counts[16], 1u);
    }
}
std::once_flag g_census_once;
void xcd_census()
{
    if (g_xcd_disabled.load(std::memory_order_relaxed) || std::getenv("T4A_NO_XCD")) return;
    unsigned* d = nullptr;
    unsigned h[17] = {0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return;
    bool ok = hipMemset(d, 0, sizeof(h)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(xcd_census_kernel, dim3(256), dim3(64), 0, nullptr, d);
        ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) {
        (void)hipGetLastError();
        return;
    }
    bool even = true;
    for (int x = 0; x < 8; ++x) even &= (h[x] == 32u);
    // (block b on XCC b % 8 is what tools/xcd_bench.hip measured; the kernel itself only needs grid / 8 workgroups per XCC)
    if (!even) xcd_disable();
}
} // namespace
int xcd_assign() { return g_xcd_next.fetch_add(1, std::memory_order_relaxed) & 7; }
// A launch has 8 W workgroups of which 7 W pass through the other XCDs and need a free compute unit there for a moment.  While
// this engine is the only one, a plan may fill all 32 compute units of its XCD (and reserves the whole chip for the launch);
// as soon as several engines are alive, plans stop at kXcdSharedMaxW workgroups, so that every engine only ever reserves its own
// XCD — before round 3 a chain with one 29-workgroup plan in it held the chip-wide reservation for its whole half-sweep and
// eight handles on eight host threads ran one after the other (34.3 ms per patch against 38.6 ms for one alone).
int xcd_plan_max_w() { return g_live_engines.load(std::memory_order_relaxed) > 1 ? kXcdSharedMaxW : 32; }
// Ownership is kept IN THE ARBITER (owner thread + recursion count per XCD under one mutex), not in thread-local counters: one
// thread may drive several handles whose reservations overlap in time — optimize_group launches one chain per handle when the chains
// do not line up, and a handle whose chain fell back runs chip-wide kernels while its siblings still hold their XCDs (ADVICE round 3:
// that used to self-deadlock on non-recursive mutexes) — and a Lock may be released by another thread than the one that acquired it
// (a handle handed to a worker thread: ADVICE round 4 — with thread-local counts that drove a count negative and left the mutex
// locked for good).  A thread that already holds ANY part of the chip never blocks for more, neither for a single XCD nor for the
// whole chip: it takes what is free and shares the rest — the arbiter is a performance device, every persistent kernel has bounded
// spins and a fallback — so two threads that each hold an XCD and want each other's (more than eight live engines: xcd_assign wraps)
// or both want the whole chip cannot wait for each other.
namespace {
std::mutex g_arb_mutex;
std::condition_variable g_arb_cv;
std::thread::id g_arb_owner[8];
int g_arb_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}
void XcdArbiter::Lock::acquire(int xcc)
{
    release();
    const std::thread::id me = std::this_thread::get_id();
    std::unique_lock<std::mutex> lk(g_arb_mutex);
    auto holds_some = [&] {
        for (int i = 0; i < 8; ++i)
            if (g_arb_count[i] > 0 && g_arb_owner[i] == me) return true;
        return false;
    };
    auto take = [&](int i) {
        g_arb_owner[i] = me;
        ++g_arb_count[i];
        held_ |= 1 << i;
    };
    owner_ = me;
    if (xcc >= 0) {
        const int i = xcc & 7;
        if (g_arb_count[i] > 0 && g_arb_owner[i] == me) take(i);
        else if (holds_some()) {
            if (g_arb_count[i] == 0) take(i); // (somebody else's XCD otherwise: shared for the time being, never waited for)
        } else {
            g_arb_cv.wait(lk, [&] { return g_arb_count[i] == 0; });
            take(i);
        }
    } else {
        const bool some = holds_some();
        for (int i = 0; i < 8; ++i) { // fixed order: no deadlock between two chip-wide owners that start from nothing
            if (g_arb_count[i] > 0 && g_arb_owner[i] == me) take(i);
            else if (some) {
                if (g_arb_count[i] == 0) take(i);
            } else {
                g_arb_cv.wait(lk, [&] { return g_arb_count[i] == 0; });
                take(i);
            }
        }
    }
}
void XcdArbiter::Lock::release()
{
    if (!held_) return;
    {
        std::lock_guard<std::mutex> lk(g_arb_mutex);
        for (int i = 7; i >= 0; --i)
            if ((held_ & (1 << i)) && g_arb_count[i] > 0 && g_arb_owner[i] == owner_) --g_arb_count[i]; // (whichever thread calls: the Lock knows its owner)
        held_ = 0;
    }
    g_arb_cv.notify_all();
}

void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error_ref() { return g_last_error; }

void require_device()
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw Error(T4A_GPU_NO_DEVICE,
                    "no HIP device available: the tensor4all-rs MI355X backend has no CPU fallback (hipGetDeviceCount: " +
                        std::string(hipGetErrorString(e)) + ")");
}

void tri_extract_launch(const double* in, int ldi, int rows, int cols, int keep_lower, int unit_diag, double* out,
                        int ldo, hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((rows + 255) / 256, cols < 1024 ? cols : 1024);
    hipLaunchKernelGGL(tri_extract_kernel, grid, dim3(256), 0, stream, in, ldi, rows, cols, keep_lower, unit_diag, out,
                       ldo);
}

Engine::Engine()
{
    require_device();
    num_cus_ = pool::compute_units();
    // highest priority: the latency-critical chain of bond updates gets its own hardware queue, separate from the
    // (lowest-priority) fill_site_tensors stream that runs beside it.  (Streams are recycled through the process-wide cache.)
    stream_ = pool::stream_get(1);
    ev_rrlu_.init();
    ev_fac_.init();
    static const int xcc_env = std::getenv("T4A_XCD_ID") ? std::atoi(std::getenv("T4A_XCD_ID")) : -1;
    // (T4A_XCD_ID=8: an XCC id no workgroup reports — exercises the fallback to the chip-wide kernel in the tests)
    xcc_ = xcc_env >= 0 ? (xcc_env > 8 ? (xcc_env & 7) : xcc_env) : xcd_assign();
    g_live_engines.fetch_add(1, std::memory_order_relaxed);
    std::call_once(g_census_once, xcd_census);
}

Engine::~Engine()
{
    g_live_engines.fetch_sub(1, std::memory_order_relaxed);
    if (stream_) pool::stream_put(stream_, 1); // (synchronises it)
}

LuciResult Engine::luci(const double* d_a, int M, int N, const RrLUOptions& opts, bool need_factors, bool want_lu_copy,
                        const FusedPi* fused)
{
    const double* const d_a_in = d_a;
    LuciResult r;
    r.M = M;
    r.N = N;
    r.row_perm.resize(M);
    r.col_perm.resize(N);
    for (int i = 0; i < M; ++i) r.row_perm[i] = i;
    for (int j = 0; j < N; ++j) r.col_perm[j] = j;
    if (M == 0 || N == 0) {
        r.rank = 0;
        r.last_error = 0.0; // n >= min(nr, nc) = 0 -> error = 0 (matrixlu.rs:811)
        r.pivot_errors = {0.0};
        r.has_factors = need_factors;
        return r;
    }
    // the register- and LDS-resident kernels pack positions into 16 bits; larger matrices go to the HBM-resident kernel
    const bool huge = M > 65535 || N > 65535;
    if ((size_t)M * (size_t)N > ((size_t)1 << 33))
        throw Error(T4A_GPU_NOT_IMPLEMENTED, "rrLU: matrices with more than 2^33 entries are not supported");

    size_t ms = opts.max_bond_dim;
    if (ms > (size_t)M) ms = M;
    if (ms > (size_t)N) ms = N;
    const int max_steps = (int)ms;

    // one packed result block: [dresult 2 f64][iresult 4 i32][pivot_vals max_steps f64][row_perm M i32][col_perm N i32]
    // -> one memset of the 32-byte header, one device-to-host copy per bond
    const size_t off_piv = 32;
    const size_t off_rp = off_piv + sizeof(double) * (size_t)(max_steps > 0 ? max_steps : 1);
    const size_t off_cp = off_rp + sizeof(int) * (size_t)M;
    const size_t out_bytes = (off_cp + sizeof(int) * (size_t)N + 7) / 8 * 8;
    d_out_.reserve(out_bytes);
    h_out_.reserve(out_bytes);
    double* d_dres = reinterpret_cast<double*>(d_out_.get());
    int* d_ires = reinterpret_cast<int*>(d_out_.get() + 16);
    double* d_pivvals = reinterpret_cast<double*>(d_out_.get() + off_piv);
    int* d_rowperm = reinterpret_cast<int*>(d_out_.get() + off_rp);
    int* d_colperm = reinterpret_cast<int*>(d_out_.get() + off_cp);
    d_rowperm_ptr_ = d_rowperm;
    d_colperm_ptr_ = d_colperm;
    const bool keep_lu = need_factors || want_lu_copy;
    if (keep_lu) d_lu_.reserve((size_t)M * N);
    // the register kernel resets the header at its end; everything else (first use, regrown buffer, LDS kernel, a launch
    // that did not finish cleanly) clears it here
    const bool header_clean = header_clean_ && d_out_.get() == header_ptr_;
    header_clean_ = false;
    if (!header_clean) T4A_HIP(hipMemsetAsync(d_out_.get(), 0, 32, stream_));
    static const bool want_stamps = std::getenv("T4A_RRLU_STAMPS") != nullptr;
    static const bool force_lds = std::getenv("T4A_RRLU_IMPL") != nullptr && std::string(std::getenv("T4A_RRLU_IMPL")) == "lds";
    static const bool force_global = std::getenv("T4A_RRLU_IMPL") != nullptr && std::string(std::getenv("T4A_RRLU_IMPL")) == "global";
    if (want_stamps) {
        d_stamps_.reserve(24);
        T4A_HIP(hipMemsetAsync(d_stamps_.get(), 0, 24 * sizeof(unsigned long long), stream_));
    }

    // Fast path: register-resident kernel (left-orthogonal only).  A right-orthogonal factorisation is the
    // left-orthogonal one of A^T with row-major tie order; rows/columns swap roles on the way out.
    const bool left = opts.left_orthogonal;
    const int kM = left ? M : N, kN = left ? N : M;
    RrluRegPlan rplan;
    RrluXcdPlan xplan;
    static const bool force_reg = std::getenv("T4A_RRLU_IMPL") != nullptr && std::string(std::getenv("T4A_RRLU_IMPL")) == "reg";
    // first choice: all workgroups on one XCD (exchange through that XCD's L2); disabled for good once a launch timed out
    // (a retry after non-finite values: none of the register kernels of this family — they hand such matrices back — but the chip-wide
    // register / LDS / global kernels below, which implement the NaN-incumbent rule)
    const int xcd_v = xcd_version();
    // matrices that fit one workgroup: the LDS-exchange kernel (not on a retry after non-finite values: it does not handle them)
    static const long wg_min = std::getenv("T4A_WG_MIN") ? std::atol(std::getenv("T4A_WG_MIN")) : 64;
    const bool use_wg = !huge && !force_lds && !force_global && !force_reg && !xcd_retry_v1_ && (long)kM * kN > wg_min &&
                        (rrlu_w1_make_plan(kM, kN, &xplan, 0) || rrlu_wg_make_plan(kM, kN, &xplan, 0));
    const bool use_xcd = use_wg || (!huge && !force_lds && !force_global && !force_reg && !xcd_disabled() &&
                         !xcd_retry_v1_ &&
                         (rrlu_xcd_make_plan(kM, kN, &xplan, false, xcd_plan_max_w()) || rrlu_xcd_make_plan(kM, kN, &xplan, false, 32, true)));
    const bool use_reg = !use_xcd && !huge && !force_lds && !force_global && rrlu_reg_make_plan(kM, kN, num_cus_, &rplan);
    bool fuse = false;
    bool xcd_src_transposed = false;
    if (fused) {
        fuse = use_reg && rplan.RPT * rplan.CPT <= RRLU_FUSED_MAX_VALUES;
        if (!fuse) { // this plan cannot build the matrix in registers: materialise it like the Π kernel would
            const uint64_t* ra = fused->d_rowacc;
            const uint64_t* ca = fused->d_colacc;
            if (fused->host_resident) { // (the Π kernel re-reads accumulators many times: never over PCIe)
                const size_t nr = (size_t)M * fused->fn.n_acc, nc = (size_t)N * fused->fn.n_acc;
                d_accstage_.reserve(nr + nc);
                stage_copy_launch(ra, d_accstage_.get(), nr, stream_);
                stage_copy_launch(ca, d_accstage_.get() + nr, nc, stream_);
                ra = d_accstage_.get();
                ca = d_accstage_.get() + nr;
            }
            double* buf = pi((size_t)M * N);
            if (use_xcd && !left) { // the kernel works on the transpose: evaluate it in that layout straight away
                pi_eval_launch(fused->fn, ca, N, ra, M, buf, N, false, nullptr, stream_);
                xcd_src_transposed = true;
            } else {
                pi_eval_launch(fused->fn, ra, M, ca, N, buf, M, false, nullptr, stream_);
            }
            d_a = buf;
        }
    }
    int plan_W = 1, plan_T = 0, plan_code = 0;
    bool mirrored = false;
    unsigned spin_token = 0u; // non-zero: a single-workgroup launch that announces its completion in the pinned block
    if (prof.enabled) T4A_HIP(hipEventRecord(ev_rrlu_.a, stream_));
    XcdArbiter::Lock xcd_lock; // multi-workgroup persistent kernels need their compute units to themselves
    if (use_xcd) {
        const double* src = d_a;
        if (!left && !xcd_src_transposed) {
            d_at_.reserve((size_t)M * N);
            transpose_launch(d_a, M, N, M, d_at_.get(), N, stream_);
            src = d_at_.get();
        }
        const size_t need_keys = rrlu_xcd_keys_bytes(xplan) / sizeof(unsigned long long);
        const size_t need_cols = rrlu_xcd_cols_bytes(xplan, kM) / sizeof(unsigned long long);
        ++xcd_salt_;
        if (need_keys + need_cols > d_xkeys_.cap || xcd_salt_ > 65535u || !d_xticket_.get()) {
            // granules carry launch-salted tags: clear the mailbox (keys, then column slots) whenever it moves or the 16-bit
            // salt wraps.  (Plans differ in where the column slots start; a stale granule of another plan still carries
            // another launch's salt.)
            d_xkeys_.reserve(need_keys + need_cols);
            d_xticket_.reserve(16);
            T4A_HIP(hipMemsetAsync(d_xkeys_.get(), 0, d_xkeys_.cap * sizeof(unsigned long long), stream_));
            T4A_HIP(hipMemsetAsync(d_xticket_.get(), 0, 16 * sizeof(unsigned), stream_));
            xcd_ticket_base_ = xcd_ticket_base_multi_ = 0;
            xcd_salt_ = 1;
        }
        RrluXcdArgs a;
        a.A = src;
        a.Aout = keep_lu ? d_lu_.get() : nullptr;
        if (keep_lu) d_xurows_.reserve((size_t)(max_steps > 0 ? max_steps : 1) * kN);
        a.urows = keep_lu ? d_xurows_.get() : nullptr;
        a.M = kM;
        a.N = kN;
        a.max_steps = max_steps;
        a.rel_tol = opts.rel_tol;
        a.abs_tol = opts.abs_tol;
        a.tie_row_major = left ? 0 : 1;
        a.out_transposed = left ? 0 : 1;
        a.W = xplan.W;
        a.xcc = xcc_;
        xcd_take_tickets(xplan, a); // exactly grid / 8 workgroups of a launch land on one XCD (the one-workgroup kernel takes no tickets)
        a.row_perm = left ? d_rowperm : d_colperm;
        a.col_perm = left ? d_colperm : d_rowperm;
        a.iresult = d_ires;
        a.dresult = d_dres;
        a.pivot_vals = d_pivvals;
        a.keys = d_xkeys_.get();
        a.salt = xcd_salt_;
        static const double xspec = diag_env("T4A_XCD_SPECFRAC") ? std::atof(diag_env("T4A_XCD_SPECFRAC")) : 0.8;
        a.spec_frac = xspec;
        a.stamps = want_stamps ? d_stamps_.get() : nullptr;
        std::memset(h_out_.get(), 0, 32);
        a.h_block = reinterpret_cast<unsigned long long*>(h_out_.get());
        a.block_u64 = (int)(out_bytes / 8);
        a.dims = nullptr;
        a.dims_swap = 0;
        a.rowmap = nullptr;
        std::memset(&a.spec, 0, sizeof(a.spec));
        a.ts_u64 = 0;
        mirrored = true;
        // a launch has 8 W workgroups of which 7 W pass through the other XCDs and need a free compute unit there for a moment:
        // two handles that each fill (nearly) all 32 compute units of their XCD would block each other's dispatch
        xcd_lock.acquire((xplan.W > kXcdSharedMaxW || xplan.K > 1) ? -1 : xcc_);
        rrlu_xcd_launch_v(xcd_v, xplan, a, stream_);
        plan_W = xplan.W;
        plan_T = 512;
        plan_code = (xplan.wg == 2 ? 300000 + xplan.RPT * 1000 : xplan.wg ? 200000 + xplan.RPT * 1000 : xplan.big() ? 400000 + xplan.K * 10000 + xplan.RPT * 100 : 100000 + xplan.RPT * 100) +
                    xplan.CPT * 10 + (a.tie_row_major ? 4 : 0); // (the one-workgroup / one-wave kernels under their own codes)
    } else if (use_reg) {
        const double* src = d_a;
        if (!left && !fuse) {
            d_at_.reserve((size_t)M * N);
            transpose_launch(d_a, M, N, M, d_at_.get(), N, stream_);
            src = d_at_.get();
        }
        bool keys_ready = false;
        unsigned long long* keys_this = nullptr;
        unsigned long long* keys_next = nullptr;
        int keys_half = 0;
        if (rplan.W > 1) {
            // mailbox granules carry launch-salted tags: zero the buffers whenever they are (re)allocated or the
            // 16-bit salt wraps, so that no stale granule can ever match a live tag (no per-launch memset).
            // The key table exists twice: a launch polls one copy and clears the other one for its successor.
            const size_t need_keys = 2 * (rrlu_reg_keys_bytes(rplan) / sizeof(unsigned long long));
            const size_t need_cols = rrlu_reg_cols_bytes(rplan, kM) / sizeof(unsigned long long);
            ++rrlu_salt_;
            if (need_keys > d_rkeys_.cap || need_cols > d_rcols_.cap || rrlu_salt_ > 65535u || !keys_clean_) {
                d_rkeys_.reserve(2 * (size_t)(2 * 256 * 2)); // room for any W up to 256 in both copies
                d_rkeys_.reserve(need_keys);
                d_rcols_.reserve(need_cols);
                T4A_HIP(hipMemsetAsync(d_rkeys_.get(), 0, d_rkeys_.cap * sizeof(unsigned long long), stream_));
                T4A_HIP(hipMemsetAsync(d_rcols_.get(), 0, d_rcols_.cap * sizeof(unsigned long long), stream_));
                rrlu_salt_ = 1;
                keys_clean_ = true;
            }
            keys_half = (int)(d_rkeys_.cap / 2);
            keys_this = d_rkeys_.get() + (size_t)key_parity_ * keys_half;
            keys_next = d_rkeys_.get() + (size_t)(1 - key_parity_) * keys_half;
            key_parity_ = 1 - key_parity_;
            keys_ready = true;
        }
        RrluRegArgs a;
        a.A = src;
        a.Aout = keep_lu ? d_lu_.get() : nullptr;
        a.M = kM;
        a.N = kN;
        a.max_steps = max_steps;
        a.rel_tol = opts.rel_tol;
        a.abs_tol = opts.abs_tol;
        a.tie_row_major = left ? 0 : 1;
        a.out_transposed = left ? 0 : 1;
        a.W = rplan.W;
        a.TR = rplan.TR;
        a.TC = rplan.TC;
        a.row_perm = left ? d_rowperm : d_colperm;
        a.col_perm = left ? d_colperm : d_rowperm;
        a.iresult = d_ires;
        a.dresult = d_dres;
        a.pivot_vals = d_pivvals;
        a.keys = keys_this;
        a.cols = d_rcols_.get();
        a.keys_next = keys_next;
        a.keys_next_u64 = keys_ready ? keys_half : 0;
        a.salt = rrlu_salt_;
        static const int col_delay = diag_env("T4A_RRLU_COLDELAY") ? std::atoi(diag_env("T4A_RRLU_COLDELAY")) : 0;
        a.col_delay = col_delay;
        // measured optimum of the poller's initial sleep (tools/probe_delay.py): 12 units below ~100 workgroups, 14 above
        static const int poll_delay_env = diag_env("T4A_RRLU_POLLDELAY") ? std::atoi(diag_env("T4A_RRLU_POLLDELAY")) : -1;
        a.poll_delay = poll_delay_env >= 0 ? poll_delay_env : (rplan.W > 100 ? 14 : 12);
        static const int ncopy_env = diag_env("T4A_RRLU_NCOPY") ? std::atoi(diag_env("T4A_RRLU_NCOPY")) : 1;
        a.ncopy = ncopy_env < 1 ? 1 : (ncopy_env > RRLU_MAX_COPIES ? RRLU_MAX_COPIES : ncopy_env);
        static const int spec_env = diag_env("T4A_RRLU_SPEC") ? std::atoi(diag_env("T4A_RRLU_SPEC")) : 2;
        a.spec = spec_env < 0 ? 0 : (spec_env > 2 ? 2 : spec_env);
        static const double spec_frac_env = diag_env("T4A_RRLU_SPECFRAC") ? std::atof(diag_env("T4A_RRLU_SPECFRAC")) : 0.8;
        a.spec_frac = spec_frac_env;
        static const int key16_env = diag_env("T4A_RRLU_KEY16") ? std::atoi(diag_env("T4A_RRLU_KEY16")) : 1;
        a.key16 = key16_env; // bit 0: 16-byte key loads, bit 1: 16-byte key store
        a.spin_limit = 1u << 20;
        a.stamps = want_stamps ? d_stamps_.get() : nullptr;
        // results land in the pinned mirror straight from the kernel: no device-to-host copy afterwards
        std::memset(h_out_.get(), 0, 32);
        a.fused = fuse ? 1 : 0;
        if (fuse) { // the kernel's rows are the matrix rows (left) or the matrix columns (right-orthogonal = transposed)
            a.rowacc = left ? fused->d_rowacc : fused->d_colacc;
            a.colacc = left ? fused->d_colacc : fused->d_rowacc;
            a.fn = fused->fn;
        } else {
            a.rowacc = nullptr;
            a.colacc = nullptr;
            std::memset(&a.fn, 0, sizeof(a.fn));
        }
        a.h_block = reinterpret_cast<unsigned long long*>(h_out_.get());
        a.block_u64 = (int)(out_bytes / 8);
        a.done_token = 0u;
        a.dims = nullptr;
        a.dims_swap = 0;
        a.dev_token = 0u;
        a.rowmap = nullptr;
        a.ts_u64 = 0;
        static const bool no_token_spin = diag_env("T4A_NO_TOKEN_SPIN") != nullptr;
        if (rplan.W == 1 && !prof.enabled && !no_token_spin) {
            if (++done_token_ == 0u) ++done_token_;
            a.done_token = done_token_;
            spin_token = done_token_;
        }
        a.trace = nullptr;
#ifdef T4A_RRLU_TRACE
        static const char* trace_file = std::getenv("T4A_RRLU_TRACE_FILE");
        const size_t trace_words = (size_t)rplan.W * (1 + 4 * (size_t)(max_steps + 1));
        const bool tracing = trace_file && rplan.W > 1 && kM > 600 && trace_dumps_ >= 20 && trace_dumps_ < 24;
        if (trace_file && rplan.W > 1 && kM > 600) ++trace_dumps_;
        if (tracing) {
            d_trace_.reserve(trace_words);
            T4A_HIP(hipMemsetAsync(d_trace_.get(), 0, trace_words * 8, stream_));
            a.trace = d_trace_.get();
        }
#endif
        mirrored = true;
        keys_clean_ = false; // becomes true again once the launch is known to have finished cleanly
        if (rplan.W > 1) xcd_lock.acquire(-1); // chip-wide persistent launch: every XCD
        rrlu_reg_launch(rplan, a, stream_, true);
        plan_W = rplan.W;
        plan_T = rplan.T;
        plan_code = rplan.RPT * 1000 + rplan.CPT * 10 + (a.tie_row_major ? 4 : 0) + (rplan.W == 1 ? 2 : 0) + ((rplan.TR % 64) == 0 ? 1 : 0);
    } else if (huge || force_global || rrlu_make_plan(M, N, num_cus_).lds_bytes > 160 * 1024) {
        // neither the register file nor the LDS of the chip holds this matrix: HBM-resident kernel pair per pivot step
        const int gb = rrlu_global_blocks(M, N);
        d_gints_.reserve(rrlu_global_int_words(M, N, gb));
        d_gdbls_.reserve(rrlu_global_double_words(M, N, gb));
        RrluGlobalArgs a{};
        a.A = d_a;
        a.Aout = keep_lu ? d_lu_.get() : nullptr;
        a.M = M;
        a.N = N;
        a.max_steps = max_steps;
        a.rel_tol = opts.rel_tol;
        a.abs_tol = opts.abs_tol;
        a.left_orth = opts.left_orthogonal ? 1 : 0;
        a.row_perm = d_rowperm;
        a.col_perm = d_colperm;
        a.iresult = d_ires;
        a.dresult = d_dres;
        a.pivot_vals = d_pivvals;
        rrlu_global_launch(a, d_gints_.get(), d_gdbls_.get(), stream_);
        plan_W = gb;
        plan_T = 256;
        plan_code = -3; // HBM-resident kernel
    } else {
        const RrluPlan plan = rrlu_make_plan(M, N, num_cus_);
        if (plan.W > 1) {
            d_keys_.reserve(rrlu_keys_bytes(plan) / sizeof(unsigned long long));
            d_cols_.reserve(rrlu_cols_bytes(plan, M) / sizeof(unsigned long long));
        }
        RrluArgs a;
        a.A = d_a;
        a.Aout = keep_lu ? d_lu_.get() : nullptr;
        a.M = M;
        a.N = N;
        a.max_steps = max_steps;
        a.rel_tol = opts.rel_tol;
        a.abs_tol = opts.abs_tol;
        a.left_orth = opts.left_orthogonal ? 1 : 0;
        a.W = plan.W;
        a.cpw = plan.cpw;
        a.Mld = plan.Mld;
        a.row_perm = d_rowperm;
        a.col_perm = d_colperm;
        a.iresult = d_ires;
        a.dresult = d_dres;
        a.pivot_vals = d_pivvals;
        a.keys = d_keys_.get();
        a.cols = d_cols_.get();
        a.spin_limit = 1u << 20;
        a.stamps = want_stamps ? d_stamps_.get() : nullptr;
        if (plan.W > 1) xcd_lock.acquire(-1);
        rrlu_launch(plan, a, stream_);
        plan_W = plan.W;
        plan_T = plan.T;
        plan_code = plan.W == 1 ? -1 : -2; // LDS kernel (single / multi workgroup)
    }
    T4A_HIP(hipGetLastError());
    if (prof.enabled) T4A_HIP(hipEventRecord(ev_rrlu_.b, stream_));

    if (!mirrored) T4A_HIP(hipMemcpyAsync(h_out_.get(), d_out_.get(), out_bytes, hipMemcpyDeviceToHost, stream_));
    if (overlap_hook) { // the device is busy for the next ~millisecond: do the caller's independent host work now
        std::function<void()> hook;
        hook.swap(overlap_hook);
        hook();
    }
    bool completed = false;
    if (spin_token != 0u) {
        // small bond: spin on the token the (only) workgroup writes after its last word to the host; the stream itself is
        // not waited for (everything that follows on it is ordered behind the kernel anyway).  Bounded: a launch that gave
        // up returns without a token and is picked up by the synchronisation below.
        volatile unsigned* tok = reinterpret_cast<volatile unsigned*>(h_out_.get() + 16) + 3;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; ++it) {
            if (*tok == spin_token) {
                completed = true;
                break;
            }
            if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!completed) T4A_HIP(hipStreamSynchronize(stream_));
    xcd_lock.release();
    if (use_xcd && (xcd_v == 2 || use_wg) && reinterpret_cast<const int*>(h_out_.get() + 16)[1] == 2) {
        // the kernel met a NaN / an infinity (input or overflow): the chip-wide kernels implement the NaN-incumbent rule of
        // matrixlu.rs:480-519 (the first-generation single-XCD kernel, kept for this case until round 5, is gone); every workgroup of the
        // launch was elected normally, the tickets stay valid
        T4A_HIP(hipMemsetAsync(d_out_.get(), 0, 32, stream_));
        header_clean_ = false;
        {
            static const bool dbg = std::getenv("T4A_CHAIN_DEBUG") != nullptr;
            if (dbg) std::fprintf(stderr, "[t4a luci] %d x %d: the %s kernel met non-finite values, re-running with the chip-wide kernels\n", M, N, use_wg ? "one-workgroup" : "single-XCD");
        }
        xcd_retry_v1_ = true;
        try {
            LuciResult r1 = luci(d_a_in, M, N, opts, need_factors, want_lu_copy, fused);
            xcd_retry_v1_ = false;
            return r1;
        } catch (...) {
            xcd_retry_v1_ = false;
            throw;
        }
    }
    if (use_xcd && (reinterpret_cast<const int*>(h_out_.get() + 16)[1] != 0 || reinterpret_cast<const int*>(h_out_.get() + 16)[3] != (int)xcd_salt_)) {
        // the placement assumption of the single-XCD kernel did not hold (or another process holds the compute units):
        // never try it again in this process and run this factorisation with the chip-wide kernels
        xcd_disable();
        T4A_HIP(hipMemsetAsync(d_out_.get(), 0, 32, stream_));
        T4A_HIP(hipMemsetAsync(d_xticket_.get(), 0, 16 * sizeof(unsigned), stream_));
        xcd_ticket_base_ = xcd_ticket_base_multi_ = 0;
        header_clean_ = false;
        return luci(d_a_in, M, N, opts, need_factors, want_lu_copy, fused);
    }

#ifdef T4A_RRLU_TRACE
    if (mirrored && d_trace_.get() && std::getenv("T4A_RRLU_TRACE_FILE") && trace_dumps_ > 20 && trace_dumps_ <= 24 && plan_W > 1) {
        const size_t words = (size_t)plan_W * (1 + 4 * (size_t)(max_steps + 1));
        std::vector<unsigned long long> ht(words);
        T4A_HIP(hipMemcpy(ht.data(), d_trace_.get(), words * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(std::getenv("T4A_RRLU_TRACE_FILE"), "ab")) {
            const long long hdr[4] = {M, N, plan_W, max_steps};
            std::fwrite(hdr, sizeof(hdr), 1, f);
            std::fwrite(ht.data(), 8, words, f);
            std::fclose(f);
        }
    }
#endif
    // host views of the packed block (hp: [4 + M + N] ints, hr: [2 + max_steps] doubles, as before)
    std::vector<int>& hpv = h_ints_;
    hpv.resize(4 + (size_t)M + N);
    std::memcpy(hpv.data(), h_out_.get() + 16, 4 * sizeof(int));
    std::memcpy(hpv.data() + 4, h_out_.get() + off_rp, sizeof(int) * (size_t)M);
    std::memcpy(hpv.data() + 4 + M, h_out_.get() + off_cp, sizeof(int) * (size_t)N);
    std::vector<double>& hrv = h_dbls_;
    hrv.resize(2 + (size_t)(max_steps > 0 ? max_steps : 1));
    std::memcpy(hrv.data(), h_out_.get(), 2 * sizeof(double));
    std::memcpy(hrv.data() + 2, h_out_.get() + off_piv, sizeof(double) * (size_t)(max_steps > 0 ? max_steps : 1));
    const int* hp = hpv.data();
    const double* hr = hrv.data();
    float rrlu_ms_this = 0.f;
    if (prof.enabled) {
        T4A_HIP(hipEventElapsedTime(&rrlu_ms_this, ev_rrlu_.a, ev_rrlu_.b));
        prof.v[0] += rrlu_ms_this;
        prof.v[1] += 1.0;
    }
    if (want_stamps) {
        unsigned long long hs[24];
        T4A_HIP(hipMemcpy(hs, d_stamps_.get(), sizeof(hs), hipMemcpyDeviceToHost));
        if (use_xcd && xplan.wg == 2) {
            const double st = hp[0] > 0 ? (double)hp[0] : 1.0;
            std::fprintf(stderr, "[rrlu stamps w1] M=%d N=%d columns=%d steps=%d cycles/step: search=%.0f stop+divide+clear=%.0f tables=%.0f update=%.0f | launch (cycles): "
                                 "load+init=%llu steps=%llu write-out=%llu | shader clock %.0f MHz\n",
                         M, N, xplan.CPT, hp[0], hs[0] / st, hs[1] / st, hs[2] / st, hs[3] / st, hs[16], hs[17], hs[18], hs[21] ? 100.0 * (double)hs[20] / (double)hs[21] : 0.0);
        } else if (use_xcd) {
            const double st = hp[0] > 0 ? (double)hp[0] : 1.0;
            std::fprintf(stderr, "[rrlu stamps xcd] M=%d N=%d W=%d steps=%d cycles/step: pass=%.0f search=%.0f publish=%.0f | prefetch=%.0f keys=%.0f pick=%.0f slow+stop=%.0f recwr=%.0f barB=%.0f | record=%.0f "
                                 "tables+u=%.0f colwait=%.0f divide=%.0f barC=%.0f lread=%.0f | pollspins=%llu | launch (cycles): election=%llu load+init=%llu steps=%llu write-out=%llu\n",
                         M, N, plan_W, hp[0], hs[0] / st, hs[1] / st, hs[2] / st, hs[6] / st, hs[8] / st, hs[14] / st, hs[15] / st, hs[9] / st, hs[3] / st, hs[10] / st, hs[11] / st, hs[12] / st,
                         hs[4] / st, hs[13] / st, hs[7] / st, hs[5], hs[16], hs[17], hs[18], hs[19]);
        } else {
            std::fprintf(stderr, "[rrlu stamps %s] M=%d N=%d W=%d T=%d steps=%d | s0=%llu s1=%llu s2=%llu s3=%llu s4=%llu pollspins=%llu colspins=%llu s7=%llu "
                                 "(cycles, wg0/thread0; lds: publish,poll,colfetch,pass,reduce; reg: pass,reduce,publish,poll,fetch)\n",
                         use_reg ? "reg" : "lds", M, N, plan_W, plan_T, hp[0], hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
        }
    }
    if (hp[1] != 0)
        throw Error(T4A_GPU_KERNEL_TIMEOUT, "rrLU kernel: inter-workgroup hand-off timed out (bounded spin gave up)");
    if (mirrored) { // the kernel ran to its end: header and next key table were left clean by workgroup 0
        header_clean_ = true;
        header_ptr_ = d_out_.get();
        keys_clean_ = true;
    }
    r.rank = hp[0];
    if (max_steps == 0) {
        // while-loop never entered: lu.error stays NaN unless the matrix is "full rank" (min(M,N) == 0)
        r.last_error = std::numeric_limits<double>::quiet_NaN();
    } else {
        r.last_error = hr[0];
    }
    {
        unsigned long long bits;
        std::memcpy(&bits, &hr[1], sizeof(bits));
        double am;
        std::memcpy(&am, &bits, sizeof(am));
        r.abs_max = am;
    }
    for (int i = 0; i < M; ++i) r.row_perm[i] = hp[4 + i];
    for (int j = 0; j < N; ++j) r.col_perm[j] = hp[4 + M + j];
    r.pivot_errors.resize(r.rank + 1);
    for (int k = 0; k < r.rank; ++k) r.pivot_errors[k] = std::sqrt(hr[2 + k] * hr[2 + k]);
    r.pivot_errors[r.rank] = r.last_error;
    // work model
    {
        double bytes = 8.0 * M * N, flops = 0.0;
        for (int k = 0; k < r.rank; ++k) {
            const double mr = (double)(M - k - 1), nr_ = (double)(N - k - 1);
            bytes += 16.0 * mr * nr_;
            flops += 2.0 * mr * nr_ + mr;
        }
        prof.v[8] += r.rank;
        prof.v[9] += bytes;
        prof.v[10] += flops;
        if (prof.enabled) { // per kernel-instantiation statistics (the dominant one feeds bench.py's roofline)
            auto& vs = variant_stats_[plan_code];
            vs[0] += rrlu_ms_this;
            vs[1] += 1.0;
            vs[2] += bytes;
            vs[3] += r.rank;
        }
    }
    if (hp[2] != 0) throw Error(T4A_GPU_NAN_ENCOUNTERED, "NaN encountered in L or U of the rrLU factorisation");

    if (need_factors) {
        if (prof.enabled) T4A_HIP(hipEventRecord(ev_fac_.a, stream_));
        build_factors(r, opts.left_orthogonal);
        if (prof.enabled) {
            T4A_HIP(hipEventRecord(ev_fac_.b, stream_));
            T4A_HIP(hipStreamSynchronize(stream_));
            float ms_f = 0.f;
            T4A_HIP(hipEventElapsedTime(&ms_f, ev_fac_.a, ev_fac_.b));
            prof.v[6] += ms_f;
            prof.v[7] += 1.0;
        }
        const double rr = r.rank;
        prof.v[10] += (M - rr) * rr * rr + 2.0 * rr * rr * N;
        r.has_factors = true;
    }
    return r;
}

// ------------------------------------------------------------------------------------------------
// bond chain: the rrLU launches of a half-sweep without a host round trip (tci2_chain.hip)
// ------------------------------------------------------------------------------------------------
// ticket counters of a launch: single-XCD plans count on d_xticket_[0] (only the workgroups on the elected XCD take one), plans over
// K > 1 XCDs on d_xticket_[8 + xcd] (EVERY workgroup of such a launch takes one from its XCD's counter); either way a counter advances
// by grid / 8 per launch
void Engine::xcd_take_tickets(const RrluXcdPlan& plan, RrluXcdArgs& a)
{
    if (plan.K > 1) {
        a.ticket = d_xticket_.get() + 8;
        a.ticket_base = xcd_ticket_base_multi_;
        xcd_ticket_base_multi_ += (unsigned)(plan.grid / 8);
    } else {
        a.ticket = d_xticket_.get();
        a.ticket_base = xcd_ticket_base_;
        if (!plan.wg) xcd_ticket_base_ += (unsigned)(plan.grid / 8);
    }
}

bool Engine::chain_plan(int kM, int kN, ChainRrluPlan* out) const
{
    if (kM < 1 || kN < 1 || kM > 65535 || kN > 65535) return false;
    ChainRrluPlan pl;
    pl.kM = kM;
    pl.kN = kN;
    static const bool no_single = diag_env("T4A_CHAIN_NO_SINGLE") != nullptr;
    // tiny matrices keep the fused single-workgroup plan (the candidate matrix is built in the registers: no extra launch);
    // everything else that fits one workgroup takes the LDS-exchange kernel, with 63 more workgroups for the speculative
    // candidate matrix of the next bond
    static const long wg_min = std::getenv("T4A_WG_MIN") ? std::atol(std::getenv("T4A_WG_MIN")) : 64;
    if ((long long)kM * kN > wg_min && (rrlu_w1_make_plan(kM, kN, &pl.xcd, 63) || rrlu_wg_make_plan(kM, kN, &pl.xcd, 63))) {
        pl.kind = 2;
        pl.code = (pl.xcd.wg == 2 ? 300000 : 200000) + pl.xcd.RPT * 1000 + pl.xcd.CPT * 10;
        *out = pl;
        return true;
    }
    if (!no_single && (long long)kM * kN <= 64 * 64 && rrlu_reg_make_plan(kM, kN, num_cus_, &pl.reg) && pl.reg.W == 1) {
        pl.kind = 1;
        pl.fused = pl.reg.RPT * pl.reg.CPT <= RRLU_FUSED_MAX_VALUES;
        pl.code = pl.reg.RPT * 1000 + pl.reg.CPT * 10 + 2 + ((pl.reg.TR % 64) == 0 ? 1 : 0); // (+4 for the row-major tie order: chain_rrlu)
        *out = pl;
        return true;
    }
    // (shapes that only fit with more than kXcdSharedMaxW workgroups keep their plan: the launch then reserves the whole chip)
    if (xcd_disabled() || !(rrlu_xcd_make_plan(kM, kN, &pl.xcd, true, xcd_plan_max_w()) || rrlu_xcd_make_plan(kM, kN, &pl.xcd, true, 32, true)))
        return false;
    pl.kind = 2;
    pl.code = (pl.xcd.big() ? 400000 + pl.xcd.K * 10000 : 100000) + pl.xcd.RPT * 100 + pl.xcd.CPT * 10;
    *out = pl;
    return true;
}

void Engine::chain_begin(const std::vector<ChainRrluPlan>& plans, size_t reserve_mailbox_words)
{
    size_t need = 0;
    int max_w = 0;
    for (const ChainRrluPlan& pl : plans)
        if (pl.kind == 2) {
            const size_t n = (rrlu_xcd_keys_bytes(pl.xcd) + rrlu_xcd_cols_bytes(pl.xcd, pl.kM)) / sizeof(unsigned long long);
            need = std::max(need, n);
            max_w = std::max(max_w, pl.xcd.K > 1 ? 33 : pl.xcd.W); // (a plan over several XCDs reserves the whole chip)
        }
    if (need > 0) {
        if (need > d_xkeys_.cap || !d_xticket_.get()) {
            d_xkeys_.reserve(std::max(need, reserve_mailbox_words));
            d_xticket_.reserve(16);
            T4A_HIP(hipMemsetAsync(d_xkeys_.get(), 0, d_xkeys_.cap * sizeof(unsigned long long), stream_));
            T4A_HIP(hipMemsetAsync(d_xticket_.get(), 0, 16 * sizeof(unsigned), stream_));
            xcd_ticket_base_ = xcd_ticket_base_multi_ = 0;
            xcd_salt_ = 0;
        }
        chain_lock_.acquire(max_w > kXcdSharedMaxW ? -1 : xcc_);
    }
}

void Engine::chain_end() { chain_lock_.release(); }

bool Engine::chain_group_plan(int kM, int kN, ChainRrluPlan* out)
{
    if (kM < 1 || kN < 1 || kM > 1024 || kN > 1024 || xcd_disabled()) return false;
    ChainRrluPlan pl;
    pl.kM = kM;
    pl.kN = kN;
    if (!rrlu_w1_make_plan(kM, kN, &pl.xcd, 0) && !rrlu_wg_make_plan(kM, kN, &pl.xcd, 0) && !rrlu_xcd_make_plan(kM, kN, &pl.xcd, true, 32)) return false;
    pl.kind = 2;
    pl.code = pl.xcd.wg == 2 ? 300000 + pl.xcd.RPT * 1000 + pl.xcd.CPT * 10 : pl.xcd.wg ? 200000 + pl.xcd.RPT * 1000 + pl.xcd.CPT * 10 : 100000 + pl.xcd.RPT * 100 + pl.xcd.CPT * 10;
    *out = pl;
    return true;
}

void Engine::chain_group_reserve(const std::vector<ChainRrluPlan>& plans, size_t reserve_mailbox_words, hipStream_t order_stream)
{
    size_t need = 0;
    for (const ChainRrluPlan& pl : plans)
        if (pl.kind == 2) need = std::max(need, (rrlu_xcd_keys_bytes(pl.xcd) + rrlu_xcd_cols_bytes(pl.xcd, pl.kM)) / sizeof(unsigned long long));
    if (need > 0 && (need > d_xkeys_.cap || !d_xticket_.get())) {
        d_xkeys_.reserve(std::max(need, reserve_mailbox_words));
        d_xticket_.reserve(16);
        T4A_HIP(hipMemsetAsync(d_xkeys_.get(), 0, d_xkeys_.cap * sizeof(unsigned long long), order_stream));
        T4A_HIP(hipMemsetAsync(d_xticket_.get(), 0, 16 * sizeof(unsigned), order_stream));
        xcd_ticket_base_ = xcd_ticket_base_multi_ = 0;
        xcd_salt_ = 0;
    }
}

unsigned Engine::chain_group_args(const ChainRrluPlan& pl, bool left, const double* d_a, const int* d_dims, size_t max_bond_dim, double rel_tol,
                                  double abs_tol, const ChainBlock& blk, int slot, RrluXcdArgs* out, hipStream_t order_stream)
{
    double* d_dres = reinterpret_cast<double*>(blk.dev);
    int* d_ires = reinterpret_cast<int*>(blk.dev + 16);
    double* d_pivvals = reinterpret_cast<double*>(blk.dev + blk.off_piv);
    int* d_rowperm = reinterpret_cast<int*>(blk.dev + blk.off_rp);
    int* d_colperm = reinterpret_cast<int*>(blk.dev + blk.off_cp);
    const int mn = pl.kM < pl.kN ? pl.kM : pl.kN;
    const int max_steps = max_bond_dim < (size_t)mn ? (int)max_bond_dim : mn;
    if (++xcd_salt_ > 65535u) { // the 16-bit launch salt wraps: stale granules could match again, clear the mailbox
        T4A_HIP(hipMemsetAsync(d_xkeys_.get(), 0, d_xkeys_.cap * sizeof(unsigned long long), order_stream));
        xcd_salt_ = 1;
    }
    RrluXcdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.A = d_a;
    a.Aout = nullptr;
    a.urows = nullptr;
    a.M = pl.kM;
    a.N = pl.kN;
    a.max_steps = max_steps;
    a.rel_tol = rel_tol;
    a.abs_tol = abs_tol;
    a.tie_row_major = left ? 0 : 1;
    a.out_transposed = left ? 0 : 1;
    a.W = pl.xcd.W;
    a.xcc = slot;
    xcd_take_tickets(pl.xcd, a); // (the one-workgroup kernel takes no tickets)
    a.row_perm = left ? d_rowperm : d_colperm;
    a.col_perm = left ? d_colperm : d_rowperm;
    a.iresult = d_ires;
    a.dresult = d_dres;
    a.pivot_vals = d_pivvals;
    a.keys = d_xkeys_.get();
    a.salt = xcd_salt_;
    static const double xspec = diag_env("T4A_XCD_SPECFRAC") ? std::atof(diag_env("T4A_XCD_SPECFRAC")) : 0.8;
    a.spec_frac = xspec;
    a.stamps = nullptr;
    a.h_block = reinterpret_cast<unsigned long long*>(blk.host);
    a.block_u64 = (int)(blk.bytes / 8);
    a.dims = d_dims;
    a.dims_swap = left ? 0 : 1;
    a.rowmap = nullptr;
    a.ts_u64 = (int)(blk.off_ts / 8);
    *out = a;
    return xcd_salt_;
}

unsigned Engine::chain_rrlu(const ChainRrluPlan& pl, bool left, const double* d_a, const int* d_rowmap, const FusedPi* fused, const int* d_dims,
                            size_t max_bond_dim, double rel_tol, double abs_tol, const ChainBlock& blk, const XcdSpecArgs* spec, double* d_aout, double* d_urows)
{
    double* d_dres = reinterpret_cast<double*>(blk.dev);
    int* d_ires = reinterpret_cast<int*>(blk.dev + 16);
    double* d_pivvals = reinterpret_cast<double*>(blk.dev + blk.off_piv);
    int* d_rowperm = reinterpret_cast<int*>(blk.dev + blk.off_rp);
    int* d_colperm = reinterpret_cast<int*>(blk.dev + blk.off_cp);
    const int mn = pl.kM < pl.kN ? pl.kM : pl.kN;
    const int max_steps = max_bond_dim < (size_t)mn ? (int)max_bond_dim : mn;
    unsigned token = 0u;
    if (pl.kind == 2) {
        RrluXcdArgs a;
        token = chain_group_args(pl, left, d_a, d_dims, max_bond_dim, rel_tol, abs_tol, blk, xcc_, &a, stream_);
        a.rowmap = d_rowmap;
        a.Aout = d_aout;
        a.urows = d_aout ? d_urows : nullptr;
        if (spec) a.spec = *spec;
        rrlu_xcd_launch_v(xcd_version(), pl.xcd, a, stream_);
    } else {
        if (++done_token_ == 0u) ++done_token_;
        RrluRegArgs a;
        std::memset(&a, 0, sizeof(a));
        a.A = d_a;
        a.Aout = d_aout;
        a.M = pl.kM;
        a.N = pl.kN;
        a.max_steps = max_steps;
        a.rel_tol = rel_tol;
        a.abs_tol = abs_tol;
        a.tie_row_major = left ? 0 : 1;
        a.out_transposed = left ? 0 : 1;
        a.W = 1;
        a.TR = pl.reg.TR;
        a.TC = pl.reg.TC;
        a.row_perm = left ? d_rowperm : d_colperm;
        a.col_perm = left ? d_colperm : d_rowperm;
        a.iresult = d_ires;
        a.dresult = d_dres;
        a.pivot_vals = d_pivvals;
        a.salt = 1;
        a.ncopy = 1;
        a.spec = 2;
        a.spec_frac = 0.8;
        a.key16 = 1;
        a.spin_limit = 1u << 20;
        a.fused = (pl.fused && fused) ? 1 : 0;
        if (a.fused) { // (the chain hands over the accumulators of the KERNEL's rows and columns)
            a.rowacc = fused->d_rowacc;
            a.colacc = fused->d_colacc;
            a.fn = fused->fn;
        }
        a.h_block = reinterpret_cast<unsigned long long*>(blk.host);
        a.block_u64 = (int)(blk.bytes / 8);
        a.dims = d_dims;
        a.dims_swap = left ? 0 : 1;
        a.rowmap = a.fused ? nullptr : d_rowmap;
        a.ts_u64 = (int)(blk.off_ts / 8);
        a.dev_token = done_token_;
        a.done_token = done_token_; // (also to int word 7 of the host mirror, like the single-XCD kernel's salt)
        rrlu_reg_launch(pl.reg, a, stream_, true);
        token = done_token_;
    }
    T4A_HIP(hipGetLastError());
    // this path leaves its own blocks behind: the next luci() call must not trust the "clean header" of its block blindly
    return token;
}

// factors_from_rrlu (matrix_luci.rs:256-279) on the factored matrix in d_lu_ (permuted coordinates).
void Engine::build_factors(const LuciResult& r, bool left_orth) { build_factors_from(d_lu_.get(), d_rowperm_ptr_, d_colperm_ptr_, r.M, r.N, r.rank, left_orth); }

void Engine::build_factors_from(const double* lu, const int* d_rowperm_ptr_, const int* d_colperm_ptr_, int M, int N, int rk, bool left_orth)
{
    d_left_.reserve((size_t)M * (rk > 0 ? rk : 1));
    d_right_.reserve((size_t)N * (rk > 0 ? rk : 1));
    if (rk == 0) return;
    static const bool no_small = diag_env("T4A_NO_SMALL_FACTORS") != nullptr;
    if (!no_small && luci_factors_small_launch(lu, M, N, rk, d_rowperm_ptr_, d_colperm_ptr_, left_orth, d_left_.get(), d_right_.get(), stream_)) {
        T4A_HIP(hipGetLastError());
        return;
    }
    d_w1_.reserve((size_t)(M > N ? M : N) * rk + (size_t)rk * rk);
    d_w2_.reserve((size_t)(M > N ? M : N) * rk + (size_t)rk * rk);
    d_trsm_.reserve(1);
    h_trsm_.reserve(1);

    if (left_orth) {
        // left = P_row^T [I_r ; L21 L11^{-1}]   (rrlu_cols_times_pivot_solve, matrix_luci.rs:206-229)
        double* Wl = d_w1_.get(); // M x rk, permuted order
        set_identity_launch(Wl, M, rk, M, stream_);
        if (rk < M) {
            double* Tt = d_w2_.get();                 // rk x rk : transpose of the leading block (upper part = L11^T)
            double* Bt = d_w2_.get() + (size_t)rk * rk; // rk x (M-rk) : L21^T
            transpose_launch(lu, rk, rk, M, Tt, rk, stream_);
            transpose_launch(lu + rk, M - rk, rk, M, Bt, rk, stream_);
            TrsmProblem tp;
            tp.T = Tt;
            tp.ldt = rk;
            tp.n = rk;
            tp.B = Bt;
            tp.ldb = rk;
            tp.nrhs = M - rk;
            tp.lower = 0;
            tp.unit_diag = 1; // L11 has a unit diagonal (dividing by 1.0 is exact)
            tp.skip_flag = nullptr;
            // (the pinned descriptor may still wait for the upload of the previous call: callers that build factors back to back —
            // the chained 1-site sweep — do not synchronise in between)
            T4A_HIP(hipStreamSynchronize(stream_));
            *h_trsm_.get() = tp;
            T4A_HIP(hipMemcpyAsync(d_trsm_.get(), h_trsm_.get(), sizeof(TrsmProblem), hipMemcpyHostToDevice, stream_));
            trsm_left_batched_launch(d_trsm_.get(), 1, rk, M - rk, stream_);
            transpose_launch(Bt, rk, M - rk, rk, Wl + rk, M, stream_);
        }
        scatter_rows_launch(Wl, M, d_rowperm_ptr_, M, rk, d_left_.get(), M, stream_);
        // right = (L11 U) P_col^T   (rrlu_rowmatrix, matrix_luci.rs:191-204)
        double* L11 = d_w2_.get();
        double* Ue = d_w1_.get(); // reuse after the scatter above has been enqueued (same stream: ordered)
        tri_extract_launch(lu, M, rk, rk, 1, 1, L11, rk, stream_);
        // U needs its own buffer: Wl is still being read by the scatter only until it completes (in-order stream)
        tri_extract_launch(lu, M, rk, N, 0, 0, Ue, rk, stream_);
        double* Rp = d_w2_.get() + (size_t)rk * rk; // rk x N
        GemmDesc g;
        g.m = rk;
        g.n = N;
        g.k = rk;
        g.A = L11;
        g.lda = rk;
        g.strideA = 0;
        g.transA = 0;
        g.B = Ue;
        g.ldb = rk;
        g.strideB = 0;
        g.transB = 0;
        g.C = Rp;
        g.ldc = rk;
        g.strideC = 0;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        gemm_launch(g, stream_);
        scatter_cols_launch(Rp, rk, rk, d_colperm_ptr_, N, d_right_.get(), rk, stream_);
    } else {
        // left = P_row^T (L U11)   (rrlu_colmatrix, matrix_luci.rs:176-189)
        double* Le = d_w1_.get();                    // M x rk lower trapezoid, diagonal kept
        double* U11 = d_w2_.get();                   // rk x rk unit upper
        double* Lp = d_w2_.get() + (size_t)rk * rk;  // M x rk product, permuted order
        tri_extract_launch(lu, M, M, rk, 1, 0, Le, M, stream_);
        tri_extract_launch(lu, M, rk, rk, 0, 1, U11, rk, stream_);
        GemmDesc g;
        g.m = M;
        g.n = rk;
        g.k = rk;
        g.A = Le;
        g.lda = M;
        g.strideA = 0;
        g.transA = 0;
        g.B = U11;
        g.ldb = rk;
        g.strideB = 0;
        g.transB = 0;
        g.C = Lp;
        g.ldc = M;
        g.strideC = 0;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        gemm_launch(g, stream_);
        scatter_rows_launch(Lp, M, d_rowperm_ptr_, M, rk, d_left_.get(), M, stream_);
        // right = [I_r , U11^{-1} U12] P_col^T   (rrlu_pivot_solve_times_rows, matrix_luci.rs:231-254)
        double* Wr = d_w1_.get(); // rk x N (ordered after the gemm that read Le on the same stream)
        set_identity_launch(Wr, rk, N, rk, stream_);
        if (rk < N) {
            gather_launch(lu + (size_t)M * rk, M, nullptr, rk, nullptr, N - rk, Wr + (size_t)rk * rk, rk, stream_);
            TrsmProblem tp;
            tp.T = lu;
            tp.ldt = M;
            tp.n = rk;
            tp.B = Wr + (size_t)rk * rk;
            tp.ldb = rk;
            tp.nrhs = N - rk;
            tp.lower = 0;
            tp.unit_diag = 1; // U11 carries a forced unit diagonal (matrixlu.rs:647-651)
            tp.skip_flag = nullptr;
            // (the pinned descriptor may still wait for the upload of the previous call: callers that build factors back to back —
            // the chained 1-site sweep — do not synchronise in between)
            T4A_HIP(hipStreamSynchronize(stream_));
            *h_trsm_.get() = tp;
            T4A_HIP(hipMemcpyAsync(d_trsm_.get(), h_trsm_.get(), sizeof(TrsmProblem), hipMemcpyHostToDevice, stream_));
            trsm_left_batched_launch(d_trsm_.get(), 1, rk, N - rk, stream_);
        }
        scatter_cols_launch(Wr, rk, rk, d_colperm_ptr_, N, d_right_.get(), rk, stream_);
    }
    T4A_HIP(hipGetLastError());
}

} // namespace t4a

// ------------------------------------------------------------------------------------------------
// lu.left(true) / lu.right(true), SVD, QR
// ------------------------------------------------------------------------------------------------
namespace t4a {

void Engine::lu_permuted_factors(const LuciResult& r, bool left_orth)
{
    const int M = r.M, N = r.N, rk = r.rank;
    d_left_.reserve((size_t)M * (rk > 0 ? rk : 1));
    d_right_.reserve((size_t)N * (rk > 0 ? rk : 1));
    if (rk == 0 || M == 0 || N == 0) return;
    const double* lu = d_lu_.get();
    d_w1_.reserve((size_t)(M > N ? M : N) * rk + (size_t)rk * rk);
    double* L = d_w1_.get();
    // L: lower trapezoid of the first rk columns (unit diagonal when left-orthogonal), rows back in original order
    tri_extract_launch(lu, M, M, rk, 1, left_orth ? 1 : 0, L, M, stream_);
    scatter_rows_launch(L, M, d_rowperm_ptr_, M, rk, d_left_.get(), M, stream_);
    d_w2_.reserve((size_t)(M > N ? M : N) * rk + (size_t)rk * rk);
    double* U = d_w2_.get();
    tri_extract_launch(lu, M, rk, N, 0, left_orth ? 0 : 1, U, rk, stream_);
    scatter_cols_launch(U, rk, rk, d_colperm_ptr_, N, d_right_.get(), rk, stream_);
    T4A_HIP(hipGetLastError());
}

// svd_backend: QR-preconditioned one-sided Jacobi (round 5).  For min(M, N) >= 64 the iteration does not run on A but on L = R^T of
// A' = Q R (A' = A or A^T, the taller orientation): L = U_L S V_L^T gives A' = (Q V_L) S U_L^T.  One-sided Jacobi on the columns of a
// LOWER triangular factor converges in far fewer sweeps when the spectrum is graded or the matrix is rank deficient — which is what the
// unfoldings of a tensor train under compression look like (Drmac / Veselic; measured with the cyclic ordering used here on 256 x 128:
// sigma_i = 2^-i 29 sweeps -> 12, a rank-40 matrix with 10 decades 28 -> 13, Gaussian 11 -> 11) — and every sweep works on n x n
// instead of m x n.  (Round 4 had tried the iteration on R itself: no gain — it is the TRANSPOSE that helps.)  Costs one Householder QR
// and one GEMM.  T4A_SVD_NO_PRECOND=1 (diagnostic builds) restores the plain iteration.
void Engine::svd(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt)
{
    if (M <= 0 || N <= 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "svd: empty matrix");
    const int kmin = M < N ? M : N, kmax = M < N ? N : M;
    static const bool no_groups = diag_env("T4A_SVD_NO_GROUPS") != nullptr;
    if (!no_groups && jacobi_fits_groups(kmax, kmin)) { // one launch: the kernel checks and scales its input itself (svd_plain)
        svd_plain(d_a, M, N, d_u, d_s, d_vt);
        return;
    }
    // One pass over the input: Inf / NaN (the reference's error, before a QR turns it into something else) and the largest magnitude.
    // The pair test of the Jacobi rotations forms alpha * beta (two squared column norms): from |a| ~ 1e77 on it overflowed, no pair
    // rotated and the factors came back non-orthogonal WITHOUT an error (tools/soak_svd_small.py, round 6; the Householder norms of the
    // QR in front overflow from 1e154 on).  As LAPACK does, a matrix whose largest entry is outside 2^-200 .. 2^200 is decomposed as
    // 2^e (A 2^-e): the scaling is exact, U and V are those of A, the singular values are scaled back.  Matrices in the usual range
    // take the same path as before, bit for bit.
    d_sflags_.reserve((size_t)kmin + 8);
    d_sabs_.reserve(1);
    int* flags = d_sflags_.get();
    T4A_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, stream_));
    T4A_HIP(hipMemsetAsync(d_sabs_.get(), 0, sizeof(unsigned long long), stream_));
    nonfinite_absmax_launch(d_a, (size_t)M * N, flags + 2, d_sabs_.get(), stream_);
    int h[4] = {0, 0, 0, 0};
    unsigned long long bits = 0;
    T4A_HIP(hipMemcpyAsync(h, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, stream_));
    T4A_HIP(hipMemcpyAsync(&bits, d_sabs_.get(), sizeof(bits), hipMemcpyDeviceToHost, stream_));
    T4A_HIP(hipStreamSynchronize(stream_));
    if (h[2]) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD computation failed: non-finite input");
    double amax;
    std::memcpy(&amax, &bits, sizeof(amax));
    const int e = pow2_scale_exponent(amax);
    if (e == 0) {
        svd_in_range(d_a, M, N, d_u, d_s, d_vt);
        return;
    }
    d_sscaled_.reserve((size_t)M * N);
    scale_pow2_launch(d_sscaled_.get(), d_a, (size_t)M * N, -e, stream_);
    svd_in_range(d_sscaled_.get(), M, N, d_u, d_s, d_vt);
    scale_pow2_launch(d_s, d_s, (size_t)kmin, e, stream_);
    T4A_HIP(hipGetLastError());
}

void Engine::svd_in_range(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt)
{
    static const bool no_precond = diag_env("T4A_SVD_NO_PRECOND") != nullptr;
    const int kmin = M < N ? M : N;
    if (no_precond || kmin < 64) {
        svd_plain(d_a, M, N, d_u, d_s, d_vt);
        return;
    }
    const bool flip = M < N;
    const int m = flip ? N : M, n = flip ? M : N;
    const double* Ap = d_a;
    if (flip) {
        d_pa_.reserve((size_t)m * n);
        transpose_launch(d_a, M, N, M, d_pa_.get(), N, stream_);
        Ap = d_pa_.get();
    }
    d_pq_.reserve((size_t)m * n);
    d_pr_.reserve((size_t)n * n);
    d_pl_.reserve((size_t)n * n);
    d_pul_.reserve((size_t)n * n);
    d_pvl_.reserve((size_t)n * n);
    qr(Ap, m, n, d_pq_.get(), d_pr_.get());                              // A' = Q R, Q m x n, R n x n
    transpose_launch(d_pr_.get(), n, n, n, d_pl_.get(), n, stream_);     // L = R^T
    svd_plain(d_pl_.get(), n, n, d_pul_.get(), d_s, d_pvl_.get());       // L = U_L S Vt_L
    // A' = Q L^T = (Q Vt_L^T) S U_L^T
    auto gemm_nt = [&](const double* A_, int lda, const double* B_, int ldb, double* C_, int ldc, int M_, int N_, int K_) {
        GemmDesc g{};
        g.m = M_;
        g.n = N_;
        g.k = K_;
        g.A = A_;
        g.lda = lda;
        g.transA = 0;
        g.B = B_;
        g.ldb = ldb;
        g.transB = 1;
        g.C = C_;
        g.ldc = ldc;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.batch = 1;
        gemm_launch(g, stream_);
    };
    if (!flip) {
        gemm_nt(d_pq_.get(), m, d_pvl_.get(), n, d_u, m, m, n, n);       // U = Q Vt_L^T (M x k)
        transpose_launch(d_pul_.get(), n, n, n, d_vt, n, stream_);       // Vt = U_L^T (k x N)
    } else {
        // A = A'^T = U_L S (Q Vt_L^T)^T: U = U_L (M x k, M = n), Vt = (Q Vt_L^T)^T (k x N, N = m)
        T4A_HIP(hipMemcpyAsync(d_u, d_pul_.get(), sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, stream_));
        d_pu_.reserve((size_t)m * n);
        gemm_nt(d_pq_.get(), m, d_pvl_.get(), n, d_pu_.get(), m, m, n, n);
        transpose_launch(d_pu_.get(), m, n, m, d_vt, n, stream_);
    }
    T4A_HIP(hipGetLastError());
}

void Engine::svd_plain(const double* d_a, int M, int N, double* d_u, double* d_s, double* d_vt)
{
    if (M <= 0 || N <= 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "svd: empty matrix");
    const bool flip = M < N;
    const int m = flip ? N : M, n = flip ? M : N;
    d_sw_.reserve((size_t)m * n);
    d_sv_.reserve((size_t)n * n);
    d_su_.reserve((size_t)m * n);
    d_svs_.reserve((size_t)n * n);
    d_ssig_.reserve((size_t)n + m);
    d_sflags_.reserve((size_t)n + 8);
    int* flags = d_sflags_.get(); // [0] rotated [1] n_dead [2] non-finite [3] converged / sweep count [4] scale exponent, [8..] dead[n]
    T4A_HIP(hipMemsetAsync(flags, 0, sizeof(int) * ((size_t)n + 8), stream_));
    // W and V of up to 96 columns fit one workgroup's LDS together: the whole iteration in ONE launch, sixteen lanes per column pair
    // (jacobi_groups_kernel; a 64 x 64 call 0.65 ms, of which the kernel 0.54, against 1.39 ms through QR + blocked tournament, 1.17 ms
    // for the blocked tournament alone and 2.23 ms for jacobi_small_kernel — profiles/r06_svd_small.txt; the kernel checks the input for
    // Inf / NaN itself and starts V from the identity).  T4A_SVD_NO_GROUPS=1 restores the round-5 routes.
    static const bool no_groups = diag_env("T4A_SVD_NO_GROUPS") != nullptr;
    const bool groups = !no_groups && jacobi_fits_groups(m, n);
    if (!groups) nonfinite_flag_launch(d_a, (size_t)M * N, flags + 2, stream_);
    double* W = d_sw_.get();
    if (flip)
        transpose_launch(d_a, M, N, M, W, N, stream_);
    else
        T4A_HIP(hipMemcpyAsync(W, d_a, sizeof(double) * (size_t)M * N, hipMemcpyDeviceToDevice, stream_));
    double* V = d_sv_.get();
    if (!groups) {
        fill_launch(V, (size_t)n * n, 0.0, stream_);
        set_identity_launch(V, n, n, n, stream_);
    }
    const int max_sweeps = 60;
    int h[4] = {0, 0, 0, 0};
    static const bool no_block = diag_env("T4A_SVD_NO_BLOCK") != nullptr;
    // tiny matrices (n <= 16: the cores of a small train, 2 x 2 rotations in the tests): the whole iteration in ONE launch of one
    // workgroup, no host round trip per sweep; everything else the blocked tournament (12.8 ms at 512 x 256 against 16.6, 2.3 ms at
    // 64 x 64 against 3.7: profiles/r04_linalg_probe.txt).  T4A_SVD_SMALL_N moves the boundary, T4A_SVD_NO_BLOCK=1 restores the
    // round-3 behaviour (one launch up to 128 columns, launch-per-round beyond).
    static const int small_n = diag_env("T4A_SVD_SMALL_N") ? std::atoi(diag_env("T4A_SVD_SMALL_N")) : 16;
    if (groups) {
        if (!jacobi_groups_launch(W, m, V, n, max_sweeps, flags + 2, stream_))
            throw Error(T4A_GPU_INTERNAL_ERROR, "svd: jacobi_fits_groups and jacobi_groups_launch disagree");
    } else if (jacobi_fits_small(m, n) && (no_block || n <= small_n)) {
        jacobi_small_launch(W, m, V, n, max_sweeps, stream_);
    } else {
        // sweeps in batches of two: the kernels of a sweep behind a converged one return at once (flags[3], jacobi_sweep_end_kernel), the
        // host reads the flags once per batch (a copy and a stream synchronisation per sweep were ~0.5 ms of a 512 x 256 decomposition)
        constexpr int batch = 2;
        for (int sweep = 0; sweep < max_sweeps; sweep += batch) {
            for (int k = 0; k < batch; ++k) {
                // blocked iteration (a tournament over column blocks, the pairs of a block pair inside one workgroup's LDS);
                // columns too long for the LDS keep the launch-per-round form
                if (no_block || !jacobi_block_sweep_launch(W, m, V, n, flags, stream_)) jacobi_sweep_launch(W, m, V, n, flags, stream_);
                jacobi_sweep_end_launch(flags, stream_);
            }
            T4A_HIP(hipMemcpyAsync(h, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, stream_));
            T4A_HIP(hipStreamSynchronize(stream_));
            if (h[2]) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD computation failed: non-finite input");
            if (h[3]) {
                static const bool dbg = std::getenv("T4A_SVD_DEBUG") != nullptr;
                if (dbg) std::fprintf(stderr, "[t4a svd] %d x %d: converged within %d sweeps\n", m, n, sweep + batch);
                break;
            }
        }
    }
    // the taller factor (m x n) and the square one (n x n): write straight to the outputs where no transpose is needed
    double* Ubig = flip ? d_su_.get() : d_u;   // m x n
    double* Vsq = flip ? d_u : d_svs_.get();   // n x n
    svd_finalize_launch(W, m, V, n, d_ssig_.get(), Ubig, d_s, Vsq, flags + 8, flags + 1, stream_);
    int h5[5] = {0, 0, 0, 0, 0};
    T4A_HIP(hipMemcpyAsync(h5, flags, sizeof(int) * 5, hipMemcpyDeviceToHost, stream_));
    T4A_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < 4; ++i) h[i] = h5[i];
    if (h[2]) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD computation failed: non-finite input");
    // (the one-launch kernel iterated on 2^-e A and left W that way: U = W / sigma does not see the factor, the singular values do)
    if (groups && h5[4] != 0) scale_pow2_launch(d_s, d_s, (size_t)n, h5[4], stream_);
    if (groups) {
        static const bool dbg = std::getenv("T4A_SVD_DEBUG") != nullptr;
        if (dbg) std::fprintf(stderr, "[t4a svd] %d x %d: one launch, converged within %d sweeps\n", m, n, h[3]);
    }
    if (h[1] > 0) svd_complete_launch(Ubig, m, n, flags + 8, d_ssig_.get() + n, stream_);
    if (flip)
        transpose_launch(Ubig, m, n, m, d_vt, n, stream_); // Vt (M x N) = U'^T, U' is N x M
    else
        transpose_launch(Vsq, n, n, n, d_vt, n, stream_);  // Vt (N x N) = V^T
    T4A_HIP(hipGetLastError());
}

void Engine::qr(const double* d_a, int M, int N, double* d_q, double* d_r)
{
    const int k = M < N ? M : N;
    if (k <= 0) return;
    d_sw_.reserve((size_t)M * N);
    d_ssig_.reserve((size_t)3 * k);
    double* W = d_sw_.get();
    // The copy the factorisation works on is 2^-e A for a matrix whose largest entry is outside 2^-200 .. 2^200 (the Householder norms are
    // sums of squares: entries of 1e-200 made every column "zero" and R came back 0 without an error — tools/soak_dense_small.py, round
    // 6); e is found and applied on the device (this routine has no host synchronisation), Q is that of A and R gets the factor back.
    d_sflags_.reserve(8);
    d_sabs_.reserve(1);
    T4A_HIP(hipMemsetAsync(d_sabs_.get(), 0, sizeof(unsigned long long), stream_));
    nonfinite_absmax_launch(d_a, (size_t)M * N, d_sflags_.get() + 5, d_sabs_.get(), stream_); // (the Inf / NaN flag is not used here)
    scale_pow2_dev_launch(W, d_a, (size_t)M * N, d_sabs_.get(), -1, stream_);
    double* diag = d_ssig_.get();
    double* tau = diag + k;
    double* v0s = tau + k;
    // blocked Householder: explicit reflectors (M x k), triangular factors per panel, two QR_PANEL x max(N, k) scratch blocks
    const size_t wcols = (size_t)(N > k ? N : k);
    d_sv_.reserve((size_t)M * k);
    d_svs_.reserve((size_t)qr_panels(k) * QR_PANEL * QR_PANEL + 2 * (size_t)QR_PANEL * wcols);
    double* Vall = d_sv_.get();
    double* Tall = d_svs_.get();
    double* Wa = Tall + (size_t)qr_panels(k) * QR_PANEL * QR_PANEL;
    double* Wb = Wa + (size_t)QR_PANEL * wcols;
    qr_factor_launch(W, M, N, diag, tau, v0s, Vall, Tall, Wa, Wb, stream_);
    qr_form_launch(W, M, N, diag, Vall, Tall, Wa, Wb, d_q, d_r, stream_);
    scale_pow2_dev_launch(d_r, d_r, (size_t)k * N, d_sabs_.get(), +1, stream_);
    T4A_HIP(hipGetLastError());
}

} // namespace t4a
