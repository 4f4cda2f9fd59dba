// kernels.hpp — launch interfaces of the hand-written gfx950 kernels.
// All pointers are device pointers; all matrices column-major.  Launchers enqueue on `stream` and do
// not synchronise.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/t4a_testfunctions.h"
#include "diag.hpp"

namespace t4a {

// ------------------------------------------------------------------------------------------------
// K2: full-pivot rank-revealing LU (replaces rrlu_mut, tensor4all-core/src/matrixlu.rs:735-819)
// ------------------------------------------------------------------------------------------------
struct RrluWorkspace;

struct RrluPlan {
    int W = 1;       // cooperating workgroups (1 => single-workgroup kernel, no mailbox)
    int T = 256;     // threads per workgroup
    int cpw = 0;     // columns per workgroup
    int Mld = 0;     // LDS leading dimension of a slab column
    size_t lds_bytes = 0;
};

// Chooses (W, T) for an M x N problem.  Honours T4A_RRLU_W / T4A_RRLU_T environment overrides.
RrluPlan rrlu_make_plan(int M, int N, int num_cus);

struct RrluArgs {
    const double* A;            // M x N input (ld = M)
    double* Aout;               // M x N factored matrix in permuted coordinates, or nullptr
    int M, N;
    int max_steps;              // min(max_bond_dim, M, N)
    double rel_tol, abs_tol;
    int left_orth;
    int W, cpw, Mld;
    int* row_perm;              // [M]  posrow: original row sitting at permuted position p
    int* col_perm;              // [N]
    int* iresult;               // [0] npivots  [1] timeout flag  [2] NaN-in-LU flag
    double* dresult;            // [0] last error (RrLU::error)  [1] max |a_ij| sample seen in A (sqrt(v*v))
    double* pivot_vals;         // [max_steps] value of the k-th pivot (diag of U resp. L)
    unsigned long long* keys;   // [2][W][4] tagged key granules (zeroed before every launch)
    unsigned long long* cols;   // [2][W][M] candidate pivot columns (f64 bits)
    unsigned spin_limit;
    unsigned long long* stamps; // diagnostic only (nullptr in production): 8 phase cycle counters
};

// Bytes of mailbox storage needed for a plan.
size_t rrlu_keys_bytes(const RrluPlan& plan);
size_t rrlu_cols_bytes(const RrluPlan& plan, int M);

// Enqueue memset of the key mailbox + the kernel.
void rrlu_launch(const RrluPlan& plan, const RrluArgs& args, hipStream_t stream);

// ---- HBM-resident fallback (kernels_rrlu_global.hip): any shape, three launches per pivot step ----
struct RrluGlobalArgs {
    const double* A;            // M x N input (ld = M), left untouched
    double* Aout;               // factored matrix in permuted coordinates (ld = M) or nullptr
    int M, N;
    int max_steps;
    double rel_tol, abs_tol;
    int left_orth;
    int* row_perm;
    int* col_perm;
    int* iresult;               // [0] npivots [2] NaN flag
    double* dresult;            // [0] last error [1] bits of max sqrt(v*v)
    double* pivot_vals;
    // workspace views, filled by rrlu_global_launch
    double* W;                  // working copy
    int *rowpos, *colpos, *posrow, *poscol;
    double *partials_sc, *partials_val;
    unsigned long long* partials_pos;
    int* partials_ij;
    int* istate;                // [0] ticket counter [1] stop flag [2] pivot row [3] pivot column (physical) [4] npivots
    double* dstate;             // [0] pivot [1] max_error [2] lu.error
};
int rrlu_global_blocks(int M, int N);
size_t rrlu_global_int_words(int M, int N, int blocks);
size_t rrlu_global_double_words(int M, int N, int blocks);
void rrlu_global_launch(RrluGlobalArgs args, int* iwork, double* dwork, hipStream_t stream);

// built-in device functor (include/t4a_testfunctions.h): id, number of integer accumulators, parameters
struct FnDevice {
    int fid;
    int n_acc;
    double params[T4A_FN_MAX_PARAMS];
};

// ---- register-resident fast path (kernels_rrlu_reg.hip): left-orthogonal elimination only ----
constexpr int RRLU_MAX_COPIES = 8;
constexpr int RRLU_FUSED_MAX_VALUES = 16; // fused Π build only for plans with RPT * CPT <= this
struct RrluRegPlan {
    int W = 1, T = 256;
    int TR = 256, TC = 1;   // thread grid inside a workgroup: rows x column groups
    int RPT = 1, CPT = 1;   // rows / columns per thread (template parameters)
    size_t lds_bytes = 0;
};
struct RrluRegArgs {
    const double* A;            // M x N input (ld = M)
    double* Aout;               // factored matrix in permuted coordinates (or nullptr)
    int M, N;
    int max_steps;
    double rel_tol, abs_tol;
    int tie_row_major;          // 1: ties go to the smallest (rowpos, colpos) — used for transposed problems
    int out_transposed;         // 1: Aout[colpos + N*rowpos] (i.e. the transpose, ld = N)
    int W, TR, TC;
    int* row_perm;
    int* col_perm;
    int* iresult;               // [0] npivots [1] timeout [2] NaN flag
    double* dresult;            // [0] last error [1] bits of max sqrt(v*v)
    double* pivot_vals;
    unsigned long long* keys;   // [2][W][2] shared key table of THIS launch: two 16-bit-tagged granules per workgroup and step parity
    unsigned long long* cols;   // [2][ncopy][M][2] tagged pivot-column granules; tag = salt*65536 + (step % 65535 + 1)
    unsigned salt;              // launch-unique 16-bit value (1..65535); the buffers are zeroed when it wraps
    int col_delay;              // >0: readers sleep briefly before their first pivot-column sweep
    int poll_delay;             // >0: the polling wave sleeps ~1000 cycles before its first key sweep
    int ncopy;                  // replicas of the published pivot column (<= RRLU_MAX_COPIES)
    int key16;                  // 1: the poller reads every key with one 16-byte load
    int spec;                   // 1: every workgroup publishes its candidate column with its key (cols is [2][W][M][2]);
                                // 2: only those whose candidate score is >= spec_frac * (previous pivot)^2
    double spec_frac;
    unsigned spin_limit;
    unsigned long long* stamps; // diagnostic only
    // optional host-visible (pinned) mirror of the packed result block that starts at `dresult`
    // ([dresult 2 f64][iresult 4 i32][pivot_vals][row_perm][col_perm]): workgroup 0 copies the block there at the end
    // of the kernel and the flag setters write their flags directly, so no device-to-host copy is needed afterwards
    unsigned long long* h_block;
    int block_u64;
    // key table of the NEXT multi-workgroup launch (the two tables alternate): cleared by workgroup 0 at the very end
    // of this launch together with the accumulated max|a| word, so that no memset is needed between launches
    unsigned long long* keys_next;
    int keys_next_u64;
    // arrival trace (only read by builds with -DT4A_RRLU_TRACE, nullptr otherwise): [W] XCC ids, then per step and
    // workgroup the real-time clock at key publication and at the end of the key gather
    unsigned long long* trace;
    // fused candidate-matrix build: when `fused` != 0 the kernel never reads A; entry (i, j) of ITS row / column
    // numbering is fn(rowacc[i] + colacc[j]) (accumulators [count][fn.n_acc] uint64, see kernels_pi.hip)
    int fused;
    const uint64_t* rowacc;
    const uint64_t* colacc;
    FnDevice fn;
    // single-workgroup launches only: when non-zero, written to int word 7 of h_block after everything else the workgroup sends
    // to the host (system-scope fence + barrier in front of it), so that the host may spin on it instead of waiting for the
    // stream — a small bond is a 10 us kernel behind a 12 us completion
    unsigned done_token;
    // bond chain (kernels_chain.hip; single-workgroup launches only): when `dims` != nullptr the matrix is dims[0] x dims[1]
    // (dims[1] x dims[0] with dims_swap) instead of M x N — M, N are then the upper bounds the launch was planned for —,
    // max_steps caps min(M, N), and `dev_token` is written to iresult[3] when the workgroup has run to its end
    const int* dims;
    int dims_swap;
    unsigned dev_token;
    const int* rowmap;          // non-null: entry (i, j) of the kernel's matrix is A[rowmap[i] + dims[3] * j] (speculative candidate matrix)
    int ts_u64;                 // > 0: start / end time of the workgroup (wall_clock64) at u64 words ts_u64, ts_u64 + 1 of the result block (single-workgroup launches)
};
// false if the shape is outside the fast path (fall back to the LDS kernel)
bool rrlu_reg_make_plan(int M, int N, int num_cus, RrluRegPlan* out);
size_t rrlu_reg_keys_bytes(const RrluRegPlan& plan);
size_t rrlu_reg_cols_bytes(const RrluRegPlan& plan, int M);
// keys_zeroed: the key table is already clear (the previous launch left it clean, see RrluRegArgs::keys_next)
void rrlu_reg_launch(const RrluRegPlan& plan, const RrluRegArgs& args, hipStream_t stream, bool keys_zeroed = false);

// ---- single-XCD register-resident kernel (kernels_rrlu_xcd.hip): all participating workgroups share one L2 ----
// Bond chain: the candidate matrix of the NEXT bond, evaluated speculatively by the launch's pass-through workgroups (the 7/8
// of the grid that land on the other XCDs and used to return at once) while the elected XCD factorises this bond:
// out[j * lda + c] = f(acc(c) + ind_acc[j]), candidate c < M * d: child (parent c / d of this bond's dependent list = the
// kernel's rows, digit c % d), beyond: extra c - M * d; lda = M * d + ne (kernels_chain.hip has the same layout).
struct XcdSpecArgs {
    double* out;               // nullptr: nothing to speculate on
    const uint64_t* dep_acc;   // [M][K] accumulators of the kernel's rows (this bond's dependent list)
    const uint64_t* w_site;    // weights of the site that extends the dependent side: w + woff[site], row stride `total`
    const uint64_t* ext_acc;   // [ne][K] accumulators of the next bond's extras
    const int* ext_cnt;        // -> ne (nullptr: none)
    const uint64_t* ind_acc;   // [ni][K] independent list of the next bond
    const int* ind_cnt;        // -> ni
    unsigned* tile_counter;    // work distribution among the pass-through workgroups (zero before the launch)
    int total, d;
    FnDevice fn;
};
struct RrluXcdPlan {
    int W = 1;              // participating workgroups (<= 32, one per compute unit of the elected XCD); agents = 8 W waves
    int RPT = 1, CPT = 1;   // rows per lane / columns per wave (template parameters)
    int grid = 8;           // launched workgroups = 8 W (blocks b and b + 8 share an XCD)
    size_t lds_bytes = 0;
    int wg = 0;             // 1: the one-workgroup kernel (kernels_rrlu_wg.hip): RPT rows per lane, CPT columns per WAVE, grid = 1 + speculating workgroups; 2: the one-wave kernel (kernels_rrlu_w1.hip)
    int K = 1;              // XCDs the agents live on (round 5, second generation only): agents = 8 K W; K > 1 or RPT > 16: kernels_rrlu_xcd2m.hip
    bool big() const { return K > 1 || RPT > 16; }
};
struct RrluXcdArgs {
    const double* A;            // M x N input (ld = M)
    double* Aout;               // factored matrix in permuted coordinates (or nullptr)
    double* urows;              // [max_steps][N] finished rows of U by original column index (needed when Aout != nullptr)
    int M, N;
    int max_steps;
    double rel_tol, abs_tol;
    int tie_row_major;          // 1: ties go to the smallest (rowpos, colpos) — transposed problems
    int out_transposed;         // 1: Aout[colpos + N*rowpos]
    int W;                      // participating workgroups
    int xcc;                    // XCC id (HW_REG_XCC_ID) of the elected XCD (K > 1: the first of K neighbours, mod 8)
    unsigned* ticket;           // monotonic ticket counter of this engine; ranks are ticket - ticket_base (K > 1: eight counters, one per XCD)
    unsigned ticket_base;
    int* row_perm;
    int* col_perm;
    int* iresult;               // [0] npivots [1] timeout [2] NaN flag
    double* dresult;            // [0] last error [1] bits of max sqrt(v*v)
    double* pivot_vals;
    unsigned long long* keys;   // the mailbox: [2][8 W] 16-byte keys {value lo, value hi, meta, tag ^ fold} followed by
                                // [2][8 W][64 RPT] 16-byte column rows {lo, hi, 0, tag ^ fold} (rrlu_xcd_keys_bytes + rrlu_xcd_cols_bytes)
    unsigned salt;              // launch-unique 16-bit value (1..65535); tag = salt << 16 | (step + 1)
    double spec_frac;
    unsigned long long* stamps; // diagnostic only
    unsigned long long* h_block; // pinned mirror of the packed result block (see RrluRegArgs)
    int block_u64;
    // bond chain (kernels_chain.hip): when `dims` != nullptr the matrix is dims[0] x dims[1] (dims[1] x dims[0] with
    // dims_swap) instead of M x N — M, N are then the upper bounds the launch was planned for — and max_steps caps min(M, N).
    // Rank 0 writes `salt` to iresult[3] when it has run to its end (the next kernel of the chain checks it).
    const int* dims;
    int dims_swap;
    const int* rowmap;          // non-null: entry (i, j) of the kernel's matrix is A[rowmap[i] + dims[3] * j] (speculative candidate matrix)
    XcdSpecArgs spec;
    int ts_u64;                 // > 0: rank 0 stores its start / end time (wall_clock64, 100 MHz) at u64 words ts_u64, ts_u64 + 1 of the result block and its mirror
};
// any_size: take every shape the plan family can hold (the bond chain has no other multi-workgroup kernel); otherwise tiny
// matrices are left to the single-workgroup plan of the chip-wide kernel
// max_w: most workgroups the plan may use (<= 32 = the compute units of an XCD; fewer when other handles share the chip, see
// XcdArbiter in engine.hip)
// allow_big: plans beyond one XCD's 1024 x 1024 (up to 1536 rows; columns over K <= 4 XCDs) — second-generation kernel only, so not
// on the retry after non-finite values
bool rrlu_xcd_make_plan(int M, int N, RrluXcdPlan* out, bool any_size = false, int max_w = 32, bool allow_big = false);
size_t rrlu_xcd_keys_bytes(const RrluXcdPlan& plan);
size_t rrlu_xcd_cols_bytes(const RrluXcdPlan& plan, int M);
// Eight factorisations in one launch, one per XCD (slot x is run by the workgroups that land on XCD x; a slot with xcc = -1 is
// empty).  All slots share the plan (made for the largest upper-bound shape among them) and the tie order; every slot brings
// its own mailbox, ticket counter and result block.  kernels_rrlu_xcd_group.hip.
struct RrluXcdGroupArgs {
    RrluXcdArgs p[8];
};
// Second generation of the same kernel (kernels_rrlu_xcd2.hip: one-word record, stop tests and tables off the critical path, no
// hand-zeroed pivot rows): same plan, arguments and mailbox.  It handles finite matrices only: on a NaN / infinity in the input or
// an overflow in the trailing block the launch gives up with iresult[1] == 2 and the caller runs the first generation.
void rrlu_xcd2_launch(const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream);
// plans with big() (kernels_rrlu_xcd2m.hip): the same kernel body with agents on K XCDs and / or 20 - 24 row slots per lane
void rrlu_xcd2m_launch(const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream);
void rrlu_xcd2_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& args, bool tie_row_major, hipStream_t stream);
// One-workgroup kernel (kernels_rrlu_wg.hip): matrices up to 64 x 512 / 128 x 384 in the registers of one compute
// unit, exchange through its LDS (one barrier per pivot step).  Same arguments and result block; finite matrices only (gives up
// with iresult[1] == 2 like the second-generation single-XCD kernel).  spec_blocks: workgroups beside the factorising one that
// evaluate the next bond's candidate matrix (bond chain).  The group launch runs slot x in workgroup x.
bool rrlu_wg_make_plan(int M, int N, RrluXcdPlan* out, int spec_blocks = 0);
void rrlu_wg_launch(const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream);
void rrlu_wg_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& args, bool tie_row_major, hipStream_t stream);
// One-wave kernel (kernels_rrlu_w1.hip): matrices up to 64 x 64 in the registers of ONE wavefront (lane = row), no barrier and
// no exchange at all.  Plans carry wg = 2, RPT = 1, CPT = register columns.  Same arguments, result block and give-up rule.
bool rrlu_w1_make_plan(int M, int N, RrluXcdPlan* out, int spec_blocks = 0);
void rrlu_w1_launch(const RrluXcdPlan& plan, const RrluXcdArgs& args, hipStream_t stream);
void rrlu_w1_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& args, bool tie_row_major, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// K1: candidate-matrix build (replaces the Π loop, tensor4all-tensorci/src/tensorci2.rs:1859-1893)
// out[i + ld*j] = g(rowacc[i] + colacc[j]); *max_abs_bits = max over entries of bits(sqrt(v*v)).
// rowacc/colacc are [count][n_acc] uint64.  If `transpose_out`, writes out[j + ld*i].
// ------------------------------------------------------------------------------------------------
void pi_eval_launch(const FnDevice& fn, const uint64_t* rowacc, int M, const uint64_t* colacc, int N, double* out,
                    int ld, bool transpose_out, unsigned long long* max_abs_bits, hipStream_t stream);
// Several independent matrices in one launch (fill_site_tensors: two per site): blockIdx.z = job.
struct PiJob {
    const uint64_t* rowacc;
    const uint64_t* colacc;
    double* out;
    unsigned long long* max_abs_bits; // may be null
    int M, N, ld, pad_;
};
void pi_eval_batched_launch(const FnDevice& fn, const PiJob* d_jobs, int n_jobs, int max_M, int max_N, hipStream_t stream);
// all cores of one fill_site_tensors in a single launch (one job per site)
struct PackJob {
    const double* src;
    double* core;
    const int* info; // null for the last site
    int ld, L, S, R, last, pad_;
};
struct LuProblem;
// fill_site_tensors of a SMALL problem (every pivot matrix at most 32 x 32, at most 64 right-hand sides: BASELINE configs[1], the first
// iterations of every run) as ONE launch, one workgroup per site: both evaluations into the LDS, the zero-pivot-matrix guard, the
// partial-pivot LU and the two substitutions (the arithmetic of lu_kernel + trsm_left_kernel, operation for operation) and the packing
// of the core — where the general path issues evaluation, panel, update, triangular solve and packing as five dependent launches.
// jobs: n_jobs sites in order, site k owns pis[2 k] (Pi1^T), pis[2 k + 1] (P^T), lups[k], packs[k]; when last_site != 0 the final job is
// the last site of the train (one evaluation, no solve).
constexpr int FILL_SMALL_MAX_N = 32, FILL_SMALL_MAX_RHS = 64;
void fill_small_launch(const FnDevice& fn, const PiJob* d_pis, const LuProblem* d_lups, const PackJob* d_packs, int n_jobs, int last_site,
                       hipStream_t stream);
// count u64 from a pinned (device-visible) host buffer to device memory, as a kernel on `stream`
void stage_copy_launch(const uint64_t* pinned_src, uint64_t* dst, size_t count, hipStream_t stream);
// max over a dense buffer of bits(sqrt(v*v)) (host-callback path)
void absmax_launch(const double* data, size_t count, unsigned long long* max_abs_bits, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// Bond chain (kernels_chain.hip): device-resident index-set tables and the per-bond preparation kernel of the host-free
// half-sweep (replaces the host part of update_pivots, tensorci2.rs:1833-1846, :1934-1949, between two bonds)
// ------------------------------------------------------------------------------------------------
constexpr int CHAIN_MAX_SET = 1024; // entries per site a table (and a history table) may hold
struct ChainTab {
    uint64_t* code; // [n_sites][cap]      mixed-radix code of the multi-index (kernels_chain.hip)
    uint64_t* acc;  // [n_sites][cap][K]   integer accumulators of the built-in functor
    int* cnt;                 // [n_sites]
};
// what every kernel of one half-sweep shares
struct ChainCommon {
    ChainTab I, J, HI, HJ;    // current sets I_p / J_p, history snapshot (the extras of this iteration, tensorci2.rs:1675-1685)
    ChainTab mI, mJ;          // pinned host mirrors of I, J: written by the gather, they ARE the host's copy of the sets
    int cap, K;
    const uint64_t* w;        // [K][total] weights of the functor
    int total;
    const int* ldim;          // [n_sites] local dimensions
    const int* woff;          // [n_sites] offset of a site in a weight row
    int forward, use_extras;
    int one_site;             // 1: a 1-site sweep (tensorci2.rs:918-1050): the independent side of bond b is the table itself — J_b
                              // sweeping forward, I_{b+1} sweeping backward — without Kronecker product and without extras
    uint64_t* ind_code;       // [n_bonds][ind_cap]     independent side of every bond (columns forward, rows backward)
    uint64_t* ind_acc;        // [n_bonds][ind_cap][K]
    int* ind_cnt;             // [n_bonds]
    int ind_cap;
    uint64_t* dep_code;       // [dep_cap]              dependent side of the CURRENT bond (read by the gather, then overwritten)
    uint64_t* dep_acc;        // [dep_cap][K]
    int dep_cap;
    int* rowmap;              // [dep_cap] list position -> candidate of the speculative matrix
    int* dims;                // [n_bonds][4]: M, N, poison, leading dimension of the matrix the rrLU kernel loads
    int* hdims;               // pinned mirror of dims (M, N, poison)
    // housekeeping of a new chain, done by chain_indep_kernel instead of five memset / memcpy operations on the stream:
    uint64_t* snap_dst;       // != nullptr: tables I, J (two consecutive families: codes, accumulators) are copied here ...
    int* snap_cnt_dst;        // ... and their counts here: the snapshot that becomes the extras of the next iteration
    size_t snap_words;        // u64 words of the two families
    int n_sites;
    uint64_t* zero_a;         // result blocks of all bonds ...
    size_t zero_a_words;
    uint64_t* zero_b;         // ... and dims + tile counters: cleared
    size_t zero_b_words;
};
struct ChainPrepArgs {
    int b;                    // bond to build (ignored when do_build == 0)
    int do_build;             // 0: gather only (after the last bond)
    int with_rowmap;          // the rrLU of this bond loads the speculative matrix through the row map
    int prev_b;               // previous bond of the half-sweep (< 0: none): its pivots are gathered into I_{prev_b+1}, J_{prev_b}
    const int* prev_iresult;  // [0] npiv [1] timeout [3] completion token
    const int* prev_rowperm;
    const int* prev_colperm;
    unsigned prev_token;
    unsigned long long* dbg;  // diagnostic (T4A_PREP_DEBUG): [4] gather [5] dependent list [6] whole body, 100 MHz ticks summed over the calls
    int defer_host_writes;    // 1 (persistent half-sweep): the pinned mirrors and hdims are not written here but in bulk at the end of
                              // the kernel — a store over PCIe is acknowledged after microseconds, and every barrier waits for it
};
// The persistent half-sweep (kernels_chain.hip, chain_walk_kernel): one workgroup walks all bonds — preparation, candidate matrix,
// one-wave rrLU — when every bond's matrix is at most 64 x 64.  Result blocks as the launched chain writes them.
struct ChainWalkArgs {
    char* blocks;                       // [n_bonds] packed result blocks (ChainBlock layout)
    size_t block_bytes, off_piv, off_rp, off_cp, off_ts;
    double* pi;                         // candidate matrix of the current bond (64 x 64 doubles)
    int n_bonds;
    int max_steps;                      // max_bond_dim (clipped to 64)
    double rel_tol, abs_tol;
    double* factors;                    // != nullptr: the factored matrix of bond b (permuted coordinates, original orientation) at factors + b * factors_stride
    size_t factors_stride;
    unsigned token_base;                // bond number k of the half-sweep completes with token token_base + k
    int timed;                          // device time stamps of every factorisation at off_ts
    int lean_prep;                      // 1: the preparation of a bond by one wave out of the LDS (walk_prep_wave0); 0: chain_prep_body (T4A_WALK_OLD_PREP=1)
    unsigned long long* phase_ticks;    // diagnostic (T4A_WALK_DEBUG): [8] 100 MHz ticks summed over the bonds: preparation, candidate matrix, rrLU, total; [4] gather [5] dependent list (inside the preparation)
};
void chain_walk_launch(const ChainCommon& c, const FnDevice& fn, const ChainWalkArgs& w, int columns, hipStream_t stream);
void chain_indep_launch(const ChainCommon& c, int n_bonds, hipStream_t stream);
// behind a chain whose preparations ran with defer_host_writes: tables -> pinned mirrors, dims -> hdims, in one launch
void chain_mirror_launch(const ChainCommon& c, int n_bonds, hipStream_t stream);
// forward 1-site sweep with update_tensors: the LAST site's tensor Pi1 = f(kron(I_{n-1}, d_{n-1}), J_{n-1}) (tensorci2.rs:902-912, fill_tensor
// :813-850), evaluated from the device tables right behind the chain — core[l + L (s + S r)], L = |I_{n-1}| as the chain left it
void chain_last_core_launch(const ChainCommon& c, const FnDevice& fn, double* core, int max_entries, hipStream_t stream);
void chain_prep_launch(const ChainCommon& c, const ChainPrepArgs& a, hipStream_t stream);
// n_dep_ub / n_ind_ub: upper bounds for the launch grid (the kernel reads the real sizes on the device)
void chain_pi_launch(const ChainCommon& c, const FnDevice& fn, int b, int n_dep_ub, int n_ind_ub, double* out, hipStream_t stream);

// Group chain: up to CHAIN_GROUP_MAX handles (independent interpolations: patches of one farm) advance through their
// half-sweeps in lock step — ONE launch per kernel and bond for all of them.  The per-handle constants of a half-sweep sit in
// a device table (one slot per handle), the per-bond arguments travel as small arrays in the kernel arguments; the rrLU launch
// gives every handle its own XCD (rrlu_xcd_group_launch).  Bond indices and the sweep direction are common to the group.
constexpr int CHAIN_GROUP_MAX = 8;
struct ChainGroupSlot {
    ChainCommon c;
    FnDevice fn;
    double* pi;               // candidate matrix buffer of the handle (chain_pi_group_launch writes it)
};
struct ChainPrepGroupArgs {
    ChainPrepArgs a[CHAIN_GROUP_MAX];
};
void chain_indep_group_launch(const ChainGroupSlot* d_slots, int n_handles, int n_bonds, hipStream_t stream);
void chain_prep_group_launch(const ChainGroupSlot* d_slots, const ChainPrepGroupArgs& a, int n_handles, hipStream_t stream);
void chain_pi_group_launch(const ChainGroupSlot* d_slots, int n_handles, int b, int n_dep_ub, int n_ind_ub, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// Dense helpers
// ------------------------------------------------------------------------------------------------
// C[m x n] (ldc) = alpha * op(A)[m x k] * op(B)[k x n] + beta * C, batched with element strides.
// f64 MFMA 16x16x4 tiles staged through LDS.
struct GemmDesc {
    int m, n, k;
    const double* A; int lda; long long strideA; int transA;
    const double* B; int ldb; long long strideB; int transB;
    double* C; int ldc; long long strideC;
    double alpha, beta;
    int batch;
    // split-K (set by gemm_launch, not by callers): slice s of a problem multiplies k-tiles [s * ktiles_per, ...) and stores its raw
    // accumulators to partial[(problem * ksplit + s) * m * n + col * m + row]; gemm_splitk_reduce_kernel forms C from them
    int ksplit = 1;
    double* partial = nullptr;
};
void gemm_launch(const GemmDesc& d, hipStream_t stream);

// LUCI factors of a small factorisation (rank <= 16, M, N <= 1024) in one launch: lu = factored M x N matrix in permuted coordinates (ld M),
// left: M x rk (ld M), right: rk x N (ld rk), both in original row / column order.  Returns false when the shape is not taken.
// The site tensors of a chained forward 1-site sweep (tensorci2.rs:1020-1038) for every bond whose rank is at most 16, in ONE launch
// behind the chain: job b reads its shape, rank and row permutation where the chain left them on the device, forms
// left = P_row^T [I_r ; L21 L11^{-1}] exactly like luci_factors_small_kernel and writes it as core[l, s, r] (column-major, R = max(rank, 1)).
// Bonds of higher rank, poisoned or unfinished bonds are left alone (the host builds those one by one).
constexpr int LUCI_LEFT_CORES_MAX_JOBS = 80;
constexpr int LUCI_LEFT_CORES_MAX_RANK = 16;
struct LeftCoreJob {
    const double* lu;       // factored matrix (permuted coordinates, ld = rows)
    const int* dims;        // {rows, columns, poisoned, -}
    const int* iresult;     // {rank, gave-up code, NaN flag, completion token}
    const int* row_perm;
    double* core;           // [L = rows / S][S][R]
    int S;
    unsigned token;         // the completion token the bond must carry
};
struct LeftCoreJobs {
    LeftCoreJob j[LUCI_LEFT_CORES_MAX_JOBS];
};
void luci_left_cores_batched_launch(const LeftCoreJobs& jobs, int n_jobs, int max_rows, hipStream_t stream);
bool luci_factors_small_launch(const double* lu, int M, int N, int rk, const int* row_perm, const int* col_perm, bool left_orth,
                               double* left, double* right, hipStream_t stream);
// out[c + ldo*r] = in[r + ldi*c]  (rows x cols input)
void transpose_launch(const double* in, int rows, int cols, int ldi, double* out, int ldo, hipStream_t stream);

// Batched left-side triangular solve, one problem per descriptor:  T X = B, X overwrites B.
//  T: n x n (ldt), lower or upper, optional unit diagonal.  B: n x nrhs (ldb).
struct TrsmProblem {
    const double* T; int ldt; int n;
    double* B; int ldb; int nrhs;
    int lower; int unit_diag;
    const int* skip_flag;   // optional: the problem is skipped when *skip_flag != 0
};
void trsm_left_batched_launch(const TrsmProblem* d_problems, int n_problems, int max_n, int max_nrhs, hipStream_t stream);

// Batched partial-pivot LU in place (one workgroup per problem): A = P^T L U, piv[k] = row swapped with k.
struct LuProblem {
    double* A; int lda; int n;
    int* piv;        // [n]
    int* info;       // [1]: 0 ok, k+1 = exactly-zero pivot at step k
    double* B; int ldb; int nrhs; // right-hand sides to which the row swaps are applied (may be null)
    const unsigned long long* pmax_bits; // optional: bits of max|a_ij|; below EPS the problem is flagged info = -1
};
void lu_batched_launch(const LuProblem* d_problems, int n_problems, int max_n, hipStream_t stream);
// Blocked variant that also applies the forward substitution to the right-hand sides: A = P^T L U, B <- L^{-1} P B
// (bitwise the result of lu_batched_launch followed by the unit-lower triangular solve).  Returns false when
// max_n > 1024 (nothing was launched; use the two-step path).
// tickets (optional): LU_MAX_PANEL_STEPS zeroed counters; the trailing updates then hand their work items out dynamically and the
// workgroups that land on XCD avoid_xcc (>= 0) return at once (see lu_update_kernel).
constexpr int LU_MAX_PANEL_STEPS = 128;
bool lu_forward_blocked_launch(const LuProblem* d_problems, int n_problems, int max_n, int max_nrhs, hipStream_t stream, int avoid_xcc = -1,
                               unsigned* tickets = nullptr);
// Round 5: the whole solve, B <- A^{-1} B: blocked LU of A alone (panel + trailing update of the factor's own column tiles), then ONE
// launch that gathers the right-hand sides through the composed row permutation and runs both triangular solves with each chunk of
// columns resident in the LDS (lu_solve_kernel).  Returns false when the sizes are outside its range (nothing was launched: use
// lu_forward_blocked_launch + the upper trsm_left_batched_launch).  tickets / avoid_xcc as above (the last counter serves the solve).
bool lu_solve_blocked_launch(const LuProblem* d_problems, int n_problems, int max_n, int max_nrhs, hipStream_t stream, int avoid_xcc = -1,
                             unsigned* tickets = nullptr);

// gather rows/cols:  out[i + ldo*j] = in[rows[i] + ldi*cols[j]] (rows/cols may be nullptr = identity)
void gather_launch(const double* in, int ldi, const int* rows, int nrows, const int* cols, int ncols, double* out,
                   int ldo, hipStream_t stream);
// scatter rows: out[rows[i] + ldo*j] = in[i + ldi*j];  scatter cols: out[i + ldo*cols[j]] = in[i + ldi*j]
void scatter_rows_launch(const double* in, int ldi, const int* rows, int nrows, int ncols, double* out, int ldo,
                         hipStream_t stream);
void scatter_cols_launch(const double* in, int ldi, int nrows, const int* cols, int ncols, double* out, int ldo,
                         hipStream_t stream);
void fill_launch(double* p, size_t count, double value, hipStream_t stream);
// C (M x N, ldc) = A (M x K, lda) * B (K x N, ldb) with every entry accumulated k-ascending, multiply and add rounded
// separately: the order the oracle's mat_mul restates for the reference's third-party matmul (aci.hip)
void seq_matmul_launch(const double* A, int lda, const double* B, int ldb, double* C, int ldc, size_t M, size_t N, size_t K,
                       hipStream_t stream);
// out (m x r): identity on top (r x r) and zeros below
void set_identity_launch(double* p, int m, int n, int ld, hipStream_t stream);

// Core packing (tensorci2.rs:1957-1999): see tci2.hip
// TT evaluation for a batch of points: cores[s] is (l_s, d_s, r_s) col-major; idx is n_sites x n_pts (uint32).
struct TtCoreDesc { const double* data; int l, d, r; };
void tt_evaluate_launch(const TtCoreDesc* d_cores, int n_sites, int max_bond, const uint32_t* d_idx, int n_pts,
                        double* d_out, hipStream_t stream);


// ------------------------------------------------------------------------------------------------
// kernels_tt.hip — tensor-train reshapes, sum / norm2, TTCache environments
// ------------------------------------------------------------------------------------------------
// mode 0: core -> left matrix (row l*S+s)   1: left matrix -> core   2: core -> right matrix (col s*R+r)   3: back
void core_reshape_launch(const double* in, int L, int S, int R, int mode, double* out, hipStream_t stream);
void tt_sum_launch(const TtCoreDesc* d_cores, int n_sites, int max_bond, double* d_out, hipStream_t stream);
void tt_norm2_step_launch(const TtCoreDesc& core, const double* d_cur, bool first, double* d_nxt, hipStream_t stream);
void tt_env_left_launch(const TtCoreDesc* d_cores, int split, int max_bond, const uint32_t* d_idx, int n_items,
                        double* d_out, int ld, hipStream_t stream);
void tt_env_right_launch(const TtCoreDesc* d_cores, int n_sites, int split, int max_bond, const uint32_t* d_idx,
                         int n_items, double* d_out, int ld, hipStream_t stream);
void tt_env_dot_launch(const double* d_left, const double* d_right, int len, int ld, const uint32_t* d_il,
                       const uint32_t* d_ir, size_t n_pts, double* d_out, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// kernels_small.hip — the small-problem engine (round 6): the WHOLE optimize_with_finder loop of a small TensorCI2 problem
// (tensorci2.rs:1626-1802: iteration loop, update_pivots chain :1821-2007, fill_site_tensors :1065-1186, convergence_criterion
// :1407-1437, final sweep1site :1787-1794) in ONE launch.  Index sets, history snapshots and lists live in the LDS, the candidate
// matrix of a bond in the registers of one wavefront (8 x 8: one entry per lane, 16 x 16: four, 32 x 32: sixteen).
// When a list or a matrix outgrows the tile the kernel stops at the start of that iteration and hands the state back (tci2_small.hip
// continues with the general path from there).
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_CAP = 16;        // entries per index set
constexpr int SMALL_TILE = 32;       // candidate matrices up to SMALL_TILE x SMALL_TILE
constexpr int SMALL_MAX_SITES = 32;
constexpr int SMALL_MAX_ITER = 64;   // iterations whose errors / ranks the result block holds
constexpr int SMALL_MAX_W = 512;     // K * total weights
struct SmallHeader { // travels in the kernel arguments; offsets in bytes from the start of the (pinned) input block
    int n, K, fid, total;
    int max_iter, ncheck, sweep_strategy, flags; // flags: 1 normalize_error, 2 strictly_nested, 4 final_sweep1site, 8 phase stamps, 16 PivotSearchStrategy::Rook
    int max_bond_dim, cap_in, tile_max, pad1;    // cap_in: entries per (family, site) in the input tables (<= SMALL_CAP); tile_max: rows / columns of a candidate matrix the launch takes (16 or 32)
    double tolerance, max_sample_value;
    double params[T4A_FN_MAX_PARAMS];
    // ldim[n] woff[n] | w[K*total] | cnt[2n] (I sets then J sets) | code[2n][cap_in] | acc[2n][cap_in][K] | cores[n] (device pointers)
    int o_ldim, o_woff, o_w, o_cnt, o_code, o_acc, o_cores, bytes;
};
// result block (pinned, written by the kernel): header, then arrays at the offsets small_out_layout() gives
struct SmallOutHeader {
    int status;      // 1: the whole call completed (incl. the final 1-site sweep when asked for)  2: handed over  3: failed (restart on the general path)
    int iters_done;  // iterations completed by the kernel (the state handed back is the one at the start of iteration iters_done)
    int converged, termination, n_pivot_errors, final_done, hist_valid, reason;
    double max_sample_value;
    unsigned long long clocks[12]; // [0..2] 100 MHz ticks: input, loop, results; [3..10] shader cycles per phase when flag 8 is set (lists, evaluation, pivot steps, gather, factors, fill, snapshots, convergence)
};
struct SmallOutLayout {
    size_t o_err, o_rank, o_bond, o_pe, o_shapes, o_cdims, o_cnt, o_code, o_hcnt, o_hcode, o_flag, bytes;
};
__host__ __device__ inline SmallOutLayout small_out_layout(int n)
{
    SmallOutLayout L;
    size_t o = (sizeof(SmallOutHeader) + 15) / 16 * 16;
    L.o_err = o;    o += sizeof(double) * SMALL_MAX_ITER;
    L.o_rank = o;   o += sizeof(int) * SMALL_MAX_ITER;
    L.o_bond = o;   o += sizeof(double) * (size_t)n;
    L.o_pe = o;     o += sizeof(double) * (SMALL_TILE + 2);
    L.o_shapes = o; o += sizeof(int) * 3 * (size_t)n + 4;
    o = (o + 15) / 16 * 16;
    L.o_cdims = o;  o += sizeof(int) * 3 * (size_t)n + 4;
    o = (o + 15) / 16 * 16;
    L.o_cnt = o;    o += sizeof(int) * 2 * (size_t)n;
    o = (o + 15) / 16 * 16;
    L.o_code = o;   o += sizeof(uint64_t) * 2 * (size_t)n * SMALL_CAP;
    L.o_hcnt = o;   o += sizeof(int) * 2 * (size_t)n;
    o = (o + 15) / 16 * 16;
    L.o_hcode = o;  o += sizeof(uint64_t) * 2 * (size_t)n * SMALL_CAP;
    L.o_flag = o;   o += 16;
    L.bytes = o;
    return L;
}
struct SmallArgs {
    SmallHeader h;
    const char* in;    // the arrays the header points into (pinned host memory, read once)
    char* out;         // SmallOutHeader + arrays (pinned host memory)
    double* scratch;   // device: n * SMALL_CAP * dmax * SMALL_CAP doubles (site tensors of the iterations whose cores nobody reads)
    size_t scratch_stride;
    unsigned token;    // written to the completion flag
};
size_t small_lds_bytes(int n, int K, int total);   // dynamic LDS the launch needs (0: does not fit a compute unit)
void small_optimize_launch(const SmallArgs& a, int n, int K, int total, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// kernels_linalg.hip — one-sided Jacobi SVD and Householder QR building blocks
// ------------------------------------------------------------------------------------------------
void nonfinite_flag_launch(const double* data, size_t count, int* d_flag, hipStream_t stream);
// the same pass also returns the largest magnitude (bit pattern of the double, *d_absmax_bits zeroed by the caller)
void nonfinite_absmax_launch(const double* data, size_t count, int* d_flag, unsigned long long* d_absmax_bits, hipStream_t stream);
void scale_pow2_launch(double* dst, const double* src, size_t count, int e, hipStream_t stream); // dst = src * 2^e, exact
// The exponent Engine::svd / Engine::qr scale their input by (2^-e A is decomposed, the factor goes back on S / R): 0 for a largest
// magnitude inside 2^-200 .. 2^200 and for a zero matrix.  Squared column norms — the Jacobi pair test's alpha * beta, the Householder
// norms — overflow from ~1e77 / ~1e154 on and underflow below ~1e-154: without the scaling such inputs gave wrong factors and no error.
__host__ __device__ inline int pow2_scale_exponent(double amax) { return (amax > 0.0 && (amax > 0x1p200 || amax < 0x1p-200)) ? ilogb(amax) : 0; }
// dst = src * 2^(sign * e) with e = pow2_scale_exponent of the magnitude nonfinite_absmax_launch left in *d_absmax_bits (device side)
void scale_pow2_dev_launch(double* dst, const double* src, size_t count, const unsigned long long* d_absmax_bits, int sign, hipStream_t stream);
bool jacobi_fits_small(int m, int n);
// all sweeps inside one workgroup (m >= n, n <= 128)
void jacobi_small_launch(double* W, int m, double* V, int n, int max_sweeps, hipStream_t stream);
// all sweeps inside one workgroup, a group of 8 or 16 lanes per column pair, W and V in the LDS (m >= n, n <= 96; see jg_plan);
// V is an output only, d_nonfinite[0] is set for an Inf / NaN input, d_nonfinite[1] receives the sweep count, d_nonfinite[2] the exponent e
// when the kernel iterated on (and returns) 2^-e W
bool jacobi_fits_groups(int m, int n);
bool jacobi_groups_launch(double* W, int m, double* V, int n, int max_sweeps, int* d_nonfinite, hipStream_t stream);
// one full sweep = n-1 tournament rounds, one launch per round; *d_rotated is set when any pair rotated
void jacobi_sweep_launch(double* W, int m, double* V, int n, int* d_rotated, hipStream_t stream);
void jacobi_sweep_end_launch(int* d_flags, hipStream_t stream); // flags [0] rotated [3] converged (kernels_linalg.hip)
// one full sweep of the BLOCKED iteration: a tournament over column blocks, one launch per block round, the pairs of a block pair
// rotated inside one workgroup's LDS (false: the columns do not fit the LDS, nothing was launched)
bool jacobi_block_sweep_launch(double* W, int m, double* V, int n, int* d_rotated, hipStream_t stream);
// sigma = column norms, sorted non-increasing; U = W / sigma, Vs = V gathered; dead[j] = 1 for sigma == 0
void svd_finalize_launch(const double* W, int m, const double* V, int n, double* sig_tmp, double* U, double* S,
                         double* Vs, int* d_dead, int* d_ndead, hipStream_t stream);
void svd_complete_launch(double* U, int m, int n, int* d_dead, double* tmp_m, hipStream_t stream);
// Blocked Householder QR in place (compact WY, panels of QR_PANEL columns): reflectors stay below the diagonal of A, R's diagonal
// goes to diag[]; Vall (m x k) receives the explicit reflector matrix, Tall (qr_panels(k) blocks of QR_PANEL x QR_PANEL) the
// triangular factors; W, W2: QR_PANEL x max(n, k) scratch each.
constexpr int QR_PANEL = 32;
inline int qr_panels(int k) { return (k + QR_PANEL - 1) / QR_PANEL; }
void qr_factor_launch(double* A, int m, int n, double* diag, double* tau, double* v0s, double* Vall, double* Tall, double* W, double* W2,
                      hipStream_t stream);
void qr_form_launch(const double* A, int m, int n, const double* diag, const double* Vall, const double* Tall, double* W, double* W2, double* Q,
                    double* R, hipStream_t stream);

} // namespace t4a
