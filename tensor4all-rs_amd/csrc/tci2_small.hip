// tci2_small.hip — host side of the small-problem engine (kernels_small.hip): optimize_with_finder (tensorci2.rs:1626-1802) of a
// small problem as ONE launch.  Tci2::opt_begin offers every optimize() call to it first:
//
//   * eligible: built-in functor with at most two accumulators, PivotSearchStrategy::Full, no global pivot search (nsearch == 0 or
//     max_nglobal_pivot == 0: the finder cannot add a pivot), a fresh history, index sets of at most SMALL_CAP entries, the tables
//     fit the LDS of one compute unit;
//   * the kernel runs iterations until the run converges (then the final 1-site sweep as well) or until a list or a candidate
//     matrix outgrows its tile; in that case it hands back the state AT THE START of that iteration and the general path
//     (bond chain / per-bond path) continues from there — the first iterations of every run are small;
//   * the host reads ONE result block: index sets as codes (decoded here), the history snapshot, errors, ranks, termination; the
//     site tensors are written to the handle's device buffers by the kernel.
//
// Nothing here changes a result (tests/test_gpu_small.py runs engine, general path and oracle side by side).
#include "tci2.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace t4a {

bool Tci2::small_engine_eligible(const TCI2Options& options) const
{
    static const bool no_chain = std::getenv("T4A_NO_CHAIN") != nullptr; // (the documented "everything bond by bond" fallback switches the engine off too)
    if (no_chain || !small_enabled || !chain_enabled || chain_verify || chain_event_timing) return false;
    if (fn_kind_ != FnKind::Builtin || fn_dev_.n_acc > 2 || (options.pivot_search != 0 && options.pivot_search != 1)) return false;
    if (options.pivot_search == 1) {
        static const bool rook_host = std::getenv("T4A_ROOK_HOST") != nullptr; // (A/B of the search drivers: rook.hip's host-driven loop is asked for)
        if (rook_host) return false;
    }
    if (!(options.nsearch == 0 || options.max_nglobal_pivot == 0)) return false;
    if (!history.empty() || shard_world != 1 || keep_site_tensors || pi_shard.active()) return false;
    if (n_ > (size_t)SMALL_MAX_SITES || options.max_iter > (size_t)SMALL_MAX_ITER || options.ncheck_history > (size_t)SMALL_MAX_ITER) return false;
    if (small_lds_bytes((int)n_, fn_dev_.n_acc, (int)total_) == 0) return false;
    long double space = 1.0L;
    for (size_t d : local_dims) {
        if (d > (size_t)SMALL_TILE) return false;
        space *= (long double)d;
    }
    if (space >= 9.0e18L) return false; // codes are 63-bit mixed-radix numbers
    for (size_t p = 0; p < n_; ++p)
        if (i_set[p].count < 1 || j_set[p].count < 1 || i_set[p].count > (size_t)SMALL_CAP || j_set[p].count > (size_t)SMALL_CAP) return false;
    return true;
}

// Returns true when the engine ran (the handle's state and `r` were advanced); r.small_complete: nothing is left to do.
bool Tci2::small_engine_run(OptRun& r)
{
    const TCI2Options& options = r.options;
    if (!r.allow_small || !small_engine_eligible(options)) {
        ++small_stats[3];
        return false;
    }
    sync_digits();
    const int n = (int)n_, K = fn_dev_.n_acc, total = (int)total_;
    size_t cap_in = 1;
    for (size_t p = 0; p < n_; ++p) cap_in = std::max(cap_in, std::max(i_set[p].count, j_set[p].count));
    // ---- input block ----
    SmallArgs a{};
    SmallHeader& h = a.h;
    h.n = n;
    h.K = K;
    h.fid = fn_dev_.fid;
    h.total = total;
    h.max_iter = (int)options.max_iter;
    h.ncheck = (int)options.ncheck_history;
    h.sweep_strategy = options.sweep_strategy;
    h.flags = (options.normalize_error ? 1 : 0) | (options.strictly_nested ? 2 : 0) | (r.final_sweep1site ? 4 : 0) | (small_stamps ? 8 : 0) | (options.pivot_search == 1 ? 16 : 0);
    h.max_bond_dim = (int)std::min<size_t>(options.max_bond_dim_or_max(), (size_t)1 << 30);
    h.cap_in = (int)cap_in;
    h.tile_max = small_tile_max;
    h.tolerance = options.tolerance;
    h.max_sample_value = max_sample_value;
    std::memcpy(h.params, fn_dev_.params, sizeof(h.params));
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 15) / 16 * 16;
        return at;
    };
    h.o_ldim = (int)take(sizeof(int) * n_);
    h.o_woff = (int)take(sizeof(int) * n_);
    h.o_w = (int)take(sizeof(uint64_t) * (size_t)K * total_);
    h.o_cnt = (int)take(sizeof(int) * 2 * n_);
    h.o_code = (int)take(sizeof(uint64_t) * 2 * n_ * cap_in);
    h.o_acc = (int)take(sizeof(uint64_t) * 2 * n_ * cap_in * (size_t)K);
    h.o_cores = (int)take(sizeof(double*) * n_);
    h.bytes = (int)o;
    small_in_.reserve(o);
    char* in = small_in_.get();
    {
        int* ldim = reinterpret_cast<int*>(in + h.o_ldim);
        int* woff = reinterpret_cast<int*>(in + h.o_woff);
        for (size_t p = 0; p < n_; ++p) {
            ldim[p] = (int)local_dims[p];
            woff[p] = (int)offset_[p];
        }
        std::memcpy(in + h.o_w, weights_.data(), sizeof(uint64_t) * (size_t)K * total_);
        int* cnt = reinterpret_cast<int*>(in + h.o_cnt);
        uint64_t* code = reinterpret_cast<uint64_t*>(in + h.o_code);
        uint64_t* acc = reinterpret_cast<uint64_t*>(in + h.o_acc);
        std::memset(code, 0, sizeof(uint64_t) * 2 * n_ * cap_in);
        std::memset(acc, 0, sizeof(uint64_t) * 2 * n_ * cap_in * (size_t)K);
        std::vector<uint64_t> av;
        for (int side = 0; side < 2; ++side)
            for (size_t p = 0; p < n_; ++p) {
                const IndexSet& s = side == 0 ? i_set[p] : j_set[p];
                const size_t first = side == 0 ? 0 : p + 1;
                const size_t fp = (size_t)side * n_ + p;
                cnt[fp] = (int)s.count;
                accumulate(s, first, av);
                for (size_t k = 0; k < s.count; ++k) code[fp * cap_in + k] = code_of(s.at(k), first, s.width, side == 0);
                std::memcpy(acc + fp * cap_in * (size_t)K, av.data(), av.size() * sizeof(uint64_t));
            }
        double** cptr = reinterpret_cast<double**>(in + h.o_cores);
        size_t dmax = 1;
        for (size_t p = 0; p < n_; ++p) {
            dmax = std::max(dmax, local_dims[p]);
            cores[p].buf.reserve((size_t)SMALL_CAP * local_dims[p] * (size_t)SMALL_CAP);
            cptr[p] = cores[p].buf.get();
        }
        a.scratch_stride = (size_t)SMALL_CAP * dmax * (size_t)SMALL_CAP;
        small_scratch_.reserve(a.scratch_stride * n_);
        a.scratch = small_scratch_.get();
    }
    const SmallOutLayout OL = small_out_layout(n);
    small_out_.reserve(OL.bytes);
    char* out = small_out_.get();
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(out + OL.o_flag);
    a.in = in;
    a.out = out;
    a.token = ++small_token_;
    if (a.token == 0) a.token = ++small_token_;
    *flag = 0u;
    hipStream_t st = eng.stream();
    small_optimize_launch(a, n, K, total, st);
    T4A_HIP(hipGetLastError());
    // ---- wait: the kernel's last store is the token (system scope); a few microseconds sooner than the stream's completion signal ----
    {
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spin = 0;; ++spin) {
            if (*flag == a.token) {
                seen = true;
                break;
            }
            if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        }
        if (!seen) T4A_HIP(hipStreamSynchronize(st)); // (a fault surfaces here)
        if (*flag != a.token) throw Error(T4A_GPU_INTERNAL_ERROR, "small-problem engine: the launch did not complete");
    }
    const SmallOutHeader oh = *reinterpret_cast<const SmallOutHeader*>(out);
    for (int q = 0; q < 12; ++q) small_last_clocks_[q] = oh.clocks[q];
    small_last_reason_ = oh.reason;
    rook_work_.n_device_searches += (size_t)(oh.clocks[11] >> 32); // rook bonds the launch ran device-resident
    rook_work_.n_device_visits += (size_t)(oh.clocks[11] & 0xFFFFFFFFull);
    if (oh.status == 3) { // a site tensor could not be filled inside the launch: the general path runs the call from the start
        ++small_stats[2];
        return false;
    }
    if (oh.status != 1 && oh.status != 2) throw Error(T4A_GPU_INTERNAL_ERROR, "small-problem engine: unexpected status");
    const size_t iters = (size_t)oh.iters_done;
    if (oh.status == 2 && iters == 0 && !oh.final_done) { // nothing was advanced: the general path starts from the state as it is
        ++small_stats[2];
        return false;
    }
    // ---- the state the reference would hold now ----
    const int* cnt = reinterpret_cast<const int*>(out + OL.o_cnt);
    const uint64_t* code = reinterpret_cast<const uint64_t*>(out + OL.o_code);
    for (size_t p = 0; p < n_; ++p) {
        i_set[p].count = (size_t)cnt[p];
        j_set[p].count = (size_t)cnt[n_ + p];
        decode_set(i_set[p], code + p * SMALL_CAP, 0, true);
        decode_set(j_set[p], code + (n_ + p) * SMALL_CAP, p + 1, false);
    }
    history.clear();
    if (oh.hist_valid) { // the sets at the start of the last completed iteration (tensorci2.rs:1686-1689): codes, decoded on demand
        const int* hcnt = reinterpret_cast<const int*>(out + OL.o_hcnt);
        const uint64_t* hcode = reinterpret_cast<const uint64_t*>(out + OL.o_hcode);
        HistEntry e;
        e.serial = ++chain_.hist_serial;
        e.digits_valid = false;
        e.cap = SMALL_CAP;
        e.is.resize(n_);
        e.js.resize(n_);
        for (size_t p = 0; p < n_; ++p) {
            e.is[p].width = p;
            e.is[p].count = (size_t)hcnt[p];
            e.js[p].width = n_ - p - 1;
            e.js[p].count = (size_t)hcnt[n_ + p];
        }
        e.code.assign(hcode, hcode + 2 * n_ * SMALL_CAP);
        history.push_back(std::move(e));
    }
    const double* berr = reinterpret_cast<const double*>(out + OL.o_bond);
    for (size_t b = 0; b + 1 < n_; ++b) bond_errors[b] = berr[b];
    max_sample_value = oh.max_sample_value;
    const double* errs = reinterpret_cast<const double*>(out + OL.o_err);
    const int* rks = reinterpret_cast<const int*>(out + OL.o_rank);
    errors_hist.assign(errs, errs + iters);
    ranks_hist.assign(iters, 0);
    for (size_t k = 0; k < iters; ++k) ranks_hist[k] = (size_t)rks[k];
    r.nglobal_hist.assign(iters, 0);
    r.iter = iters;
    r.done = oh.converged != 0;
    if (oh.converged) termination = oh.termination;
    pivot_errors.clear();
    if (iters > 0) {
        const int* sh = reinterpret_cast<const int*>(out + OL.o_shapes);
        last_sweep_shapes.assign(n_ - 1, {0, 0, 0});
        for (size_t b = 0; b + 1 < n_; ++b) last_sweep_shapes[b] = {(size_t)sh[3 * b], (size_t)sh[3 * b + 1], (size_t)sh[3 * b + 2]};
    }
    const bool loop_over = r.done || iters >= options.max_iter;
    // (without the final sweep the reference leaves every site tensor invalidated: add_global_pivots of the last iteration, tensorci2.rs:707-708)
    const bool cores_valid = oh.status == 1 && oh.final_done;
    if (oh.final_done) {
        const double* pe = reinterpret_cast<const double*>(out + OL.o_pe);
        pivot_errors.assign(pe, pe + oh.n_pivot_errors);
        r.final_sweep1site = false; // done in the launch
    }
    if (cores_valid) {
        const int* cd = reinterpret_cast<const int*>(out + OL.o_cdims);
        for (size_t b = 0; b < n_; ++b) {
            cores[b].l = (size_t)cd[3 * b];
            cores[b].s = (size_t)cd[3 * b + 1];
            cores[b].r = (size_t)cd[3 * b + 2];
        }
    } else {
        invalidate_site_tensors();
    }
    mark_sets_changed();
    invalidate_fill_cache();
    prep_.valid = false;
    r.small_complete = oh.status == 1 && loop_over;
    // statistics: the engine's half-sweeps count as chained, persistent half-sweeps (nothing of them ran with the host in the loop)
    small_stats[0] += r.small_complete ? 1 : 0;
    small_stats[1] += iters;
    small_stats[2] += oh.status == 2 ? 1 : 0;
    chain_stats[0] += iters;
    chain_stats[1] += iters * (n_ - 1);
    chain_stats_ext[0] += iters + (oh.final_done ? 1 : 0);
    chain_stats_ext[1] += oh.final_done ? 1 : 0;
    return true;
}

} // namespace t4a
