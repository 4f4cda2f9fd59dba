// tci2_chain.hip — the host-free half-sweep of the TCI2 driver ("bond chain").
//
// update_pivots (tensorci2.rs:1821-2007) of bond b needs the pivots bond b-1 (forward) / b+1 (backward) selected, so a
// half-sweep (tensorci2.rs:1695-1725) is a chain of dependent steps.  tci2.hip runs it bond by bond with the host in the loop:
// build the row / column sets, upload accumulators, three launches, wait, read the permutations.  Here the whole half-sweep
// is ENQUEUED at once for built-in functors (kernels_chain.hip has the device side):
//
//   * the index sets live on the device as tables of (code, accumulators); every rrLU's pivots are gathered into them — and
//     into a pinned host mirror — by the preparation kernel of the next bond.  The host's master copy of I / J after a chain
//     IS that mirror: counts are read back, digit tables are decoded from the codes only when somebody asks for them
//     (sync_digits: getters, the per-bond path, global pivots), the fill accumulators are copied from it;
//   * between two rrLU launches only the preparation kernel runs (gather + dependent side of the next bond); the side that
//     does not depend on the chain is built for all bonds by one launch up front, the candidate matrix of bond b is
//     evaluated speculatively while the rrLU of the previous bond still runs — by that launch's own pass-through workgroups
//     (the single-XCD kernel launches 8 W workgroups of which 7 W land on the other XCDs and used to return at once) — and
//     read through a row map; no second stream, no events: two launches per bond;
//   * launches are planned for upper bounds that follow from the set sizes alone; kernels read the real dimensions on the
//     device;
//   * optimize() enqueues the chain of iteration t + 1 before it issues fill_site_tensors of iteration t (two mirrors
//     alternate), so the host work of the fill overlaps the device's bond updates as well.
//
// Nothing here changes a result: the row / column lists are the same lists in the same order, the rrLU kernels are the
// bit-exact ones of engine.hip (tests/test_gpu_chain.py runs chain, per-bond path and oracle side by side; with set_chain(verify)
// the decoded mirror is checked against codes and accumulators recomputed on the host and against the device tables).  The LUCI
// factors of a bond are not built: every caller (sweep2site :746-798, optimize_with_finder :1659-1776) overwrites all site
// tensors with fill_site_tensors right after the half-sweep.
//
// Not eligible (the per-bond path of tci2.hip runs instead): host callbacks, PivotSearchStrategy::Rook, index spaces beyond
// 63 bits, sets beyond CHAIN_MAX_SET entries, bonds whose upper-bound shape no device-dimension kernel takes
// (> 1024 rows / columns: the chip-wide kernels), T4A_NO_CHAIN=1, set_chain(false).
#include "tci2.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace t4a {

namespace {
size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }
// chains in flight in this process (handles on several host threads: one XCD each).  The speculative candidate matrix rides on
// the pass-through workgroups of an rrLU launch, i.e. on the OTHER XCDs: fine while those are idle, a nuisance when other
// handles factorise there (their launches' pass-through workgroups then queue for the few free compute units of every XCD).
// A chain that is not alone evaluates its candidate matrices directly (8 us more per bond, nobody else disturbed).
std::atomic<int> g_chains_inflight{0};
} // namespace
thread_local double g_chain_wait_seconds = 0.0; // time chain_finish spent waiting for the device (T4A_GROUP_PROF, tci2.hip)

bool Tci2::chain_usable(const TCI2Options& options) const
{
    static const bool off = std::getenv("T4A_NO_CHAIN") != nullptr;
    if (off || !chain_enabled || fn_kind_ != FnKind::Builtin || options.pivot_search != 0) return false;
    long double space = 1.0L;
    for (size_t d : local_dims) space *= (long double)d;
    if (space >= 9.0e18L) return false; // codes are 63-bit mixed-radix numbers
    for (size_t p = 0; p < n_; ++p) {
        if (i_set[p].count == 0 || j_set[p].count == 0) return false;
        if (i_set[p].count > (size_t)CHAIN_MAX_SET || j_set[p].count > (size_t)CHAIN_MAX_SET) return false;
    }
    return true;
}

// prefix sets I_p (sites 0 .. width-1): code = v[w-1] + d_{w-1} (v[w-2] + d_{w-2} ( ... )); suffix sets J_p (sites first ..
// first+width-1): code = v[0] + d_first (v[1] + d_{first+1} ( ... )).  Either way the Kronecker step of kernels_chain.hip is
// code(child) = s + d * code(parent).
uint64_t Tci2::code_of(const uint32_t* v, size_t first_site, size_t width, bool prefix) const
{
    uint64_t c = 0;
    if (prefix) {
        for (size_t s = 0; s < width; ++s) c = (uint64_t)v[s] + (uint64_t)local_dims[first_site + s] * c;
    } else {
        for (size_t s = width; s-- > 0;) c = (uint64_t)v[s] + (uint64_t)local_dims[first_site + s] * c;
    }
    return c;
}

// inverse of code_of for a whole set (s.width and s.count are given)
void Tci2::decode_set(IndexSet& s, const uint64_t* codes, size_t first_site, bool prefix) const
{
    const size_t w = s.width;
    s.d.assign(s.count * w, 0u);
    for (size_t k = 0; k < s.count; ++k) {
        uint64_t c = codes[k];
        uint32_t* v = s.d.data() + k * w;
        if (prefix) {
            for (size_t q = w; q-- > 0;) {
                const uint64_t d = local_dims[first_site + q];
                v[q] = (uint32_t)(c % d);
                c /= d;
            }
        } else {
            for (size_t q = 0; q < w; ++q) {
                const uint64_t d = local_dims[first_site + q];
                v[q] = (uint32_t)(c % d);
                c /= d;
            }
        }
    }
}

ChainTab Tci2::chain_tab(int family) const
{
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const size_t fam = n_ * cap * (1 + K);
    ChainTab t;
    t.code = chain_.tab.get() + (size_t)family * fam;
    t.acc = t.code + n_ * cap;
    t.cnt = chain_.cnt.get() + (size_t)family * n_;
    return t;
}

ChainTab Tci2::chain_mirror(int which, int family) const
{
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const size_t fam = n_ * cap * (1 + K);
    ChainTab t;
    t.code = chain_.mtab.get() + ((size_t)which * 2 + (size_t)family) * fam;
    t.acc = t.code + n_ * cap;
    t.cnt = chain_.mcnt.get() + ((size_t)which * 2 + (size_t)family) * n_;
    return t;
}

void Tci2::sync_digits()
{
    if (!chain_.digits_stale) return;
    const size_t cap = chain_.cap;
    const ChainTab mi = chain_mirror(chain_.mcur, 0), mj = chain_mirror(chain_.mcur, 1);
    for (size_t p = 0; p < n_; ++p) {
        if ((size_t)mi.cnt[p] != i_set[p].count || (size_t)mj.cnt[p] != j_set[p].count)
            throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: mirror counts disagree with the host's sets");
        decode_set(i_set[p], mi.code + p * cap, 0, true);
        decode_set(j_set[p], mj.code + p * cap, p + 1, false);
    }
    chain_.digits_stale = false;
}

void Tci2::hist_digits(HistEntry& e)
{
    if (e.digits_valid) return;
    for (size_t p = 0; p < n_; ++p) {
        decode_set(e.is[p], e.code.data() + p * e.cap, 0, true);
        decode_set(e.js[p], e.code.data() + (n_ + p) * e.cap, p + 1, false);
    }
    e.digits_valid = true;
}

void Tci2::chain_layout(size_t cap)
{
    const int K = fn_dev_.n_acc;
    if (cap <= chain_.cap && K == chain_.n_acc && chain_.tab.get()) return;
    sync_digits(); // (the mirror is about to move: the digit tables take over as the master copy)
    cap = std::max(cap, chain_.cap);
    chain_.cap = cap;
    chain_.n_acc = K;
    const size_t fam = n_ * cap * (1 + (size_t)K);
    chain_.tab.reserve(6 * fam);
    chain_.cnt.reserve(6 * n_);
    chain_.mtab.reserve(4 * fam);
    chain_.mcnt.reserve(4 * n_);
    chain_.tables_valid = false;
    chain_.snap_serial[0] = chain_.snap_serial[1] = ~0ull;
}

// host sets -> current pinned mirror -> device tables (families 0, 1).  Rare: first use, after the host changed the sets.
void Tci2::chain_upload_current()
{
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const size_t fam = n_ * cap * (1 + K);
    hipStream_t st = eng.stream();
    const ChainTab mi = chain_mirror(chain_.mcur, 0), mj = chain_mirror(chain_.mcur, 1);
    if (!chain_.digits_stale) { // (otherwise the mirror already is the master copy)
        std::vector<uint64_t> a;
        for (size_t p = 0; p < n_; ++p) {
            for (int side = 0; side < 2; ++side) {
                const IndexSet& s = side == 0 ? i_set[p] : j_set[p];
                const ChainTab& m = side == 0 ? mi : mj;
                if (s.count > cap) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: index set larger than the device tables");
                const size_t first = side == 0 ? 0 : p + 1;
                accumulate(s, first, a);
                for (size_t k = 0; k < s.count; ++k) m.code[p * cap + k] = code_of(s.at(k), first, s.width, side == 0);
                std::memcpy(m.acc + p * cap * K, a.data(), a.size() * sizeof(uint64_t));
                m.cnt[p] = (int)s.count;
            }
        }
    }
    // the other mirror gets the same content: a chain writes every entry but I_0 and J_{n-1}, which never change
    const ChainTab oi = chain_mirror(1 - chain_.mcur, 0);
    std::memcpy(oi.code, mi.code, 2 * fam * sizeof(uint64_t));
    std::memcpy(oi.cnt, mi.cnt, 2 * n_ * sizeof(int));
    T4A_HIP(hipMemcpyAsync(chain_tab(0).code, mi.code, 2 * fam * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    T4A_HIP(hipMemcpyAsync(chain_tab(0).cnt, mi.cnt, 2 * n_ * sizeof(int), hipMemcpyHostToDevice, st));
    T4A_HIP(hipStreamSynchronize(st));
    chain_.tables_valid = true;
}

// a history entry -> device snapshot slot (only when the slot that was filled on the device is gone)
void Tci2::chain_upload_hist(HistEntry& e, int slot)
{
    hist_digits(e);
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const size_t fam = n_ * cap * (1 + K);
    std::vector<uint64_t> buf(2 * fam, 0);
    std::vector<int> cnt(2 * n_, 0);
    std::vector<uint64_t> a;
    for (size_t p = 0; p < n_; ++p)
        for (int side = 0; side < 2; ++side) {
            const IndexSet& s = side == 0 ? e.is[p] : e.js[p];
            if (s.count > cap) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: history set larger than the device tables");
            const size_t first = side == 0 ? 0 : p + 1;
            uint64_t* code = buf.data() + (size_t)side * fam;
            uint64_t* acc = code + n_ * cap;
            accumulate(s, first, a);
            for (size_t k = 0; k < s.count; ++k) code[p * cap + k] = code_of(s.at(k), first, s.width, side == 0);
            std::memcpy(acc + p * cap * K, a.data(), a.size() * sizeof(uint64_t));
            cnt[(size_t)side * n_ + p] = (int)s.count;
        }
    T4A_HIP(hipMemcpy(chain_tab(2 + 2 * slot).code, buf.data(), buf.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    T4A_HIP(hipMemcpy(chain_tab(2 + 2 * slot).cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
}

// accumulators fill_site_tensors needs for site b (J_b, kron(I_b, d_b), I_{b+1}; tensorci2.rs:1101-1145) from the mirror
void Tci2::prepare_fill_site_from_mirror(size_t b)
{
    if (fill_cache_.size() != n_) fill_cache_.assign(n_, FillAcc());
    if (shard_world > 1 && (b % shard_world) != shard_rank) return;
    FillAcc& f = fill_cache_[b];
    f.valid = false;
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const ChainTab mi = chain_mirror(chain_.mcur, 0), mj = chain_mirror(chain_.mcur, 1);
    const size_t ni = i_set[b].count, nj = j_set[b].count;
    if (ni == 0 || nj == 0) return;
    f.accJ.assign(mj.acc + b * cap * K, mj.acc + b * cap * K + nj * K);
    const size_t d = local_dims[b];
    f.accK.resize(ni * d * K);
    for (size_t i = 0; i < ni; ++i)
        for (size_t s = 0; s < d; ++s)
            for (size_t k = 0; k < K; ++k)
                f.accK[(i * d + s) * K + k] = mi.acc[(b * cap + i) * K + k] + weights_[k * total_ + offset_[b] + s];
    if (b + 1 < n_)
        f.accI.assign(mi.acc + (b + 1) * cap * K, mi.acc + (b + 1) * cap * K + i_set[b + 1].count * K);
    else
        f.accI.clear();
    f.valid = true;
}

bool Tci2::chain_enqueue(bool forward, const TCI2Options& options, long ext_idx, bool in_optimize, bool launch)
{
    if (chain_.inflight || chain_.prepared) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: a chain is already in flight");
    if (!chain_usable(options)) {
        ++(chain_.one_site ? chain_stats_ext[2] : chain_stats[3]);
        return false;
    }
    const size_t nb = n_ - 1;
    const size_t K = (size_t)fn_dev_.n_acc;
    const size_t chi = options.max_bond_dim_or_max();
    hipStream_t st = eng.stream();
    const bool one = chain_.one_site;
    if (one && (ext_idx >= 0 || in_optimize || !launch)) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: a 1-site sweep has no extras, no snapshot and no group");
    HistEntry* ext = ext_idx >= 0 ? &history[(size_t)ext_idx] : nullptr;

    // ---- 1. upper bounds of every bond's shape (set sizes only), launch plans ----
    std::vector<size_t> cI(n_), cJ(n_), eI(n_, 0), eJ(n_, 0);
    size_t need_cap = 1;
    bool use_extras = false;
    for (size_t p = 0; p < n_; ++p) {
        cI[p] = i_set[p].count;
        cJ[p] = j_set[p].count;
        if (ext) {
            eI[p] = ext->is[p].count;
            eJ[p] = ext->js[p].count;
        }
        use_extras |= eI[p] != 0 || eJ[p] != 0;
        need_cap = std::max({need_cap, cI[p], cJ[p], eI[p], eJ[p]});
    }
    std::vector<size_t> order(nb);
    for (size_t k = 0; k < nb; ++k) order[k] = forward ? k : nb - 1 - k;
    std::vector<size_t> dep_ub(nb), ind_ub(nb), lda_ub(nb, 0);
    std::vector<ChainRrluPlan> plans(nb);
    size_t Mcap = 1, Ncap = 1, steps_cap = 1, pi_cap = 0, spec_cap = 0, dep_cap = 1, ind_cap = 1;
    for (size_t k = 0; k < nb; ++k) {
        const size_t b = order[k];
        // (1-site sweep: the independent side is J_b / I_{b+1} itself, tensorci2.rs:931-944)
        const size_t Mub = one ? (forward ? cI[b] * local_dims[b] : cI[b + 1]) : cI[b] * local_dims[b] + eI[b + 1];
        const size_t Nub = one ? (forward ? cJ[b] : cJ[b + 1] * local_dims[b + 1]) : cJ[b + 1] * local_dims[b + 1] + eJ[b];
        const size_t rub = std::max<size_t>(std::min({Mub, Nub, chi}), 1);
        cI[b + 1] = rub; // bond b writes I_{b+1} and J_b
        cJ[b] = rub;
        need_cap = std::max(need_cap, rub);
        dep_ub[b] = forward ? Mub : Nub; // the kernel's rows are the dependent side in both directions
        ind_ub[b] = forward ? Nub : Mub;
        if (dep_ub[b] > 65535 || ind_ub[b] > 65535 || !eng.chain_plan((int)dep_ub[b], (int)ind_ub[b], &plans[b])) {
            ++(one ? chain_stats_ext[2] : chain_stats[3]);
            return false;
        }
        plans[b].code += forward ? 0 : 4; // (row-major tie order of the transposed problem)
        Mcap = std::max(Mcap, Mub);
        Ncap = std::max(Ncap, Nub);
        dep_cap = std::max(dep_cap, dep_ub[b]);
        ind_cap = std::max(ind_cap, ind_ub[b]);
        steps_cap = std::max(steps_cap, rub);
        const bool fused = plans[b].kind == 1 && plans[b].fused;
        if (!launch) pi_cap = std::max(pi_cap, dep_ub[b] * ind_ub[b]); // (a group chain evaluates every candidate matrix directly)
        if (!fused) {
            pi_cap = std::max(pi_cap, dep_ub[b] * ind_ub[b]); // (evaluated directly when the previous launch could not speculate)
            if (k > 0 && plans[order[k - 1]].kind == 2) {
                const size_t site = forward ? b : b + 1;
                lda_ub[b] = dep_ub[order[k - 1]] * local_dims[site] + (forward ? eI[b + 1] : eJ[b]);
                spec_cap = std::max(spec_cap, lda_ub[b] * ind_ub[b]);
            }
        }
    }
    if (need_cap > (size_t)CHAIN_MAX_SET) {
        ++(one ? chain_stats_ext[2] : chain_stats[3]);
        return false;
    }

    // ---- 2. buffers ----
    // With a bounded max_bond_dim every buffer is sized for its final shape the first time: a buffer that grows is released
    // through the process-wide cache, which waits for the whole device (pool.hip) — while ranks grow from scratch that would be
    // a device-wide stall per iteration and handle, and it is what kept concurrent handles from overlapping.
    size_t cap_side = 0;
    {
        size_t dmax = 1;
        for (size_t d : local_dims) dmax = std::max(dmax, d);
        if (chi <= (size_t)CHAIN_MAX_SET && chi * dmax + chi <= 2048) {
            need_cap = std::max(need_cap, chi);
            const size_t side = chi * dmax + chi;
            dep_cap = std::max(dep_cap, side);
            ind_cap = std::max(ind_cap, side);
            Mcap = std::max(Mcap, side);
            Ncap = std::max(Ncap, side);
            steps_cap = std::max(steps_cap, chi);
            spec_cap = std::max(spec_cap, (side * dmax + chi) * side);
            pi_cap = std::max(pi_cap, side * side);
            cap_side = side;
        }
    }
    chain_layout(round_up_sz(need_cap, 64));
    const size_t cap = chain_.cap;
    if (!chain_.weights_valid) {
        chain_.weights.reserve(weights_.size());
        chain_.siteinfo.reserve(2 * n_);
        std::vector<int> si(2 * n_);
        for (size_t p = 0; p < n_; ++p) {
            si[p] = (int)local_dims[p];
            si[n_ + p] = (int)offset_[p];
        }
        T4A_HIP(hipMemcpyAsync(chain_.weights.get(), weights_.data(), weights_.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(chain_.siteinfo.get(), si.data(), si.size() * sizeof(int), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
        chain_.weights_valid = true;
    }
    chain_.ind.reserve(nb * ind_cap * (1 + K));
    chain_.ind_cnt.reserve(nb);
    chain_.dep.reserve(dep_cap * (1 + K));
    chain_.rowmap.reserve(dep_cap);
    chain_.pi.reserve(std::max<size_t>(pi_cap, 64 * 64)); // (at least what the persistent half-sweep needs)
    chain_.factors_stride = 0;
    if (one && chain_.one_factors) {
        size_t stride = 1;
        for (size_t b = 0; b < nb; ++b) stride = std::max(stride, dep_ub[b] * ind_ub[b]);
        if (nb * stride > ((size_t)1 << 27)) { // more than 1 GiB of factored matrices: this sweep runs bond by bond
            ++chain_stats_ext[2];
            return false;
        }
        chain_.factors_stride = stride;
        chain_.factors.reserve(nb * stride);
        chain_.urows.reserve(std::max<size_t>(steps_cap, 1) * std::max(dep_cap, ind_cap));
    }
    if (spec_cap) {
        chain_.spec[0].reserve(spec_cap);
        chain_.spec[1].reserve(spec_cap);
    }
    ChainBlock proto;
    proto.off_piv = 32;
    proto.off_rp = proto.off_piv + sizeof(double) * steps_cap;
    proto.off_cp = proto.off_rp + sizeof(int) * Mcap;
    proto.off_ts = round_up_sz(proto.off_cp + sizeof(int) * Ncap, 8);
    proto.bytes = proto.off_ts + 16;
    chain_.blocks.reserve(nb * proto.bytes);
    chain_.hblocks.reserve(nb * proto.bytes);
    chain_.dims.reserve(nb * 5 + 2); // {M, N, poison, lda} per bond, then one tile counter per bond (speculative evaluation)
    chain_.hdims.reserve(nb * 4);
    const bool timed = eng.prof.enabled;
    const bool timed_events = timed && chain_event_timing;
    while (timed_events && chain_.t0.size() < nb) {
        hipEvent_t a = nullptr, b2 = nullptr;
        T4A_HIP(hipEventCreate(&a));
        T4A_HIP(hipEventCreate(&b2));
        chain_.t0.push_back(a);
        chain_.t1.push_back(b2);
    }

    // ---- 3. tables: current sets, the extras of this iteration, the snapshot for the next one ----
    if (!chain_.tables_valid) chain_upload_current();
    int ext_slot = 0;
    if (use_extras) {
        const uint64_t want = ext->serial;
        if (chain_.snap_serial[0] == want)
            ext_slot = 0;
        else if (chain_.snap_serial[1] == want)
            ext_slot = 1;
        else {
            ext_slot = (in_optimize && chain_.snap_serial[0] == chain_.hist_serial) ? 1 : 0;
            chain_upload_hist(*ext, ext_slot);
            chain_.snap_serial[ext_slot] = want;
        }
    }
    int snap = -1;
    if (in_optimize) { // the sets as they are now are the extras of the next iteration (tensorci2.rs:1675-1689): copied by the first kernel
        snap = use_extras ? 1 - ext_slot : 0;
        chain_.snap_serial[snap] = chain_.hist_serial;
    }

    // ---- 4. the kernel constants of the half-sweep ----
    // (result blocks, dims and tile counters are cleared and the snapshot is taken by chain_indep_kernel: one launch instead
    // of five memset / memcpy operations in front of the chain)
    for (size_t b = 0; b < nb; ++b) {
        std::memset(chain_.hblocks.get() + b * proto.bytes, 0, 32);
        std::memset(chain_.hdims.get() + b * 4, 0xFF, 4 * sizeof(int)); // (-1: not written yet)
    }
    ChainCommon c;
    std::memset(&c, 0, sizeof(c));
    c.I = chain_tab(0);
    c.J = chain_tab(1);
    c.HI = chain_tab(2 + 2 * ext_slot);
    c.HJ = chain_tab(3 + 2 * ext_slot);
    c.mI = chain_mirror(1 - chain_.mcur, 0); // this chain's results go to the other mirror (the current one may still feed a fill)
    c.mJ = chain_mirror(1 - chain_.mcur, 1);
    c.cap = (int)cap;
    c.K = (int)K;
    c.w = chain_.weights.get();
    c.total = (int)total_;
    c.ldim = chain_.siteinfo.get();
    c.woff = chain_.siteinfo.get() + n_;
    c.forward = forward ? 1 : 0;
    c.use_extras = use_extras ? 1 : 0;
    c.one_site = one ? 1 : 0;
    c.ind_code = chain_.ind.get();
    c.ind_acc = chain_.ind.get() + nb * ind_cap;
    c.ind_cnt = chain_.ind_cnt.get();
    c.ind_cap = (int)ind_cap;
    c.dep_code = chain_.dep.get();
    c.dep_acc = chain_.dep.get() + dep_cap;
    c.dep_cap = (int)dep_cap;
    c.rowmap = chain_.rowmap.get();
    c.dims = chain_.dims.get();
    c.hdims = chain_.hdims.get();
    c.n_sites = (int)n_;
    if (snap >= 0) {
        c.snap_dst = chain_tab(2 + 2 * snap).code;
        c.snap_cnt_dst = chain_tab(2 + 2 * snap).cnt;
        c.snap_words = 2 * n_ * cap * (1 + K);
    }
    c.zero_a = reinterpret_cast<uint64_t*>(chain_.blocks.get());
    c.zero_a_words = nb * proto.bytes / 8;
    c.zero_b = reinterpret_cast<uint64_t*>(chain_.dims.get());
    c.zero_b_words = (nb * 5 * sizeof(int) + 7) / 8;

    chain_.common = c;
    chain_.prepared = true;
    chain_.forward = forward;
    chain_.in_optimize = in_optimize;
    chain_.ext_idx = ext_idx;
    chain_.chi = chi;
    chain_.tol = options.tolerance;
    chain_.order = std::move(order);
    chain_.plans = std::move(plans);
    chain_.dep_ub = std::move(dep_ub);
    chain_.ind_ub = std::move(ind_ub);
    chain_.use_extras = use_extras;
    chain_.ind_cap = ind_cap;
    chain_.cap_side = cap_side;
    chain_.proto = proto;
    chain_.timed = timed;
    chain_.timed_events = timed_events;
    chain_.tokens.assign(nb, 0u);
    if (launch) chain_launch();
    return true;
}

// the launches of a prepared half-sweep on this handle's own stream and XCD
void Tci2::chain_launch()
{
    if (!chain_.prepared) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: nothing prepared to launch");
    chain_.prepared = false;
    const size_t nb = n_ - 1;
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const bool forward = chain_.forward, use_extras = chain_.use_extras, timed_events = chain_.timed_events;
    const size_t chi = chain_.chi, ind_cap = chain_.ind_cap, cap_side = chain_.cap_side;
    const ChainCommon& c = chain_.common;
    const ChainBlock& proto = chain_.proto;
    const std::vector<size_t>& order = chain_.order;
    const std::vector<ChainRrluPlan>& plans = chain_.plans;
    const std::vector<size_t>& dep_ub = chain_.dep_ub;
    const std::vector<size_t>& ind_ub = chain_.ind_ub;
    std::vector<unsigned>& tokens = chain_.tokens;
    hipStream_t st = eng.stream();
    // result blocks reach the host in ONE copy behind the chain (T4A_CHAIN_HOST_MIRROR=1: every rrLU kernel mirrors its own block
    // into pinned memory as it ends — stores over PCIe that the kernel's end has to wait for, once per bond)
    static const bool per_launch_mirror = diag_env("T4A_CHAIN_HOST_MIRROR") != nullptr;
    auto block_of = [&](size_t b) {
        ChainBlock k = proto;
        k.dev = chain_.blocks.get() + b * proto.bytes;
        k.host = per_launch_mirror ? chain_.hblocks.get() + b * proto.bytes : nullptr;
        return k;
    };
    {
        size_t reserve_words = 0; // the largest shape this handle can reach: the mailbox is sized once
        ChainRrluPlan cp;
        if (cap_side && eng.chain_plan((int)cap_side, (int)cap_side, &cp) && cp.kind == 2)
            reserve_words = (rrlu_xcd_keys_bytes(cp.xcd) + rrlu_xcd_cols_bytes(cp.xcd, (int)cap_side)) / sizeof(unsigned long long);
        eng.chain_begin(plans, reserve_words);
    }
    bool counted = false;
    try {
        chain_indep_launch(c, (int)nb, st);
        unsigned* tile_counters = reinterpret_cast<unsigned*>(chain_.dims.get() + nb * 4);
        static const bool no_spec = diag_env("T4A_CHAIN_NO_SPEC") != nullptr;
        const bool solo = g_chains_inflight.fetch_add(1) == 0 && !no_spec;
        counted = true;
        // every matrix of the half-sweep fits the one-wave kernel: ONE persistent workgroup walks the bonds (kernels_chain.hip)
        static const bool no_walk = std::getenv("T4A_NO_WALK") != nullptr;
        size_t walk_cols = 0;
        bool walk = !no_walk && !timed_events && !per_launch_mirror && nb <= 128; // (128: WALK_MAX_BONDS, kernels_chain.hip)
        for (size_t b = 0; walk && b < nb; ++b) {
            walk = dep_ub[b] <= 64 && ind_ub[b] <= 32; // (wider: the launched chain's one-workgroup kernel beats one wave)
            walk_cols = std::max(walk_cols, ind_ub[b]);
        }
        if (walk) {
            ChainWalkArgs w;
            std::memset(&w, 0, sizeof(w));
            w.blocks = chain_.blocks.get();
            w.block_bytes = proto.bytes;
            w.off_piv = proto.off_piv;
            w.off_rp = proto.off_rp;
            w.off_cp = proto.off_cp;
            w.off_ts = proto.off_ts;
            w.pi = chain_.pi.get();
            w.n_bonds = (int)nb;
            w.max_steps = (int)std::min<size_t>(chi, 64);
            w.rel_tol = chain_.tol;
            w.abs_tol = chain_.one_site ? chain_.abs_tol : 0.0;
            w.factors = chain_.factors_stride ? chain_.factors.get() : nullptr;
            w.factors_stride = chain_.factors_stride;
            if (chain_.walk_token > 0xFFFF0000u || chain_.walk_token == 0u) chain_.walk_token = 1u;
            w.token_base = chain_.walk_token;
            chain_.walk_token += (unsigned)nb;
            w.timed = chain_.timed ? 1 : 0;
            static const bool old_prep = diag_env("T4A_WALK_OLD_PREP") != nullptr;
            w.lean_prep = old_prep ? 0 : 1;
            static const bool walk_dbg = std::getenv("T4A_WALK_DEBUG") != nullptr;
            if (walk_dbg) {
                chain_.walk_dbg.reserve(8);
                T4A_HIP(hipMemsetAsync(chain_.walk_dbg.get(), 0, 8 * sizeof(unsigned long long), st));
                w.phase_ticks = chain_.walk_dbg.get();
            }
            chain_.walked = true;
            for (size_t k = 0; k < nb; ++k) tokens[order[k]] = w.token_base + (unsigned)k;
            chain_walk_launch(c, fn_dev_, w, (int)walk_cols, st);
            ++chain_stats_ext[0];
        }
        // T4A_CHAIN_DEFER_MIRROR=1: the launched chain's preparations leave the pinned mirrors alone and one bulk copy follows the
        // chain (measured: 21.44 against 21.39 ms per cfg3 sweep — the stores over PCIe are not what a preparation's 8.6 us consist
        // of, and the extra launch costs what they cost; the persistent half-sweep always copies in bulk)
        static const bool defer_mirror = diag_env("T4A_CHAIN_DEFER_MIRROR") != nullptr;
        static const bool want_prep_dbg = std::getenv("T4A_PREP_DEBUG") != nullptr; // phase times of the preparation kernels of this chain
        unsigned long long* prep_dbg = nullptr;
        if (want_prep_dbg && !walk) {
            chain_.walk_dbg.reserve(8);
            T4A_HIP(hipMemsetAsync(chain_.walk_dbg.get(), 0, 8 * sizeof(unsigned long long), st));
            prep_dbg = chain_.walk_dbg.get();
            chain_.prep_dbg = true;
        }
        bool spec_pending = false; // the previous bond's launch evaluates this bond's candidate matrix
        for (size_t k = 0; !walk && k < nb; ++k) {
            const size_t b = order[k];
            const ChainRrluPlan& pl = plans[b];
            const bool fused = pl.kind == 1 && pl.fused;
            const bool spec_here = spec_pending && !fused;
            const ChainBlock blk = block_of(b);
            ChainPrepArgs pa;
            std::memset(&pa, 0, sizeof(pa));
            pa.b = (int)b;
            pa.do_build = 1;
            pa.dbg = prep_dbg;
            pa.defer_host_writes = defer_mirror ? 1 : 0;
            pa.with_rowmap = spec_here ? 1 : 0;
            pa.prev_b = -1;
            if (k > 0) {
                const size_t pb = order[k - 1];
                const ChainBlock pblk = block_of(pb);
                pa.prev_b = (int)pb;
                pa.prev_iresult = reinterpret_cast<const int*>(pblk.dev + 16);
                pa.prev_rowperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_rp);
                pa.prev_colperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_cp);
                pa.prev_token = tokens[pb];
            }
            chain_prep_launch(c, pa, st);
            // nobody speculated on this bond (first bond, or the previous launch was a single workgroup): evaluate it now
            if (!fused && !spec_here) chain_pi_launch(c, fn_dev_, (int)b, (int)dep_ub[b], (int)ind_ub[b], chain_.pi.get(), st);
            // the candidate matrix of the NEXT bond rides on this bond's launch when that is a single-XCD launch
            XcdSpecArgs sp;
            std::memset(&sp, 0, sizeof(sp));
            spec_pending = false;
            if (solo && k + 1 < nb && pl.kind == 2) {
                const size_t nx = order[k + 1];
                if (!(plans[nx].kind == 1 && plans[nx].fused)) {
                    const size_t site = forward ? nx : nx + 1;
                    const ChainTab& H = forward ? c.HI : c.HJ;
                    const size_t hsite = forward ? nx + 1 : nx;
                    sp.out = chain_.spec[(k + 1) & 1].get();
                    sp.dep_acc = c.dep_acc;
                    sp.w_site = c.w + offset_[site];
                    sp.ext_acc = H.acc + hsite * cap * K;
                    sp.ext_cnt = use_extras ? H.cnt + hsite : nullptr;
                    sp.ind_acc = c.ind_acc + nx * ind_cap * K;
                    sp.ind_cnt = c.ind_cnt + nx;
                    sp.tile_counter = tile_counters + nx;
                    sp.total = (int)total_;
                    sp.d = (int)local_dims[site];
                    sp.fn = fn_dev_;
                    spec_pending = true;
                }
            }
            FusedPi fp;
            fp.fn = fn_dev_;
            fp.d_rowacc = c.dep_acc;
            fp.d_colacc = c.ind_acc + b * ind_cap * K;
            fp.host_resident = false;
            const double* A = fused ? nullptr : (spec_here ? chain_.spec[k & 1].get() : chain_.pi.get());
            if (timed_events) T4A_HIP(hipEventRecord(chain_.t0[b], st));
            tokens[b] = eng.chain_rrlu(pl, forward, A, spec_here ? c.rowmap : nullptr, fused ? &fp : nullptr, c.dims + b * 4, chi, chain_.tol,
                                       chain_.one_site ? chain_.abs_tol : 0.0, blk, spec_pending ? &sp : nullptr,
                                       chain_.factors_stride ? chain_.factors.get() + b * chain_.factors_stride : nullptr, chain_.urows.get());
            if (timed_events) T4A_HIP(hipEventRecord(chain_.t1[b], st));
        }
        if (!walk) { // the pivots of the last bond
            const size_t pb = order[nb - 1];
            const ChainBlock pblk = block_of(pb);
            ChainPrepArgs pa;
            std::memset(&pa, 0, sizeof(pa));
            pa.do_build = 0;
            pa.defer_host_writes = defer_mirror ? 1 : 0;
            pa.prev_b = (int)pb;
            pa.prev_iresult = reinterpret_cast<const int*>(pblk.dev + 16);
            pa.prev_rowperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_rp);
            pa.prev_colperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_cp);
            pa.prev_token = tokens[pb];
            chain_prep_launch(c, pa, st);
            if (defer_mirror) chain_mirror_launch(c, (int)nb, st);
        }
        chain_.cores_batched = false;
        static const bool no_batched_cores = diag_env("T4A_NO_BATCHED_CORES") != nullptr;
        if (chain_.one_site && chain_.factors_stride && forward && !no_batched_cores && nb <= (size_t)LUCI_LEFT_CORES_MAX_JOBS) {
            // the site tensors of every bond of rank <= 16 in one launch behind the chain (kernels_dense.hip): shapes, ranks and
            // permutations are read where the chain left them; the buffers are sized for the upper bounds
            LeftCoreJobs jobs;
            std::memset(&jobs, 0, sizeof(jobs));
            size_t max_rows = 1;
            for (size_t b = 0; b < nb; ++b) {
                const ChainBlock blk = block_of(b);
                const size_t rub = std::max<size_t>(std::min({dep_ub[b], ind_ub[b], chi}), 1);
                DevCore& core = cores[b];
                core.buf.reserve(std::max<size_t>(dep_ub[b] * rub, 1));
                LeftCoreJob& j = jobs.j[b];
                j.lu = chain_.factors.get() + b * chain_.factors_stride;
                j.dims = c.dims + b * 4;
                j.iresult = reinterpret_cast<const int*>(blk.dev + 16);
                j.row_perm = reinterpret_cast<const int*>(blk.dev + blk.off_rp);
                j.core = core.buf.get();
                j.S = (int)local_dims[b];
                j.token = tokens[b];
                max_rows = std::max(max_rows, dep_ub[b]);
            }
            luci_left_cores_batched_launch(jobs, (int)nb, (int)max_rows, st);
            chain_.cores_batched = true;
        }
        chain_.last_core_launched = false;
        static const bool no_last_core = diag_env("T4A_NO_CHAIN_LAST_CORE") != nullptr;
        if (chain_.one_site && chain_.factors_stride && forward && !no_last_core) {
            // ... and the last site's tensor (tensorci2.rs:902-912): I_{n-1} is final behind the last bond; evaluated from the tables, no
            // host round trip (accumulators built on the host, staged, evaluated, packed, synchronised: ~50 us of a small solve)
            const size_t last = n_ - 1;
            const size_t a_ub = std::max<size_t>(std::min({dep_ub[nb - 1], ind_ub[nb - 1], chi}), 1);
            const size_t entries = a_ub * local_dims[last] * std::max<size_t>(j_set[last].count, 1);
            if (entries <= ((size_t)1 << 24)) {
                cores[last].buf.reserve(entries);
                chain_last_core_launch(c, fn_dev_, cores[last].buf.get(), (int)entries, st);
                chain_.last_core_launched = true;
            }
        }
        if (!per_launch_mirror)
            T4A_HIP(hipMemcpyAsync(chain_.hblocks.get(), chain_.blocks.get(), nb * proto.bytes, hipMemcpyDeviceToHost, st));
        T4A_HIP(hipGetLastError());
    } catch (...) {
        (void)hipStreamSynchronize(st);
        eng.chain_end();
        if (counted) g_chains_inflight.fetch_sub(1);
        chain_.tables_valid = false;
        chain_.snap_serial[0] = chain_.snap_serial[1] = ~0ull; // (the slot claimed for this iteration was never, or only partly, written)
        throw;
    }
    chain_.inflight = true;
    chain_.wait_stream = st;
    chain_.group_role = 0;
}

// The prepared half-sweeps of several handles as one chain of launches.  Per bond: one preparation kernel (a workgroup per
// handle), one candidate-matrix kernel, one rrLU launch in which the workgroups of XCD i factorise handle i's matrix — three
// launches for the whole group instead of two to three per handle, and the host enqueues a group's half-sweep in the time
// it took for one handle.  No speculation (there are no idle XCDs to do it) and no single-workgroup plans (a small bond costs
// a group one launch, i.e. an eighth of it per handle).  The handles' results are those of their own chains: same lists, same
// kernels, same order.
void Tci2::chain_group_launch(const std::vector<Tci2*>& hs)
{
    static const bool no_group = diag_env("T4A_CHAIN_NO_GROUP") != nullptr;
    const size_t nh = hs.size();
    if (nh == 0) return;
    bool ok = nh >= 2 && nh <= (size_t)CHAIN_GROUP_MAX && !no_group;
    Tci2* lead = hs[0];
    for (Tci2* h : hs) {
        if (!h->chain_.prepared) throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: a handle of the group has nothing prepared");
        ok = ok && h->n_ == lead->n_ && h->chain_.forward == lead->chain_.forward && h->chain_.chi == lead->chain_.chi &&
             h->chain_.tol == lead->chain_.tol && !h->chain_.timed_events;
    }
    const size_t nb = lead->n_ - 1;
    const bool forward = lead->chain_.forward;
    std::vector<ChainRrluPlan> gplans(nb);
    size_t side_ub = 0;
    if (ok) {
        for (size_t b = 0; b < nb && ok; ++b) {
            size_t dm = 1, im = 1;
            for (Tci2* h : hs) {
                dm = std::max(dm, h->chain_.dep_ub[b]);
                im = std::max(im, h->chain_.ind_ub[b]);
            }
            ok = Engine::chain_group_plan((int)dm, (int)im, &gplans[b]);
            gplans[b].code += forward ? 0 : 4;
        }
        for (Tci2* h : hs) side_ub = std::max(side_ub, h->chain_.cap_side);
    }
    if (!ok) {
        for (Tci2* h : hs) h->chain_launch();
        return;
    }
    hipStream_t st = lead->eng.stream();
    size_t reserve_words = 0;
    {
        ChainRrluPlan cp;
        if (side_ub && Engine::chain_group_plan((int)side_ub, (int)side_ub, &cp))
            reserve_words = (rrlu_xcd_keys_bytes(cp.xcd) + rrlu_xcd_cols_bytes(cp.xcd, (int)side_ub)) / sizeof(unsigned long long);
    }
    std::vector<char> counted(nh, 0);
    bool locked = false;
    try {
        lead->chain_.gslots.reserve(CHAIN_GROUP_MAX);
        lead->chain_.hgslots.reserve(CHAIN_GROUP_MAX);
        for (size_t i = 0; i < nh; ++i) {
            Tci2* h = hs[i];
            h->chain_.prepared = false;
            if (h != lead) { // the group's stream continues behind whatever this handle's own stream still holds
                if (!h->chain_.group_ev) T4A_HIP(hipEventCreateWithFlags(&h->chain_.group_ev, hipEventDisableTiming));
                T4A_HIP(hipEventRecord(h->chain_.group_ev, h->eng.stream()));
                T4A_HIP(hipStreamWaitEvent(st, h->chain_.group_ev, 0));
            }
            h->eng.chain_group_reserve(gplans, reserve_words, st);
            ChainGroupSlot& slot = lead->chain_.hgslots.get()[i];
            slot.c = h->chain_.common;
            slot.fn = h->fn_dev_;
            slot.pi = h->chain_.pi.get();
            g_chains_inflight.fetch_add(1);
            counted[i] = 1;
        }
        T4A_HIP(hipMemcpyAsync(lead->chain_.gslots.get(), lead->chain_.hgslots.get(), nh * sizeof(ChainGroupSlot), hipMemcpyHostToDevice, st));
        const ChainGroupSlot* d_slots = lead->chain_.gslots.get();
        lead->eng.chain_group_lock();
        locked = true;
        static const bool per_launch_mirror = diag_env("T4A_CHAIN_HOST_MIRROR") != nullptr;
        auto block_of = [&](Tci2* h, size_t b) {
            ChainBlock k = h->chain_.proto;
            k.dev = h->chain_.blocks.get() + b * k.bytes;
            k.host = per_launch_mirror ? h->chain_.hblocks.get() + b * k.bytes : nullptr;
            return k;
        };
        auto prev_of = [&](Tci2* h, size_t pb, ChainPrepArgs& pa) {
            const ChainBlock pblk = block_of(h, pb);
            pa.prev_b = (int)pb;
            pa.prev_iresult = reinterpret_cast<const int*>(pblk.dev + 16);
            pa.prev_rowperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_rp);
            pa.prev_colperm = reinterpret_cast<const int*>(pblk.dev + pblk.off_cp);
            pa.prev_token = h->chain_.tokens[pb];
        };
        chain_indep_group_launch(d_slots, (int)nh, (int)nb, st);
        const std::vector<size_t>& order = lead->chain_.order;
        for (size_t k = 0; k < nb; ++k) {
            const size_t b = order[k];
            ChainPrepGroupArgs pg;
            std::memset(&pg, 0, sizeof(pg));
            for (size_t i = 0; i < nh; ++i) {
                ChainPrepArgs& pa = pg.a[i];
                pa.b = (int)b;
                pa.do_build = 1;
                pa.prev_b = -1;
                if (k > 0) prev_of(hs[i], order[k - 1], pa);
            }
            chain_prep_group_launch(d_slots, pg, (int)nh, st);
            chain_pi_group_launch(d_slots, (int)nh, (int)b, gplans[b].kM, gplans[b].kN, st);
            RrluXcdGroupArgs ga;
            std::memset(&ga, 0, sizeof(ga));
            for (int x = 0; x < 8; ++x) ga.p[x].xcc = -1; // (an empty slot: its workgroups return at once)
            for (size_t i = 0; i < nh; ++i) {
                Tci2* h = hs[i];
                h->chain_.tokens[b] = h->eng.chain_group_args(gplans[b], forward, h->chain_.pi.get(), h->chain_.common.dims + b * 4, h->chain_.chi,
                                                              h->chain_.tol, 0.0, block_of(h, b), (int)i, &ga.p[i], st);
            }
            rrlu_xcd_group_launch_v(xcd_version(), gplans[b].xcd, ga, !forward, st);
        }
        { // the pivots of the last bond
            ChainPrepGroupArgs pg;
            std::memset(&pg, 0, sizeof(pg));
            for (size_t i = 0; i < nh; ++i) prev_of(hs[i], order[nb - 1], pg.a[i]);
            chain_prep_group_launch(d_slots, pg, (int)nh, st);
        }
        if (!per_launch_mirror)
            for (Tci2* h : hs)
                T4A_HIP(hipMemcpyAsync(h->chain_.hblocks.get(), h->chain_.blocks.get(), nb * h->chain_.proto.bytes, hipMemcpyDeviceToHost, st));
        T4A_HIP(hipGetLastError());
    } catch (...) {
        (void)hipStreamSynchronize(st);
        if (locked) lead->eng.chain_end();
        for (size_t i = 0; i < nh; ++i) {
            if (counted[i]) g_chains_inflight.fetch_sub(1);
            hs[i]->chain_.prepared = false;
            hs[i]->chain_.tables_valid = false;
            hs[i]->chain_.snap_serial[0] = hs[i]->chain_.snap_serial[1] = ~0ull;
        }
        throw;
    }
    for (Tci2* h : hs) {
        h->chain_.plans = gplans;
        h->chain_.inflight = true;
        h->chain_.wait_stream = st;
        h->chain_.group_role = h == lead ? 1 : 2;
    }
}

// An exception left optimize() / optimize_group() between the launch of a chain and its chain_finish (a fill of the previous
// iteration reporting a singular pivot matrix, a runtime error while issuing it): wait for the chain, give the XCD (or the
// chip) back and forget its results — the host's sets are those from before the chain (it wrote the OTHER mirror), the device
// tables are re-uploaded by the next chain.  Without this the handle kept its reservation and every later call on it failed
// with "a chain is already in flight" while other handles waited for the XCD forever.
void Tci2::chain_abort() noexcept
{
    if (chain_.prepared) {
        // chain_enqueue claimed a snapshot slot for this iteration before anything was launched: a chain that
        // never ran — or only partly — must not leave a slot behind that the next optimize() takes for the history's extras
        chain_.snap_serial[0] = chain_.snap_serial[1] = ~0ull;
        chain_.tables_valid = false;
    }
    chain_.prepared = false;
    chain_.walked = false;
    if (!chain_.inflight) return;
    chain_.inflight = false;
    hipStream_t st = chain_.wait_stream ? chain_.wait_stream : eng.stream();
    (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    if (chain_.group_role != 2) eng.chain_end();
    chain_.group_role = 0;
    chain_.wait_stream = nullptr;
    g_chains_inflight.fetch_sub(1);
    chain_.tables_valid = false;
    chain_.snap_serial[0] = chain_.snap_serial[1] = ~0ull; // (the snapshot taken by the aborted chain is not the history's newest entry any more)
}

void Tci2::chain_finish(const TCI2Options& options)
{
    if (!chain_.inflight) return;
    chain_.inflight = false;
    hipStream_t st = chain_.wait_stream ? chain_.wait_stream : eng.stream();
    const size_t nb = n_ - 1;
    const size_t K = (size_t)chain_.n_acc, cap = chain_.cap;
    const bool forward = chain_.forward;
    const ChainBlock& proto = chain_.proto;
    const auto wait_t0 = std::chrono::steady_clock::now();
    const hipError_t sync_err = hipStreamSynchronize(st);
    g_chain_wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - wait_t0).count();
    if (chain_.group_role != 2) eng.chain_end(); // (a member of a group chain holds nothing: the leader reserved the chip)
    const bool was_group = chain_.group_role != 0;
    chain_.group_role = 0;
    chain_.wait_stream = nullptr;
    g_chains_inflight.fetch_sub(1);
    if (sync_err != hipSuccess) {
        chain_.tables_valid = false;
        T4A_HIP(sync_err);
    }
    if (chain_.walked && chain_.walk_dbg.get() && std::getenv("T4A_WALK_DEBUG")) {
        unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpy(t, chain_.walk_dbg.get(), sizeof(t), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[t4a walk] %zu bonds: preparation %.1f us (gather %.1f, dependent list %.1f), candidate matrix %.1f us, rrLU %.1f us, kernel %.1f us\n", nb,
                     t[0] * 0.01, t[4] * 0.01, t[5] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01);
    }
    chain_.walked = false;
    if (chain_.prep_dbg && chain_.walk_dbg.get()) {
        unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpy(t, chain_.walk_dbg.get(), sizeof(t), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[t4a prep] %zu bonds: preparation kernels %.1f us in total: gather %.1f, dependent list %.1f, rest %.1f\n", nb, t[6] * 0.01, t[4] * 0.01,
                     t[5] * 0.01, (double)(t[6] - t[4] - t[5]) * 0.01);
    }
    chain_.prep_dbg = false;
    const int mnew = 1 - chain_.mcur;
    const ChainTab ni = chain_mirror(mnew, 0), nj = chain_mirror(mnew, 1);
    long failed_k = -1;
    bool failed_timeout = false;
    for (size_t k = 0; k < nb; ++k) {
        const size_t b = chain_.order[k];
        const int* hd = chain_.hdims.get() + b * 4;
        const char* hb = chain_.hblocks.get() + b * proto.bytes;
        const int* hi = reinterpret_cast<const int*>(hb + 16);
        if (hd[2] != 0 || hd[0] <= 0 || hd[1] <= 0) { // poisoned by its preparation: the previous bond's pivots were not gathered either
            failed_k = k > 0 ? (long)k - 1 : 0;
            break;
        }
        if (hi[1] != 0 || hi[3] != (int)chain_.tokens[b]) {
            failed_k = (long)k;
            failed_timeout = hi[1] != 2; // (2: the second-generation single-XCD kernel met non-finite values: the per-bond path handles them, nothing is wrong with the placement)
            break;
        }
        if (hi[2] != 0) { // NaN in L or U (matrixlu.rs:614-668): the per-bond path reports it before it touches the sets of this bond
            failed_k = (long)k;
            break;
        }
    }
    if (failed_k >= 0) {
        static const bool dbg = std::getenv("T4A_CHAIN_DEBUG") != nullptr;
        if (dbg) {
            const size_t b = chain_.order[(size_t)failed_k];
            const int* hd = chain_.hdims.get() + b * 4;
            const int* hi = reinterpret_cast<const int*>(chain_.hblocks.get() + b * proto.bytes + 16);
            const ChainRrluPlan& pl = chain_.plans[b];
            std::fprintf(stderr, "[t4a chain] bond %zu (k = %ld of %zu, %s) did not complete: dims %d x %d poisoned %d lda %d | npiv %d flag %d nan %d token %d (want %u) | plan kind %d %d x %d wg %d RPT %d CPT %d W %d\n",
                         b, failed_k, nb, forward ? "forward" : "backward", hd[0], hd[1], hd[2], hd[3], hi[0], hi[1], hi[2], hi[3], chain_.tokens[b], pl.kind, pl.kM, pl.kN,
                         pl.xcd.wg, pl.xcd.RPT, pl.xcd.CPT, pl.xcd.W);
        }
    }
    const size_t done = failed_k < 0 ? nb : (size_t)failed_k;
    for (size_t k = 0; k < done; ++k) {
        const size_t b = chain_.order[k];
        const int* hd = chain_.hdims.get() + b * 4;
        const char* hb = chain_.hblocks.get() + b * proto.bytes;
        const int* hi = reinterpret_cast<const int*>(hb + 16);
        const size_t M = (size_t)hd[0], N = (size_t)hd[1];
        const int rank = hi[0];
        double last_error, abs_max;
        std::memcpy(&last_error, hb, sizeof(double));
        std::memcpy(&abs_max, hb + 8, sizeof(double));
        const size_t cnt = (size_t)(rank > 0 ? rank : 1);
        if ((size_t)ni.cnt[b + 1] != cnt || (size_t)nj.cnt[b] != cnt)
            throw Error(T4A_GPU_INTERNAL_ERROR, "bond chain: gathered pivot count of bond " + std::to_string(b) + " disagrees with its rank");
        i_set[b + 1].count = cnt; // (digit tables follow on demand: sync_digits)
        j_set[b].count = cnt;
        i_set[b + 1].d.clear();
        j_set[b].d.clear();
        {   // work model of the factorisation, as Engine::luci counts it
            double bytes = 8.0 * (double)M * (double)N, flops = 0.0;
            for (int q = 0; q < rank; ++q) {
                const double mr = (double)((long)M - q - 1), nr = (double)((long)N - q - 1);
                bytes += 16.0 * mr * nr;
                flops += 2.0 * mr * nr + mr;
            }
            eng.prof.v[8] += rank;
            eng.prof.v[9] += bytes;
            eng.prof.v[10] += flops;
            eng.prof.v[11] += (double)M * (double)N;
            if (chain_.timed) { // device-side time stamps of the launch (wall_clock64: 100 MHz)
                unsigned long long ts[2];
                std::memcpy(ts, hb + proto.off_ts, sizeof(ts));
                float ms = ts[1] > ts[0] ? (float)((double)(ts[1] - ts[0]) * 1e-5) : 0.f;
                if (chain_.timed_events) T4A_HIP(hipEventElapsedTime(&ms, chain_.t0[b], chain_.t1[b]));
                eng.prof.v[0] += ms;
                eng.prof.v[1] += 1.0;
                auto& vs = eng.variant_stats_[chain_.plans[b].code];
                vs[0] += ms;
                vs[1] += 1.0;
                vs[2] += bytes;
                vs[3] += rank;
                if (chain_.chi > 0 && (size_t)rank == chain_.chi) { // sub-aggregate: the launches that ran all max_bond_dim pivot steps
                    auto& sat = eng.variant_stats_[chain_.plans[b].code + 10000000];
                    sat[0] += ms;
                    sat[1] += 1.0;
                    sat[2] += bytes;
                    sat[3] += rank;
                }
            }
        }
        if (abs_max > max_sample_value) max_sample_value = abs_max; // update_max_sample_value over Π (tensorci2.rs:2009-2014)
        bond_errors[b] = last_error; // = pivot_errors.back() (tensorci2.rs:2002-2004, :1040-1046)
        if (chain_.one_site) { // sweep1site_at_bond's tail (tensorci2.rs:1040-1049): the pivot errors of the bond join the running maxima
            std::vector<double> pe((size_t)rank + 1);
            const double* pv = reinterpret_cast<const double*>(hb + proto.off_piv);
            for (int q = 0; q < rank; ++q) pe[(size_t)q] = std::sqrt(pv[q] * pv[q]);
            pe[(size_t)rank] = last_error;
            update_pivot_errors(pe);
        } else {
            last_sweep_shapes[b] = {M, N, (size_t)rank};
        }
    }
    if (done > 0) { // the new mirror becomes the host's copy of the sets; what this chain did not write comes from the old one
        const ChainTab oi = chain_mirror(chain_.mcur, 0), oj = chain_mirror(chain_.mcur, 1);
        std::vector<char> wi(n_, 0), wj(n_, 0);
        for (size_t k = 0; k < done; ++k) {
            wi[chain_.order[k] + 1] = 1;
            wj[chain_.order[k]] = 1;
        }
        for (size_t p = 0; p < n_; ++p) {
            if (!wi[p]) {
                std::memcpy(ni.code + p * cap, oi.code + p * cap, i_set[p].count * sizeof(uint64_t));
                std::memcpy(ni.acc + p * cap * K, oi.acc + p * cap * K, i_set[p].count * K * sizeof(uint64_t));
                ni.cnt[p] = oi.cnt[p];
            }
            if (!wj[p]) {
                std::memcpy(nj.code + p * cap, oj.code + p * cap, j_set[p].count * sizeof(uint64_t));
                std::memcpy(nj.acc + p * cap * K, oj.acc + p * cap * K, j_set[p].count * K * sizeof(uint64_t));
                nj.cnt[p] = oj.cnt[p];
            }
        }
        chain_.mcur = mnew;
        chain_.digits_stale = true;
    }
    if (chain_.one_site && chain_.factors_stride) {
        // update_tensors: the LUCI factor of every completed bond becomes its site tensor (tensorci2.rs:1020-1038) — the factored
        // matrices waited in their per-bond buffers, the permutations in the result blocks
        for (size_t k = 0; k < done; ++k) {
            const size_t b = chain_.order[k];
            const int* hd = chain_.hdims.get() + b * 4;
            const int rank = reinterpret_cast<const int*>(chain_.hblocks.get() + b * proto.bytes + 16)[0];
            char* dblk = chain_.blocks.get() + b * proto.bytes;
            LuciResult lu;
            lu.M = hd[0];
            lu.N = hd[1];
            lu.rank = rank;
            if (chain_.cores_batched && rank <= LUCI_LEFT_CORES_MAX_RANK) { // already written behind the chain: only the shape is missing
                DevCore& core = cores[b];
                core.l = (b == 0) ? 1 : i_set[b].count;
                core.s = local_dims[b];
                core.r = std::max<size_t>((size_t)rank, 1);
                continue;
            }
            eng.build_factors_from(chain_.factors.get() + b * chain_.factors_stride, reinterpret_cast<const int*>(dblk + proto.off_rp),
                                   reinterpret_cast<const int*>(dblk + proto.off_cp), lu.M, lu.N, rank, forward);
            if (forward) {
                const size_t left_dim = (b == 0) ? 1 : i_set[b].count;
                set_core_from_left(b, left_dim, local_dims[b], lu);
            } else {
                const size_t site = b + 1;
                const size_t right_dim = (site == n_ - 1) ? 1 : j_set[site].count;
                set_core_from_right(site, local_dims[site], right_dim, lu);
            }
        }
    }
    chain_.last_core_ok = chain_.last_core_launched && failed_k < 0; // (a chain that stopped early left I_{n-1} unfinished: the host path evaluates the last site)
    chain_.last_core_launched = false;
    if (failed_k < 0) {
        if (chain_.one_site) {
            ++chain_stats_ext[1];
        } else {
            ++chain_stats[0];
            if (was_group) ++chain_stats[4];
            chain_stats[1] += nb;
        }
        if (chain_verify) { // tests: the mirror decodes to index sets whose codes / accumulators are the mirror's, and equals the device tables
            sync_digits();
            std::vector<uint64_t> a;
            for (size_t p = 0; p < n_; ++p)
                for (int side = 0; side < 2; ++side) {
                    const IndexSet& s = side == 0 ? i_set[p] : j_set[p];
                    const ChainTab& m = side == 0 ? ni : nj;
                    const size_t first = side == 0 ? 0 : p + 1;
                    accumulate(s, first, a);
                    for (size_t k = 0; k < s.count; ++k) {
                        bool same = m.code[p * cap + k] == code_of(s.at(k), first, s.width, side == 0);
                        for (size_t q = 0; same && q < K; ++q) same = m.acc[(p * cap + k) * K + q] == a[k * K + q];
                        if (!same)
                            throw Error(T4A_GPU_INTERNAL_ERROR, std::string("bond chain: mirror entry of ") + (side == 0 ? "I_" : "J_") +
                                                                    std::to_string(p) + " does not decode / accumulate consistently");
                    }
                }
            const size_t fam = n_ * cap * (1 + K);
            std::vector<uint64_t> dev(2 * fam);
            std::vector<int> dcnt(2 * n_);
            T4A_HIP(hipMemcpy(dev.data(), chain_tab(0).code, dev.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            T4A_HIP(hipMemcpy(dcnt.data(), chain_tab(0).cnt, dcnt.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int side = 0; side < 2; ++side)
                for (size_t p = 0; p < n_; ++p) {
                    const ChainTab& m = side == 0 ? ni : nj;
                    bool same = dcnt[(size_t)side * n_ + p] == m.cnt[p];
                    for (size_t k = 0; same && k < (size_t)m.cnt[p]; ++k) {
                        same = dev[(size_t)side * fam + p * cap + k] == m.code[p * cap + k];
                        for (size_t q = 0; same && q < K; ++q)
                            same = dev[(size_t)side * fam + n_ * cap + (p * cap + k) * K + q] == m.acc[(p * cap + k) * K + q];
                    }
                    if (!same)
                        throw Error(T4A_GPU_INTERNAL_ERROR, std::string("bond chain: device table ") + (side == 0 ? "I_" : "J_") + std::to_string(p) +
                                                                " differs from its host mirror");
                }
        }
        return;
    }
    // a bond did not complete (bounded spin gave up: the placement assumption of the single-XCD kernel failed or another
    // process holds its compute units; a capacity bound was hit; NaN in the factors): the rest of the half-sweep runs bond by
    // bond (a NaN is met again there and reported as NaNEncountered like the reference does)
    ++(chain_.one_site ? chain_stats_ext[3] : chain_stats[2]);
    sync_digits();
    chain_.tables_valid = false;
    // (only the kernels that ASSUME a placement: the one-workgroup / one-wave plans are booked as kind 2 as well but elect nobody — ADVICE round 4)
    if (failed_timeout && chain_.plans[chain_.order[(size_t)failed_k]].kind == 2 && chain_.plans[chain_.order[(size_t)failed_k]].xcd.wg == 0) xcd_disable();
    prep_.valid = false;
    prefetch_.wanted = false;
    prefetch_.fill_site = -1;
    prefetch_.flush_fill = false;
    invalidate_fill_cache();
    std::vector<IndexSet> no_i(n_), no_j(n_);
    for (size_t p = 0; p < n_; ++p) {
        no_i[p].width = p;
        no_j[p].width = n_ - p - 1;
    }
    const std::vector<IndexSet>* xi = &no_i;
    const std::vector<IndexSet>* xj = &no_j;
    if (chain_.ext_idx >= 0) {
        HistEntry& e = history[(size_t)chain_.ext_idx];
        hist_digits(e);
        xi = &e.is;
        xj = &e.js;
    }
    for (size_t k = (size_t)failed_k; k < nb; ++k) {
        const size_t b = chain_.order[k];
        if (chain_.one_site)
            sweep1site_at_bond(forward ? b : b + 1, forward, options.tolerance, chain_.abs_tol, options.max_bond_dim_or_max(), chain_.one_factors);
        else
            update_pivots(b, forward, options, (*xi)[b + 1], (*xj)[b]);
    }
}

} // namespace t4a
