// kernels_chain.hip — device side of the host-free half-sweep ("bond chain").
//
// A 2-site half-sweep (tensorci2.rs:1695-1725) is a chain of dependent bond updates: bond b needs the rows I_{b+1} (forward)
// or the columns J_b (backward) its predecessor selected.  Until round 2 that dependency ran through the host (three launches
// and a completion round trip per bond).  Here the index sets live on the device as TABLES — per site and side a list of
// (code, accumulators): `code` identifies the multi-index (mixed-radix number, see below), the accumulators are what the
// built-in functor needs (include/t4a_testfunctions.h) — and the host part of update_pivots (tensorci2.rs:1833-1846,
// :1934-1949) becomes three small kernels:
//
//   chain_indep_kernel   once per half-sweep, one workgroup per bond: the side of every bond that does NOT depend on the
//                        chain (forward: the columns kron(d_{b+1}, J_{b+1}) ∪ HJ_b; backward: the rows kron(I_b, d_b) ∪ HI_{b+1}).
//   chain_prep_kernel    between two rrLU launches, one workgroup: (1) gather — the pivots the previous bond selected
//                        (permutation prefix of its rrLU, non_empty_or_first, tensorci2.rs:1813-1819) go from that bond's
//                        row / column lists into the tables I_{b'+1}, J_{b'} (and into their pinned host mirrors: the host's
//                        master copy is refreshed without any host work); (2) the DEPENDENT side of this bond: Kronecker
//                        product with the new site (tensorci2.rs:1224-1246) ∪ the history extras that are not yet present,
//                        order preserving like the reference's `contains` loop (:1837-1846): membership of an extra in the
//                        Kronecker part is membership of its parent multi-index in the table (hash in LDS), survivors are
//                        compacted by a prefix sum; (3) the dimensions of the bond, for the kernels behind it on the stream.
//   (xcd_spec_work,      the candidate matrix of bond b, evaluated SPECULATIVELY while the rrLU of the previous bond is still
//   kernels_rrlu_xcd.hip) running: that launch occupies one XCD, and its pass-through workgroups on the other seven do this
//                        instead of returning at once.  The dependent side of bond b can only consist of children of entries
//                        of the previous bond's dependent list (a pivot is one of its candidates) and of extras, so f is
//                        evaluated for ALL of those candidates x the independent list; chain_prep_kernel then only writes a
//                        row map (list position -> candidate) and the rrLU kernel loads its matrix through it.  ~3x more
//                        evaluations than needed, all of them off the critical path: between two rrLU launches only
//                        chain_prep_kernel runs.  chain_pi_kernel evaluates a bond nobody could speculate on.
//
// Codes.  I_p holds prefixes (i_0 … i_{p-1}), J_p suffixes (j_{p+1} … j_{n-1}); both are numbered so that the Kronecker step
// is one multiply-add: code(I_{p+1} element (i, s)) = s + d_p * code(i), code(J_{p-1} element (s, j)) = s + d_p * code(j),
// the empty index has code 0.  The parent of an element is code / d.  A chain is only run on the device when the product of
// all local dimensions fits 63 bits (tci2_chain.hip).
//
// A bond whose predecessor did not complete (bounded spin gave up, capacity exceeded) is POISONED: dimensions 0, every later
// kernel of the chain does nothing, the host re-runs the half-sweep from that bond with the per-bond path.
#include "kernels.hpp"
#include "kernels_rrlu_w1_body.hpp"

namespace t4a {

namespace {

__device__ __forceinline__ const char* chain_kernarg_base()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (const char*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}

constexpr int CHAIN_T = 1024;       // threads of a list-building workgroup (launched chain; the bodies below take the launch's count)
constexpr int WALK_T = 1024;        // threads of the persistent half-sweep (measured: 256 — one wave per SIMD — is no faster through the short uniform
                                    // sections between barriers and four times slower through the candidate matrix of a wide bond)
constexpr int CHAIN_HASH = 4096;    // LDS hash slots (tables hold at most CHAIN_MAX_SET = 1024 entries per site)

__device__ __forceinline__ unsigned chain_hash(unsigned long long c)
{
    c ^= c >> 33;
    c *= 0xff51afd7ed558ccdull;
    c ^= c >> 29;
    return (unsigned)c & (CHAIN_HASH - 1);
}

// exclusive prefix sum of one flag per thread over the workgroup; returns the position of this thread, *total the sum
__device__ __forceinline__ int block_scan_flag(bool flag, int* wave_sums, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    __syncthreads(); // wave_sums may still be read from a previous call
    if (lane == 0) wave_sums[wave] = __builtin_popcountll(bal);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) {
        const int v = wave_sums[w];
        if (w < wave) base += v;
        tot += v;
    }
    *total = tot;
    return base + before;
}

// One side of a bond: out = kron(parent table, d) followed by the extras that are not in it.
// PARENT_OUTER: (parent outer, s inner: rows, tensorci2.rs:1224-1234) else (s outer, parent inner: columns, :1236-1246).
// srcpos / rowmap (optional): the candidate index of every list entry for the speculative candidate matrix —
// parent k sits at position srcpos[k] of the previous bond's dependent list, its child with digit s is candidate
// srcpos[k] * d + s; extra e is candidate n_prev * d + e.
// Returns the count (uniform), -1 on capacity overflow.
template <bool PARENT_OUTER>
__device__ int build_side(const uint64_t* __restrict__ pcode, const uint64_t* __restrict__ pacc, int np, int d,
                          const uint64_t* __restrict__ w, int K, int total_w, int w_off, const uint64_t* __restrict__ ecode,
                          const uint64_t* __restrict__ eacc, int ne, uint64_t* __restrict__ ocode, uint64_t* __restrict__ oacc, int ocap,
                          const int* srcpos, int n_prev, int* __restrict__ rowmap, unsigned long long* hkeys, int* wave_sums)
{
    const int tid = threadIdx.x;
    const int m0 = np * d;
    if (m0 > ocap) return -1;
    if (ne > 0) {
        for (int e = tid; e < CHAIN_HASH; e += (int)blockDim.x) hkeys[e] = 0ull;
        __syncthreads();
    }
    // Kronecker part
    for (int idx = tid; idx < m0; idx += (int)blockDim.x) {
        const int i = PARENT_OUTER ? idx / d : idx % np;
        const int s = PARENT_OUTER ? idx % d : idx / np;
        const uint64_t c = pcode[i];
        ocode[idx] = (uint64_t)s + (uint64_t)d * c;
        for (int k = 0; k < K; ++k) oacc[(size_t)idx * K + k] = pacc[(size_t)i * K + k] + w[(size_t)k * total_w + w_off + s];
        if (rowmap) rowmap[idx] = srcpos[i] * d + s;
    }
    if (ne <= 0) return m0;
    // hash of the parents, then the extras in order (one per thread and round; ne <= CHAIN_T in practice: one round)
    for (int i = tid; i < np; i += (int)blockDim.x) {
        const unsigned long long key = (unsigned long long)pcode[i] + 1ull;
        unsigned h = chain_hash(key);
        for (;;) {
            const unsigned long long old = atomicCAS(&hkeys[h], 0ull, key);
            if (old == 0ull || old == key) break;
            h = (h + 1) & (CHAIN_HASH - 1);
        }
    }
    __syncthreads();
    int count = m0;
    for (int base = 0; base < ne; base += (int)blockDim.x) {
        const int e = base + tid;
        bool keep = false;
        uint64_t c = 0ull;
        if (e < ne) {
            c = ecode[e];
            const unsigned long long key = (unsigned long long)(c / (uint64_t)d) + 1ull;
            unsigned h = chain_hash(key);
            keep = true;
            for (;;) {
                const unsigned long long v = hkeys[h];
                if (v == 0ull) break;
                if (v == key) {
                    keep = false;
                    break;
                }
                h = (h + 1) & (CHAIN_HASH - 1);
            }
        }
        int tot = 0;
        const int pos = block_scan_flag(keep, wave_sums, &tot);
        if (count + tot > ocap) return -1;
        if (keep) {
            const int o = count + pos;
            ocode[o] = c;
            for (int k = 0; k < K; ++k) oacc[(size_t)o * K + k] = eacc[(size_t)e * K + k];
            if (rowmap) rowmap[o] = n_prev * d + e;
        }
        count += tot;
    }
    return count;
}

// the independent side of every bond of the half-sweep: blockIdx.x = bond
__device__ __forceinline__ void chain_indep_body(const ChainCommon& c)
{
    __shared__ unsigned long long hkeys[CHAIN_HASH];
    __shared__ int wave_sums[CHAIN_T / 64];
    const int b = blockIdx.x;
    const int K = c.K;
    const size_t cap = (size_t)c.cap;
    uint64_t* ocode = c.ind_code + (size_t)b * c.ind_cap;
    uint64_t* oacc = c.ind_acc + (size_t)b * c.ind_cap * K;
    int n;
    if (c.one_site) { // 1-site sweep: the table itself (forward: the columns J_b; backward: the rows I_{b+1})
        const ChainTab& T = c.forward ? c.J : c.I;
        const int site = c.forward ? b : b + 1;
        const int np = T.cnt[site];
        n = (np >= 1 && np <= c.cap && np <= c.ind_cap) ? np : -1;
        for (int i = threadIdx.x; i < n; i += (int)blockDim.x) {
            ocode[i] = T.code[(size_t)site * cap + i];
            for (int k = 0; k < K; ++k) oacc[(size_t)i * K + k] = T.acc[((size_t)site * cap + i) * K + k];
        }
    } else if (c.forward) { // columns: kron(d_{b+1}, J_{b+1}) ∪ HJ_b
        const int np = c.J.cnt[b + 1], ne = c.use_extras ? c.HJ.cnt[b] : 0;
        n = (np >= 1 && np <= c.cap && ne <= c.cap)
                ? build_side<false>(c.J.code + (size_t)(b + 1) * cap, c.J.acc + (size_t)(b + 1) * cap * K, np, c.ldim[b + 1], c.w, K, c.total,
                                    c.woff[b + 1], c.HJ.code + (size_t)b * cap, c.HJ.acc + (size_t)b * cap * K, ne, ocode, oacc, c.ind_cap,
                                    nullptr, 0, nullptr, hkeys, wave_sums)
                : -1;
    } else { // rows: kron(I_b, d_b) ∪ HI_{b+1}
        const int np = c.I.cnt[b], ne = c.use_extras ? c.HI.cnt[b + 1] : 0;
        n = (np >= 1 && np <= c.cap && ne <= c.cap)
                ? build_side<true>(c.I.code + (size_t)b * cap, c.I.acc + (size_t)b * cap * K, np, c.ldim[b], c.w, K, c.total, c.woff[b],
                                   c.HI.code + (size_t)(b + 1) * cap, c.HI.acc + (size_t)(b + 1) * cap * K, ne, ocode, oacc, c.ind_cap,
                                   nullptr, 0, nullptr, hkeys, wave_sums)
                : -1;
    }
    if (threadIdx.x == 0) c.ind_cnt[b] = n;
    // housekeeping of the new chain (grid-stride over all workgroups): nothing of this is read before this kernel has ended
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    if (c.snap_dst) {
        for (size_t i = gtid; i < c.snap_words; i += gsz) c.snap_dst[i] = c.I.code[i]; // (I and J families are adjacent)
        for (size_t i = gtid; i < (size_t)2 * c.n_sites; i += gsz) c.snap_cnt_dst[i] = c.I.cnt[i];
    }
    for (size_t i = gtid; i < c.zero_a_words; i += gsz) c.zero_a[i] = 0ull;
    for (size_t i = gtid; i < c.zero_b_words; i += gsz) c.zero_b[i] = 0ull;
}
__global__ void __launch_bounds__(CHAIN_T) chain_indep_kernel(ChainCommon c) { chain_indep_body(c); }
// group chain: blockIdx.y = handle
__global__ void __launch_bounds__(CHAIN_T) chain_indep_group_kernel(const ChainGroupSlot* __restrict__ slots)
{
    chain_indep_body(slots[blockIdx.y].c);
}

// What the persistent half-sweep (chain_walk_kernel below) keeps in the LDS between two bonds instead of reading it back from
// global memory: a step of the walk is a chain of DEPENDENT loads (result of the previous bond -> its permutations -> the codes of
// its pivots -> the new list -> its size), each an L2 round trip of ~0.4 us when it goes through memory.
constexpr int WALK_MAX_BONDS = 128;
constexpr int WALK_MAX_LIST = 64;
constexpr int WALK_MAX_IND = 32;    // entries of an independent list (= columns of the one-wave kernel the walk uses)
constexpr int WALK_MAX_W = 512;     // weight entries (K x sum of the local dimensions) kept in the LDS
struct WalkShared {
    int prev_ires[4];                                   // {rank, gave-up code, NaN flag, token} of the previous bond's rrLU
    int perm_rp[WALK_MAX_LIST], perm_cp[WALK_MAX_LIST]; // its permutations (rows / columns of the matrix as the host sees it)
    int np;                                             // parents gathered for this bond (= that rank, at least 1)
    int pad_;
    uint64_t par_code[WALK_MAX_LIST], par_acc[WALK_MAX_LIST * T4A_FN_MAX_ACC]; // ... their codes and accumulators
    uint64_t dep_code[WALK_MAX_LIST], dep_acc[WALK_MAX_LIST * T4A_FN_MAX_ACC]; // the dependent list of the bond in flight
    int dims[WALK_MAX_BONDS * 4];                       // ChainCommon::dims of the whole sweep (copied out at the end)
    int ind_cnt[WALK_MAX_BONDS];                        // ChainCommon::ind_cnt (copied in at the start)
    // what does not depend on the walk is fetched one bond AHEAD (by the waves that idle while wave 0 factorises), bond b into slot
    // b & 1: the independent list (the candidate matrix of bond b and the gather of bond b + 1 read it) and the history extras
    uint64_t ind_code[2][WALK_MAX_IND], ind_acc[2][WALK_MAX_IND * T4A_FN_MAX_ACC];
    uint64_t ext_code[2][WALK_MAX_LIST], ext_acc[2][WALK_MAX_LIST * T4A_FN_MAX_ACC];
    int ext_cnt[2];
    int ldim[WALK_MAX_BONDS + 1], woff[WALK_MAX_BONDS + 1]; // ChainCommon::ldim / woff (read once per bond by the preparation: a scalar load that misses every time)
    uint64_t w[WALK_MAX_W];
};

__device__ __forceinline__ void chain_prep_body(const ChainCommon& c, const ChainPrepArgs& p, unsigned long long* dbg = nullptr, WalkShared* ws = nullptr)
{
    unsigned long long dbg_last = dbg ? wall_clock64() : 0ull;
    auto stamp = [&](int slot) {
        if (dbg) {
            const unsigned long long now = wall_clock64();
            if (threadIdx.x == 0) dbg[slot] += now - dbg_last;
            dbg_last = now;
        }
    };
    __shared__ unsigned long long hkeys[CHAIN_HASH];
    __shared__ int wave_sums[CHAIN_T / 64];
    __shared__ int s_srcpos[CHAIN_MAX_SET];
    __shared__ int s_poison;
    const int tid = threadIdx.x;
    const int K = c.K;
    const size_t cap = (size_t)c.cap;
    if (tid == 0) s_poison = 0;
    __syncthreads();

    // ---- 1. pivots of the previous bond -> tables I_{pb+1}, J_{pb} (device and pinned host mirror) ----
    int n_prev_dep = 0; // entries of the previous bond's dependent list (candidates' parents)
    if (p.prev_b >= 0) {
        const int pb = p.prev_b;
        const int* pd = c.dims + (size_t)pb * 4;
        const int pm = pd[0], pn = pd[1];
        const bool bad = pd[2] != 0 || p.prev_iresult[1] != 0 || p.prev_iresult[3] != (int)p.prev_token || pm <= 0 || pn <= 0;
        if (bad) {
            if (tid == 0) s_poison = 1;
        } else {
            const int r = p.prev_iresult[0];
            const int cnt = r > 0 ? r : 1; // non_empty_or_first (tensorci2.rs:1813-1819)
            if (cnt > c.cap || cnt > pm || cnt > pn) {
                if (tid == 0) s_poison = 1;
            } else {
                n_prev_dep = c.forward ? pm : pn;
                // forward: rows of the previous bond were its dependent list, columns its independent list; backward: the reverse
                const uint64_t* const pind_code = ws ? ws->ind_code[pb & 1] : c.ind_code + (size_t)pb * c.ind_cap;
                const uint64_t* const pind_acc = ws ? ws->ind_acc[pb & 1] : c.ind_acc + (size_t)pb * c.ind_cap * K;
                const uint64_t* rcode = c.forward ? c.dep_code : pind_code;
                const uint64_t* racc = c.forward ? c.dep_acc : pind_acc;
                const uint64_t* ccode = c.forward ? pind_code : c.dep_code;
                const uint64_t* cacc = c.forward ? pind_acc : c.dep_acc;
                const size_t oi = (size_t)(pb + 1) * cap, oj = (size_t)pb * cap;
                for (int k = tid; k < cnt; k += (int)blockDim.x) {
                    const int ri = r > 0 ? p.prev_rowperm[k] : 0, ci = r > 0 ? p.prev_colperm[k] : 0;
                    const uint64_t rc = rcode[ri], cc = ccode[ci];
                    c.I.code[oi + k] = rc;
                    c.J.code[oj + k] = cc;
                    if (ws) ws->par_code[k] = c.forward ? rc : cc; // (the parents of the list that is built next)
                    if (!p.defer_host_writes) {
                        c.mI.code[oi + k] = rc;
                        c.mJ.code[oj + k] = cc;
                    }
                    for (int q = 0; q < K; ++q) {
                        const uint64_t ra = racc[(size_t)ri * K + q], ca = cacc[(size_t)ci * K + q];
                        c.I.acc[(oi + k) * K + q] = ra;
                        c.J.acc[(oj + k) * K + q] = ca;
                        if (ws) ws->par_acc[(size_t)k * K + q] = c.forward ? ra : ca;
                        if (!p.defer_host_writes) {
                            c.mI.acc[(oi + k) * K + q] = ra;
                            c.mJ.acc[(oj + k) * K + q] = ca;
                        }
                    }
                    s_srcpos[k] = c.forward ? ri : ci; // position of the new parent in the previous dependent list
                }
                if (tid == 0) {
                    if (ws) ws->np = cnt;
                    c.I.cnt[pb + 1] = cnt;
                    c.J.cnt[pb] = cnt;
                    if (!p.defer_host_writes) {
                        c.mI.cnt[pb + 1] = cnt;
                        c.mJ.cnt[pb] = cnt;
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads(); // the gather has read the old dependent list and written the tables: the list may be overwritten now
    }
    stamp(4);
    if (!p.do_build) return;
    const int b = p.b;
    int* dims = c.dims + (size_t)b * 4;
    int* hdims = (c.hdims && !p.defer_host_writes) ? c.hdims + (size_t)b * 4 : nullptr;
    if (s_poison) {
        if (tid == 0) {
            dims[0] = dims[1] = dims[3] = 0;
            dims[2] = 1;
            if (hdims) {
                hdims[0] = hdims[1] = 0;
                hdims[2] = 1;
            }
        }
        return;
    }

    // ---- 2. the dependent side of bond b (rows when sweeping forward, columns when sweeping backward) ----
    const bool mapped = p.prev_b >= 0 && p.with_rowmap;
    int nd = -1, lda = 0;
    // (persistent half-sweep: the parents were gathered a moment ago and wait in the LDS)
    const bool par_lds = ws != nullptr && p.prev_b >= 0;
    const uint64_t* const xcode = ws ? ws->ext_code[b & 1] : (c.forward ? c.HI.code + (size_t)(b + 1) * cap : c.HJ.code + (size_t)b * cap);
    const uint64_t* const xacc = ws ? ws->ext_acc[b & 1] : (c.forward ? c.HI.acc + (size_t)(b + 1) * cap * K : c.HJ.acc + (size_t)b * cap * K);
    if (c.forward) {
        const int np = par_lds ? ws->np : c.I.cnt[b], ne = c.use_extras ? (ws ? ws->ext_cnt[b & 1] : c.HI.cnt[b + 1]) : 0;
        if (np >= 1 && np <= c.cap && ne <= c.cap)
            nd = build_side<true>(par_lds ? ws->par_code : c.I.code + (size_t)b * cap, par_lds ? ws->par_acc : c.I.acc + (size_t)b * cap * K, np, c.ldim[b], c.w, K, c.total, c.woff[b],
                                  xcode, xacc, ne, c.dep_code, c.dep_acc, c.dep_cap,
                                  s_srcpos, n_prev_dep, mapped ? c.rowmap : nullptr, hkeys, wave_sums);
        lda = n_prev_dep * c.ldim[b] + ne;
    } else {
        const int np = par_lds ? ws->np : c.J.cnt[b + 1], ne = c.use_extras ? (ws ? ws->ext_cnt[b & 1] : c.HJ.cnt[b]) : 0;
        if (np >= 1 && np <= c.cap && ne <= c.cap)
            nd = build_side<false>(par_lds ? ws->par_code : c.J.code + (size_t)(b + 1) * cap, par_lds ? ws->par_acc : c.J.acc + (size_t)(b + 1) * cap * K, np, c.ldim[b + 1], c.w, K, c.total,
                                   c.woff[b + 1], xcode, xacc, ne, c.dep_code, c.dep_acc,
                                   c.dep_cap, s_srcpos, n_prev_dep, mapped ? c.rowmap : nullptr, hkeys, wave_sums);
        lda = n_prev_dep * c.ldim[b + 1] + ne;
    }
    stamp(5);
    if (tid == 0) {
        const int ni = c.ind_cnt[b];
        const bool ok = nd > 0 && ni > 0;
        const int M = c.forward ? nd : ni, N = c.forward ? ni : nd;
        dims[0] = ok ? M : 0;
        dims[1] = ok ? N : 0;
        dims[2] = ok ? 0 : 1;
        dims[3] = mapped ? lda : (ok ? nd : 0); // leading dimension of the matrix the rrLU kernel loads (its rows = dependent side)
        if (hdims) {
            hdims[0] = ok ? M : 0;
            hdims[1] = ok ? N : 0;
            hdims[2] = ok ? 0 : 1;
        }
    }
}

__global__ void __launch_bounds__(CHAIN_T) chain_prep_kernel(ChainCommon c, ChainPrepArgs p)
{
    const unsigned long long t0 = p.dbg ? wall_clock64() : 0ull;
    chain_prep_body(c, p, p.dbg);
    if (p.dbg && threadIdx.x == 0) p.dbg[6] += wall_clock64() - t0;
}
// group chain: blockIdx.x = handle (a handle that has nothing to do at this bond carries prev_b < 0 and do_build == 0)
__global__ void __launch_bounds__(CHAIN_T) chain_prep_group_kernel(const ChainGroupSlot* __restrict__ slots, ChainPrepGroupArgs g)
{
    // (dynamic index into a by-value argument: read through the kernel-argument segment, not through a private copy)
    const ChainPrepArgs* pa = reinterpret_cast<const ChainPrepArgs*>(chain_kernarg_base() + sizeof(const ChainGroupSlot*)) + blockIdx.x;
    (void)g;
    chain_prep_body(slots[blockIdx.x].c, *pa);
}

// Candidate matrix of a bond nobody speculated on (first bond of a chain, or the previous launch was a single workgroup):
// out[j * nd + i] = f(dependent i, independent j)
__device__ __forceinline__ void chain_pi_body(const ChainCommon& c, const FnDevice& fn, int b, double* __restrict__ out)
{
    const int* dm = c.dims + (size_t)b * 4;
    if (dm[2] != 0) return;
    const int K = fn.n_acc;
    const int nd = c.forward ? dm[0] : dm[1], ni = c.forward ? dm[1] : dm[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * (int)blockDim.x >= nd) return;
    uint64_t racc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
    if (i < nd)
        for (int k = 0; k < K; ++k) racc[k] = c.dep_acc[(size_t)i * K + k];
    const uint64_t* ia = c.ind_acc + (size_t)b * c.ind_cap * K;
    for (int j = blockIdx.y; j < ni; j += gridDim.y) {
        if (i < nd) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < K; ++k) acc[k] = racc[k] + ia[(size_t)j * K + k];
            out[(size_t)j * nd + i] = t4a_fn_value(fn.fid, acc, fn.params);
        }
    }
}
__global__ void __launch_bounds__(256) chain_pi_kernel(ChainCommon c, FnDevice fn, int b, double* __restrict__ out) { chain_pi_body(c, fn, b, out); }
// group chain: blockIdx.z = handle
__global__ void __launch_bounds__(256) chain_pi_group_kernel(const ChainGroupSlot* __restrict__ slots, int b)
{
    const ChainGroupSlot& s = slots[blockIdx.z];
    chain_pi_body(s.c, s.fn, b, s.pi);
}


// The host's copies of what a chain wrote, in bulk behind it: every table (I_1 .. I_nb, J_0 .. J_{nb-1}) into the pinned mirror, the
// dimensions into hdims.  A preparation that stores into pinned memory itself pays the PCIe write acknowledgement at its kernel's
// end, on the critical path between two rrLU launches; here it is paid once per half-sweep.  (The gather of a poisoned bond wrote
// nothing: chain_finish ignores everything from the first poisoned bond on.)
__device__ __forceinline__ void chain_mirror_body(const ChainCommon& c, int nb)
{
    const int K = c.K;
    const size_t cap = (size_t)c.cap;
    const int gtid = (int)(blockIdx.x * blockDim.x + threadIdx.x), gsz = (int)(gridDim.x * blockDim.x);
    // a WAVE per (site, family): its lanes copy the entries, the count is one load per wave.  (Round 5: as one loop over the 2 (nb + 1)
    // tables with every thread of the grid reading every count in turn, the copy was 38 dependent device-scope loads long at d = 20 —
    // ~20 of the 135 us of a persistent half-sweep of configs[1].)
    const int lane = (int)(threadIdx.x & 63), gwave = gtid >> 6, nwaves = gsz >> 6;
    for (int pf = gwave; pf < 2 * (nb + 1); pf += nwaves) {
        const int site = pf >> 1, fam = pf & 1;
        if ((fam == 0 && site == 0) || (fam == 1 && site == nb)) continue; // (I_0 and J_{n-1} are never written)
        const ChainTab& T = fam == 0 ? c.I : c.J;
        const ChainTab& Mr = fam == 0 ? c.mI : c.mJ;
        const int cnt = __hip_atomic_load(T.cnt + site, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cnt < 0 || cnt > c.cap) continue;
        for (int e = lane; e < cnt * (1 + K); e += 64) {
            if (e < cnt) Mr.code[(size_t)site * cap + e] = T.code[(size_t)site * cap + e];
            else Mr.acc[(size_t)site * cap * K + (e - cnt)] = T.acc[(size_t)site * cap * K + (e - cnt)];
        }
        if (lane == 0) Mr.cnt[site] = cnt;
    }
    if (c.hdims)
        for (int e = gtid; e < nb * 4; e += gsz)
            if ((e & 3) != 3) c.hdims[e] = __hip_atomic_load(c.dims + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(1024) chain_mirror_kernel(ChainCommon c, int nb) { chain_mirror_body(c, nb); }

// the last site's tensor of a forward 1-site sweep, straight from the tables (see kernels.hpp): value = f(acc(I_{n-1}[l]) + w[s] + acc(J_{n-1}[r])),
// the accumulator sums of pi_eval_kernel over the rows kron(I_{n-1}, d_{n-1}) (parent outer, digit inner) and the columns J_{n-1}
__global__ void __launch_bounds__(256) chain_last_core_kernel(ChainCommon c, FnDevice fn, double* __restrict__ core, int max_entries)
{
    const int site = c.n_sites - 1, K = c.K;
    const size_t cap = (size_t)c.cap;
    const int L = c.I.cnt[site], R = c.J.cnt[site], S = c.ldim[site], woff = c.woff[site];
    if (L < 1 || L > c.cap || R < 1 || R > c.cap || S < 1) return;
    const long long total = (long long)L * S * R;
    if (total > (long long)max_entries) return; // (cannot happen: the buffer is sized for the upper bounds of the plan)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % L), s_ = (int)((e / L) % S), r = (int)(e / ((long long)L * S));
        uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
        for (int q = 0; q < K; ++q)
            acc[q] = c.I.acc[((size_t)site * cap + l) * K + q] + c.w[(size_t)q * c.total + woff + s_] + c.J.acc[((size_t)site * cap + r) * K + q];
        core[e] = t4a_fn_value(fn.fid, acc, fn.params);
    }
}

// ------------------------------------------------------------------------------------------------
// The persistent half-sweep ("walker"): ONE workgroup walks all bonds of a half-sweep whose matrices fit the one-wave rrLU kernel
// (at most 64 x 64: BASELINE configs[1], the first iterations of every run).  Per bond: the preparation above, the candidate
// matrix by all threads, the factorisation by wave 0 (kernels_rrlu_w1_body.hpp) — no launch, no dispatch latency and no
// end-of-kernel flush between them; the result blocks, dimensions, tables and mirrors are those of the launched chain, so the
// host's chain_finish does not know the difference.  VERDICT round 3, item 3 (tensorci2.rs:1695-1725).
//
// Everything one phase writes and the next one reads goes through global memory inside one kernel: the vector stores of a
// workgroup are visible to its own later loads (one compute unit, one L1), but uniform loads go through the SCALAR data cache,
// which knows nothing of them — it is invalidated behind every barrier that separates a producer from a consumer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void walk_phase_barrier(bool scalar_cache = true)
{
    if (scalar_cache) {
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        // (everything that passes from phase to phase sits in the LDS — the one-wave preparation —: the barrier orders the LDS only.  A
        // full __syncthreads also waits for every outstanding global store — the tables and the result block a bond leaves behind are
        // written for the host and for later kernels — vmcnt(0): an acknowledgement from the L2, twice per bond)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
// x / d for 0 <= x < 8192, 1 <= d <= 128 through the single-precision reciprocal (rd = 1.0f / d, correctly rounded): (x + 0.5) / d stays at least
// 0.5 / 128 away from every integer, three orders of magnitude more than the rounding of two float operations — checked exhaustively on the host.
// (A run-time 32-bit division is ~35 instructions on this ISA, a 64-bit one well over a hundred: a lone wave pays ~10 cycles for each.)
__device__ __forceinline__ int walk_div_small(int x, float rd) { return (int)(((float)x + 0.5f) * rd); }
__device__ __forceinline__ int walk_load_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The preparation of a bond of the persistent half-sweep by ONE wave, without a barrier (lists of at most 64 entries: a lane per
// entry; everything it reads — result and permutations of the previous bond, the old dependent list, the prefetched independent
// list and extras, the weights — sits in the LDS): the same gather (non_empty_or_first, tensorci2.rs:1813-1819), the same
// Kronecker order (tensorci2.rs:1224-1246) and the same order-preserving union with the extras (:1837-1846) as chain_prep_body,
// whose hash and prefix sum over 1 024 threads cost five barriers and ~3 - 4.5 us per bond for lists of a handful of entries.
// Membership of an extra in the Kronecker part = its parent code is one of the gathered parents (a loop over <= 64 LDS broadcasts).
template <bool FORWARD>
__device__ __forceinline__ void walk_prep_wave0(const ChainCommon& c, WalkShared* ws, int b, int prev_b, unsigned prev_token, bool do_build, unsigned long long* dbg = nullptr)
{
    const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;
    const int lane = threadIdx.x & 63;
    const int K = c.K;
    const size_t cap = (size_t)c.cap;
    bool poison = false;
    int np = 0;
    // ---- 1. pivots of the previous bond -> tables I_{pb+1}, J_{pb}; they are the parents of this bond's dependent list ----
    if (prev_b >= 0) {
        const int pb = prev_b;
        const int pm = ws->dims[pb * 4], pn = ws->dims[pb * 4 + 1];
        const bool bad = ws->dims[pb * 4 + 2] != 0 || ws->prev_ires[1] != 0 || ws->prev_ires[3] != (int)prev_token || pm <= 0 || pn <= 0;
        const int r = ws->prev_ires[0];
        const int cnt = r > 0 ? r : 1;
        if (bad || cnt > c.cap || cnt > pm || cnt > pn || cnt > WALK_MAX_LIST) {
            poison = true;
        } else {
            const uint64_t* const pind_code = ws->ind_code[pb & 1];
            const uint64_t* const pind_acc = ws->ind_acc[pb & 1];
            uint64_t rc = 0, cc = 0, ra[T4A_FN_MAX_ACC] = {0, 0, 0, 0}, ca[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            if (lane < cnt) {
                const int ri = r > 0 ? ws->perm_rp[lane] : 0, ci = r > 0 ? ws->perm_cp[lane] : 0;
                // forward: rows of the previous bond were its dependent list, columns its independent list; backward: the reverse
                rc = FORWARD ? ws->dep_code[ri] : pind_code[ri];
                cc = FORWARD ? pind_code[ci] : ws->dep_code[ci];
                for (int q = 0; q < K; ++q) {
                    ra[q] = FORWARD ? ws->dep_acc[(size_t)ri * K + q] : pind_acc[(size_t)ri * K + q];
                    ca[q] = FORWARD ? pind_acc[(size_t)ci * K + q] : ws->dep_acc[(size_t)ci * K + q];
                }
            }
            // (every read of the old dependent list is in registers now: the parents may be written)
            if (lane < cnt) {
                const size_t oi = (size_t)(pb + 1) * cap, oj = (size_t)pb * cap;
                c.I.code[oi + lane] = rc;
                c.J.code[oj + lane] = cc;
                ws->par_code[lane] = FORWARD ? rc : cc;
                for (int q = 0; q < K; ++q) {
                    c.I.acc[(oi + lane) * K + q] = ra[q];
                    c.J.acc[(oj + lane) * K + q] = ca[q];
                    ws->par_acc[(size_t)lane * K + q] = FORWARD ? ra[q] : ca[q];
                }
            }
            if (lane == 0) {
                c.I.cnt[pb + 1] = cnt;
                c.J.cnt[pb] = cnt;
            }
            np = cnt;
        }
    }
    const unsigned long long dbg_t1 = dbg ? wall_clock64() : 0ull;
    if (dbg && (threadIdx.x & 63) == 0) dbg[4] += dbg_t1 - dbg_t0; // (T4A_WALK_DEBUG: the gather)
    if (!do_build) return;
    int* const dims = ws->dims + b * 4;
    if (poison) {
        if (lane == 0) {
            dims[0] = dims[1] = dims[3] = 0;
            dims[2] = 1;
        }
        return;
    }
    // ---- 2. the dependent side of bond b: kron(parents, d) then the extras whose parent is not among the parents ----
    const int site = FORWARD ? b : b + 1;
    const int d = c.ldim[site], woff = c.woff[site];
    const ChainTab& PT = FORWARD ? c.I : c.J; // (first bond of the walk: the parents are the table as the previous kernels left it)
    if (prev_b < 0) {
        np = PT.cnt[site];
        if (np >= 1 && np <= WALK_MAX_LIST && lane < np) {
            ws->par_code[lane] = PT.code[(size_t)site * cap + lane];
            for (int q = 0; q < K; ++q) ws->par_acc[(size_t)lane * K + q] = PT.acc[((size_t)site * cap + lane) * K + q];
        }
    }
    const int ne = c.use_extras ? ws->ext_cnt[b & 1] : 0;
    const int m0 = np * d;
    int nd = -1;
    if (np >= 1 && np <= c.cap && np <= WALK_MAX_LIST && ne >= 0 && ne <= c.cap && ne <= WALK_MAX_LIST && m0 <= WALK_MAX_LIST) {
        uint64_t oc = 0, oa[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
        if (lane < m0) { // rows: (parent outer, digit inner), columns: (digit outer, parent inner)
            const int qd = walk_div_small(lane, 1.0f / (float)(FORWARD ? d : np)); // lane / d (forward) or lane / np
            const int rd_ = lane - qd * (FORWARD ? d : np);
            const int i = FORWARD ? qd : rd_, sdig = FORWARD ? rd_ : qd;
            oc = (uint64_t)sdig + (uint64_t)d * ws->par_code[i];
            for (int q = 0; q < K; ++q) oa[q] = ws->par_acc[(size_t)i * K + q] + c.w[(size_t)q * c.total + woff + sdig];
        }
        bool keep = false;
        uint64_t xc = 0;
        if (lane < ne) {
            xc = ws->ext_code[b & 1][lane];
            // xc / d == parent  <=>  parent d <= xc < parent d + d (no 64-bit division; a code times a local dimension does not overflow:
            // it is the code of a child)
            keep = true;
            for (int pi = 0; pi < np; ++pi) keep = keep && !((xc - ws->par_code[pi] * (uint64_t)d) < (uint64_t)d);
        }
        const unsigned long long km = __ballot(keep);
        const int nkeep = __builtin_popcountll(km);
        if (m0 + nkeep <= WALK_MAX_LIST) {
            // (the Kronecker part read the parents, not the dependent list: it may be overwritten now)
            if (lane < m0) {
                ws->dep_code[lane] = oc;
                for (int q = 0; q < K; ++q) ws->dep_acc[(size_t)lane * K + q] = oa[q];
            }
            if (keep) {
                const int pos = m0 + __builtin_popcountll(km & ((1ull << lane) - 1ull));
                ws->dep_code[pos] = xc;
                for (int q = 0; q < K; ++q) ws->dep_acc[(size_t)pos * K + q] = ws->ext_acc[b & 1][(size_t)lane * K + q];
            }
            nd = m0 + nkeep;
        }
    }
    if (lane == 0) {
        const int ni = ws->ind_cnt[b];
        const bool ok = nd > 0 && ni > 0;
        dims[0] = ok ? (FORWARD ? nd : ni) : 0;
        dims[1] = ok ? (FORWARD ? ni : nd) : 0;
        dims[2] = ok ? 0 : 1;
        dims[3] = ok ? nd : 0;
    }    if (dbg && lane == 0) dbg[5] += wall_clock64() - dbg_t1; // (T4A_WALK_DEBUG: the dependent list)
}

// the static inputs of bond b (independent list, history extras) into slot b & 1 of the LDS, by threads first .. first + count - 1
__device__ __forceinline__ void walk_prefetch(const ChainCommon& c, WalkShared* ws, int b, int first, int count)
{
    const int t = (int)threadIdx.x - first;
    if (t < 0 || count <= 0) return;
    const int K = c.K, slot = b & 1;
    int ni = ws->ind_cnt[b];
    ni = ni < 0 ? 0 : (ni > WALK_MAX_IND ? WALK_MAX_IND : ni);
    const uint64_t* icode = c.ind_code + (size_t)b * c.ind_cap;
    const uint64_t* iacc = c.ind_acc + (size_t)b * c.ind_cap * K;
    for (int e = t; e < ni * (1 + K); e += count) {
        if (e < ni) ws->ind_code[slot][e] = icode[e];
        else ws->ind_acc[slot][e - ni] = iacc[e - ni];
    }
    if (c.use_extras) {
        const ChainTab& H = c.forward ? c.HI : c.HJ;
        const int hsite = c.forward ? b + 1 : b;
        const int ne_raw = H.cnt[hsite];
        const int ne = ne_raw < 0 ? 0 : (ne_raw > WALK_MAX_LIST ? WALK_MAX_LIST : ne_raw);
        const size_t cap = (size_t)c.cap;
        for (int e = t; e < ne * (1 + K); e += count) {
            if (e < ne) ws->ext_code[slot][e] = H.code[(size_t)hsite * cap + e];
            else ws->ext_acc[slot][e - ne] = H.acc[(size_t)hsite * cap * K + (e - ne)];
        }
        if (t == 0) ws->ext_cnt[slot] = ne_raw;
    }
}

template <int NC, bool FACTORS> constexpr size_t walk_lds_bytes()
{
    // [WalkShared][the candidate matrix (64 x NC doubles), later the one-wave kernel's side buffers: the matrix is in registers by then]
    constexpr size_t w1 = W1Lds<NC, FACTORS>::bytes, pi = (size_t)WALK_MAX_LIST * NC * sizeof(double);
    return (sizeof(WalkShared) + 15) / 16 * 16 + (w1 > pi ? w1 : pi);
}

template <int NC, bool FORWARD, bool FACTORS>
__global__ void __launch_bounds__(WALK_T) chain_walk_kernel(ChainCommon c, FnDevice fn, ChainWalkArgs w)
{
    extern __shared__ __attribute__((aligned(16))) char walk_lds[];
    WalkShared* const ws = reinterpret_cast<WalkShared*>(walk_lds);
    char* const w1lds = walk_lds + (sizeof(WalkShared) + 15) / 16 * 16;
    double* const pi = reinterpret_cast<double*>(w1lds);
    const int tid = threadIdx.x;
    const int nb = w.n_bonds;
    unsigned long long ph[3] = {0ull, 0ull, 0ull};
    const unsigned long long ph_t0 = w.phase_ticks ? wall_clock64() : 0ull;
    unsigned long long ph_last = ph_t0;
    auto phase = [&](int slot) {
        if (w.phase_ticks) {
            const unsigned long long now = wall_clock64();
            ph[slot] += now - ph_last;
            ph_last = now;
        }
    };
    // the walk's own view of the chain constants: dimensions, sizes of the independent lists and the dependent list live in the LDS
    ChainCommon cw = c;
    cw.dims = ws->dims;
    cw.hdims = nullptr;
    cw.ind_cnt = ws->ind_cnt;
    cw.dep_code = ws->dep_code;
    cw.dep_acc = ws->dep_acc;
    cw.dep_cap = WALK_MAX_LIST;
    for (int e = tid; e < nb * 4; e += (int)blockDim.x) ws->dims[e] = 0;
    for (int e = tid; e < nb; e += (int)blockDim.x) ws->ind_cnt[e] = c.ind_cnt[e];
    for (int e = tid; e <= nb; e += (int)blockDim.x) { // (nb + 1 sites)
        ws->ldim[e] = c.ldim[e];
        ws->woff[e] = c.woff[e];
    }
    cw.ldim = ws->ldim;
    cw.woff = ws->woff;
    {   // the functor's weights (K rows of `total` entries), when they fit
        const int nw = c.K * c.total;
        if (nw <= WALK_MAX_W) {
            for (int e = tid; e < nw; e += (int)blockDim.x) ws->w[e] = c.w[e];
            cw.w = ws->w;
        }
    }
    walk_phase_barrier();
    walk_prefetch(c, ws, FORWARD ? 0 : nb - 1, 0, (int)blockDim.x); // (the first bond's static inputs: nobody could fetch them ahead)
    walk_phase_barrier();
    for (int k = 0; k <= nb; ++k) {
        const int b = FORWARD ? k : nb - 1 - k;
        ChainPrepArgs pa;
        pa.b = b;
        pa.do_build = k < nb ? 1 : 0; // (behind the last bond: its pivots only)
        pa.with_rowmap = 0;
        pa.prev_b = -1;
        pa.prev_iresult = nullptr;
        pa.prev_rowperm = nullptr;
        pa.prev_colperm = nullptr;
        pa.prev_token = 0u;
        pa.dbg = nullptr;
        pa.defer_host_writes = 1;
        if (k > 0) {
            pa.prev_b = FORWARD ? b - 1 : b + 1;
            pa.prev_iresult = ws->prev_ires;
            pa.prev_rowperm = ws->perm_rp;
            pa.prev_colperm = ws->perm_cp;
            pa.prev_token = w.token_base + (unsigned)(k - 1);
        }
        if (w.lean_prep) {
            if (tid < 64) walk_prep_wave0<FORWARD>(cw, ws, b, pa.prev_b, pa.prev_token, k < nb, w.phase_ticks);
        } else {
            chain_prep_body(cw, pa, w.phase_ticks, ws);
        }
        walk_phase_barrier(!w.lean_prep);
        phase(0);
        if (k == nb) break;
        // ---- the candidate matrix: pi[j * nd + i] = f(dependent i, independent j) ----
        const int d0 = ws->dims[b * 4], d1 = ws->dims[b * 4 + 1], poisoned = ws->dims[b * 4 + 2];
        const int nd = FORWARD ? d0 : d1, ni = FORWARD ? d1 : d0;
        const bool runs = poisoned == 0 && nd > 0 && ni > 0 && nd <= WALK_MAX_LIST && ni <= NC;
        if (runs) {
            const int K = fn.n_acc;
            const uint64_t* ia = ws->ind_acc[b & 1];
            const float rnd = 1.0f / (float)nd;
            for (int idx = tid; idx < nd * ni; idx += (int)blockDim.x) {
                const int j = walk_div_small(idx, rnd), i = idx - j * nd; // (nd <= 64, idx < 64 * 32)
                uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
                for (int q = 0; q < K; ++q) acc[q] = ws->dep_acc[(size_t)i * K + q] + ia[(size_t)j * K + q];
                pi[idx] = t4a_fn_value(fn.fid, acc, fn.params);
            }
        }
        walk_phase_barrier(!w.lean_prep);
        phase(1);
        // ---- the factorisation: wave 0 (rows = the dependent side in both directions); the other waves fetch the next bond's
        // static inputs meanwhile (its slot held the previous bond's lists: the gather of this bond was their last reader) ----
        if (tid >= 64 && k + 1 < nb) walk_prefetch(c, ws, FORWARD ? b + 1 : b - 1, 64, (int)blockDim.x - 64);
        if (tid < 64) {
            int npiv = -2;
            char* blk = w.blocks + (size_t)b * w.block_bytes;
            if (runs) {
                RrluXcdArgs a = {};
                a.A = pi;
                a.Aout = FACTORS ? w.factors + (size_t)b * w.factors_stride : nullptr;
                a.urows = nullptr;
                a.M = nd;
                a.N = ni;
                a.max_steps = w.max_steps < (nd < ni ? nd : ni) ? w.max_steps : (nd < ni ? nd : ni);
                a.rel_tol = w.rel_tol;
                a.abs_tol = w.abs_tol;
                a.tie_row_major = FORWARD ? 0 : 1;
                a.out_transposed = FORWARD ? 0 : 1;
                a.W = 1;
                a.xcc = 0;
                a.ticket = nullptr;
                a.ticket_base = 0u;
                a.row_perm = FORWARD ? ws->perm_rp : ws->perm_cp; // (into the LDS: the next preparation reads them there)
                a.col_perm = FORWARD ? ws->perm_cp : ws->perm_rp;
                a.iresult = reinterpret_cast<int*>(blk + 16);
                a.dresult = reinterpret_cast<double*>(blk);
                a.pivot_vals = reinterpret_cast<double*>(blk + w.off_piv);
                a.keys = nullptr;
                a.salt = w.token_base + (unsigned)k;
                a.spec_frac = 0.0;
                a.stamps = nullptr;
                a.h_block = nullptr;
                a.block_u64 = 0;
                a.dims = nullptr; // (the dimensions are known here: no device-side read, the token is written below)
                a.dims_swap = 0;
                a.rowmap = nullptr;
                a.ts_u64 = 0;
                const unsigned long long t0 = w.timed ? wall_clock64() : 0ull;
                npiv = rrlu_w1_body<NC, !FORWARD, FACTORS>(a, w1lds);
                if (npiv >= 0) {
                    // the host's copy of the permutations (and what build_factors_from reads), the completion token
                    int* const g_rp = reinterpret_cast<int*>(blk + w.off_rp);
                    int* const g_cp = reinterpret_cast<int*>(blk + w.off_cp);
                    if (tid < d0) g_rp[tid] = ws->perm_rp[tid];
                    if (tid < d1) g_cp[tid] = ws->perm_cp[tid];
                    if (w.timed && tid == 0) {
                        unsigned long long* ts = reinterpret_cast<unsigned long long*>(blk + w.off_ts);
                        ts[0] = t0;
                        ts[1] = wall_clock64();
                    }
                    if (tid == 0) a.iresult[3] = (int)a.salt; // (the host reads the block behind the end of the kernel)
                }
            }
            if (tid == 0) { // what the next preparation checks
                ws->prev_ires[0] = npiv >= 0 ? npiv : 0;
                ws->prev_ires[1] = npiv >= 0 ? 0 : 1;
                ws->prev_ires[2] = 0;
                ws->prev_ires[3] = npiv >= 0 ? (int)(w.token_base + (unsigned)k) : 0;
            }
        }
        walk_phase_barrier(!w.lean_prep);
        phase(2);
    }
    // ---- the host's copies, in bulk (chain_mirror_body): tables into the pinned mirror, dimensions into global memory and hdims ----
    for (int e = tid; e < nb * 4; e += (int)blockDim.x) c.dims[e] = ws->dims[e];
    __threadfence(); // (the copy below reads counts and dimensions with device-scope loads: every store of the walk is out first)
    __syncthreads();
    chain_mirror_body(c, nb);
    if (w.phase_ticks && tid == 0) {
        for (int i = 0; i < 3; ++i) w.phase_ticks[i] = ph[i];
        w.phase_ticks[3] = wall_clock64() - ph_t0;
    }
}

template <int NC, bool FORWARD, bool FACTORS> void chain_walk_launch_one(const ChainCommon& c, const FnDevice& fn, const ChainWalkArgs& w, hipStream_t stream)
{
    constexpr size_t lds = walk_lds_bytes<NC, FACTORS>();
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        // (static LDS of the preparation body + this must stay within the 160 KiB of a compute unit: ask for what is needed)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk_kernel<NC, FORWARD, FACTORS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((chain_walk_kernel<NC, FORWARD, FACTORS>), dim3(1), dim3(WALK_T), lds, stream, c, fn, w);
}

template <int NC> void chain_walk_launch_nc(const ChainCommon& c, const FnDevice& fn, const ChainWalkArgs& w, hipStream_t stream)
{
    if (w.factors) {
        if (c.forward) chain_walk_launch_one<NC, true, true>(c, fn, w, stream);
        else chain_walk_launch_one<NC, false, true>(c, fn, w, stream);
    } else {
        if (c.forward) chain_walk_launch_one<NC, true, false>(c, fn, w, stream);
        else chain_walk_launch_one<NC, false, false>(c, fn, w, stream);
    }
}

} // namespace

void chain_indep_launch(const ChainCommon& c, int n_bonds, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_indep_kernel, dim3(n_bonds), dim3(CHAIN_T), 0, stream, c);
}

void chain_prep_launch(const ChainCommon& c, const ChainPrepArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_prep_kernel, dim3(1), dim3(CHAIN_T), 0, stream, c, a);
}

void chain_pi_launch(const ChainCommon& c, const FnDevice& fn, int b, int n_dep_ub, int n_ind_ub, double* out, hipStream_t stream)
{
    if (n_dep_ub <= 0 || n_ind_ub <= 0) return;
    const int gy = n_ind_ub < 2048 ? n_ind_ub : 2048;
    hipLaunchKernelGGL(chain_pi_kernel, dim3((n_dep_ub + 255) / 256, gy), dim3(256), 0, stream, c, fn, b, out);
}

void chain_indep_group_launch(const ChainGroupSlot* d_slots, int n_handles, int n_bonds, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_indep_group_kernel, dim3(n_bonds, n_handles), dim3(CHAIN_T), 0, stream, d_slots);
}

void chain_prep_group_launch(const ChainGroupSlot* d_slots, const ChainPrepGroupArgs& a, int n_handles, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_prep_group_kernel, dim3(n_handles), dim3(CHAIN_T), 0, stream, d_slots, a);
}

void chain_pi_group_launch(const ChainGroupSlot* d_slots, int n_handles, int b, int n_dep_ub, int n_ind_ub, hipStream_t stream)
{
    if (n_dep_ub <= 0 || n_ind_ub <= 0 || n_handles <= 0) return;
    const int gy = n_ind_ub < 2048 ? n_ind_ub : 2048;
    hipLaunchKernelGGL(chain_pi_group_kernel, dim3((n_dep_ub + 255) / 256, gy, n_handles), dim3(256), 0, stream, d_slots, b);
}

// columns: the largest independent side of the half-sweep (<= 32: beyond that the one-workgroup kernel of the launched chain is
// faster than one wave); every dependent side <= 64
void chain_walk_launch(const ChainCommon& c, const FnDevice& fn, const ChainWalkArgs& w, int columns, hipStream_t stream)
{
    if (columns <= 8) chain_walk_launch_nc<8>(c, fn, w, stream);
    else if (columns <= 16) chain_walk_launch_nc<16>(c, fn, w, stream);
    else chain_walk_launch_nc<32>(c, fn, w, stream);
}

void chain_last_core_launch(const ChainCommon& c, const FnDevice& fn, double* core, int max_entries, hipStream_t stream)
{
    const int blocks = max_entries > 256 * 64 ? 64 : (max_entries + 255) / 256;
    hipLaunchKernelGGL(chain_last_core_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, c, fn, core, max_entries);
}

void chain_mirror_launch(const ChainCommon& c, int n_bonds, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_mirror_kernel, dim3(8), dim3(1024), 0, stream, c, n_bonds);
}

} // namespace t4a
