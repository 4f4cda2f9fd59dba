// kernels_small.hip — the small-problem engine: optimize_with_finder (tensorci2.rs:1626-1802) of a small TensorCI2 problem as ONE
// launch ("whole-optimize persistent launch + a register-resident tiny-rank bond path").
//
// What runs inside the launch, in the reference's order:
//   per iteration (tensorci2.rs:1659-1776): normalisation, sweep direction, the extras of this iteration = the sets at the start of
//   the previous one (:1675-1685), history snapshot (:1686-1689), the update_pivots chain over all bonds (:1695-1725, :1821-2007:
//   kronecker_i / kronecker_j :1224-1246, order-preserving union with the extras :1833-1846, candidate matrix :1859-1893 with
//   max_sample_value :2009-2014, full-pivot rrLU matrixlu.rs:735-819, pivots -> I_{b+1}, J_b through non_empty_or_first :1813-1819,
//   bond_errors), fill_site_tensors (:1065-1186: Pi1, P, zero-pivot-matrix guard :1154-1157, the solve :1160-1164), the error of the
//   iteration, convergence_criterion (:1407-1437); then the final 1-site sweep (:1781-1794, sweep1site :865-1050) whose LUCI left
//   factors (matrix_luci.rs:206-229) become the site tensors, and the last site's tensor (:813-850).
//
// Data: the index sets are tables of (code, accumulators) in the LDS — the current sets and three rotating snapshots (history);
// code(child) = digit + d * code(parent) as in kernels_chain.hip.  The candidate matrix of a bond lives in the registers of the
// wavefront: position p = i + MR * j (column-major) sits in lane p % 64, register p / 64; MR = 8 / 16 / 32 for matrices up to
// 8 x 8 (ONE entry per lane) / 16 x 16 / 32 x 32.  A pivot step needs no LDS round trip and no barrier: the arg-max is a DPP
// reduction over the high words (exact sweep on (v * v, position) for ties, zeros and subnormals), the pivot column and row reach
// the lanes through ds_bpermute, the quotient is the bitwise IEEE one (refined reciprocal, kernels_rrlu_xcd_common.hpp), the
// rank-1 update is separately rounded (matrixlu.rs:593-612), the reference's swaps are position tables in registers (they only
// matter for the tie order).  Bit-exact against rrlu_mut like every other rrLU kernel of this library.
//
// Not handled here (the kernel hands the state at the start of the iteration back, status 2, and tci2_small.hip continues on the
// general path): a list beyond SMALL_CAP entries, a candidate matrix beyond 32 x 32, non-finite values, a singular fill.
#include "kernels_rrlu_xcd_common.hpp"

namespace t4a {

namespace {

constexpr int SC = SMALL_CAP;
constexpr int SMALL_WAVES = 4;   // wave 0 walks the bonds, waves 1..3 run fill_site_tensors of the previous iteration beside it
constexpr int SMALL_WORKERS = SMALL_WAVES - 1;

// Everything lives in STATIC LDS at compile-time addresses (pointers and layout offsets kept in scalar registers made the first version
// of this kernel spill scalar registers into vector lanes around every phase); accumulator arrays are laid out for two accumulators.
constexpr int KS = 2;                      // accumulator stride of every table and list
constexpr int NS = SMALL_MAX_SITES;
struct SmallTab {                          // one copy of the index sets: I sets then J sets
    int cnt[2 * NS];
    uint64_t code[2 * NS * SC];
    uint64_t acc[2 * NS * SC * KS];
};
struct SmallWork {                         // fill_site_tensors workspace of one wave
    double As[SC * SC];
    double Bs[SC * 64];
    uint64_t fl[(64 + 2 * SC) * KS];
};
struct SmallStatic {
    SmallTab tab[3];             // [0] the current sets, [1], [2] the history snapshots (iteration t: slot 1 + t % 2)
    uint64_t lcode[64];          // the two lists of the bond in flight: rows 0..31, columns 32..63
    uint64_t lacc[64 * KS];
    uint64_t w[SMALL_MAX_W];
    int ldim[NS], woff[NS];
    double bond[NS];
    int shapes[3 * NS], cdims[3 * NS];
    double err[SMALL_MAX_ITER];
    int rank[SMALL_MAX_ITER];
    double pe[SMALL_TILE + 2];
    double Af[SMALL_TILE * SMALL_TILE]; // candidate matrix on its way into the registers (E > 1) / factored matrix of a 1-site bond
    double xs[SC * 64];
    int pp[64];
    int o_ptab[64], o_rpos[64]; // results of an out-of-line bond (tiles beyond 8 x 8)
    double o_pvabs[64], o_error, o_am; // (o_am: max sqrt(v * v) the bond sampled)
    SmallWork wk[SMALL_WAVES];
    unsigned long long ph[8], ph_last;
    double params[16];
    double msv;                  // max_sample_value
    int fid, n, total, n_pe, stamps, rook, rook_bonds, rook_visits;
    // fill jobs: wave 0 publishes job number e (fill of iteration e - 1 from snapshot slot 1 + e % 2) by storing e; every worker
    // answers in done[w]; fail[w] != 0: a site could not be filled
    int job, quit, done[SMALL_WAVES], fail[SMALL_WAVES];
};
__shared__ SmallStatic SH;

// The scalar stage of the built-in functor (include/t4a_testfunctions.h), ONE copy per kernel: inlined at every evaluation site the
// four function bodies were most of a 200 000-line kernel.
__device__ __attribute__((noinline)) double small_fn_value(uint64_t a0, uint64_t a1)
{
    uint64_t acc[T4A_FN_MAX_ACC] = {a0, a1, 0, 0};
    double p[T4A_FN_MAX_PARAMS];
#pragma unroll
    for (int q = 0; q < T4A_FN_MAX_PARAMS; ++q) p[q] = SH.params[q];
    return t4a_fn_value(__builtin_amdgcn_readfirstlane(SH.fid), acc, p);
}

__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int ui(int v) { return __builtin_amdgcn_readfirstlane(v); }
// x / d for 0 <= x < 8192, 1 <= d <= 128 (kernels_chain.hip: walk_div_small, checked exhaustively on the host)
__device__ __forceinline__ int div_small(int x, int d) { return (int)(((float)x + 0.5f) * (1.0f / (float)d)); }
__device__ __forceinline__ double bperm_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)lo32(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)hi32(v));
    return mk_f64((unsigned)lo, (unsigned)hi);
}
__device__ __forceinline__ uint64_t bperm_u64(uint64_t v, int src_lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane_s)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane_s);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane_s);
    return ((uint64_t)hi << 32) | lo;
}
// x / p, bitwise the IEEE quotient: the refined-reciprocal form (kernels_rrlu_xcd_common.hpp) in the common case, the full division
// sequence behind a wave-uniform branch (as plain selects the compiler evaluated both on every step)
__device__ __forceinline__ double small_div(double x, double p, double rp, bool p_mid)
{
    const double q0 = x * rp;
    const double qf = __builtin_fma(__builtin_fma(-p, q0, x), rp, q0);
    double q = (x == 0.0) ? q0 : qf;
    const bool slow = !(p_mid && (exp_mid(x) || x == 0.0));
    if (__ballot(slow) != 0ull) {
        asm volatile("; full division" ::: "memory");
        if (slow) q = x / p;
    }
    return q;
}
__device__ __forceinline__ unsigned long long small_clock() { return __builtin_amdgcn_s_memrealtime(); }

// (`on` comes from the kernel arguments: as a flag in the LDS every call site cost the walking wave an LDS round trip)
__device__ __forceinline__ void small_stamp(bool on, int slot)
{
    if (on) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) {
            SH.ph[slot] += now - SH.ph_last;
            SH.ph_last = now;
        }
    }
}

// ---- the two lists of a bond: rows in list slots 0..31, columns in 32..63; lanes 0..31 build the rows, 32..63 the columns -----------
// 2-site: rows = kron(I_b, d_b) + extras(H.I[b+1]), columns = kron(J_{b+1}, d_{b+1}) + extras(H.J[b]) (tensorci2.rs:1224-1246, :1833-1846);
// 1-site forward (:918-1050): rows = kron(I_b, d_b), columns = J_b itself.
// Two LDS round trips: every load whose address does not depend on loaded data first (counts, site data, the parents' and the extras'
// entries by lane), then the parent of each child through the cross-lane network and the digit's weight.
template <int K>
__device__ __forceinline__ bool small_lists(const int n, const int total, const int tile_max, int hist, bool use_hist, int b, bool one_site, int& M, int& N)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
    const bool direct = one_site && half;
    const int ps = half ? (direct ? b : b + 1) : b;
    const int hsite = half ? b : b + 1;
    const int tt = t < SC ? t : 0;
    const bool extras = use_hist && !one_site;
    const int fp = half * n + ps, fh = half * n + hsite;
    const int np = SH.tab[0].cnt[fp];
    const int d = direct ? 1 : SH.ldim[ps], wo = SH.woff[ps];
    const uint64_t mypc = SH.tab[0].code[fp * SC + tt];
    uint64_t mypa[K], xa[K];
#pragma unroll
    for (int q = 0; q < K; ++q) mypa[q] = SH.tab[0].acc[(fp * SC + tt) * KS + q];
    int ne = 0;
    uint64_t xc = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) xa[q] = 0;
    if (extras) {
        ne = SH.tab[hist].cnt[fh];
        xc = SH.tab[hist].code[fh * SC + tt];
#pragma unroll
        for (int q = 0; q < K; ++q) xa[q] = SH.tab[hist].acc[(fh * SC + tt) * KS + q];
    }
    const int m0 = np * d;
    bool ok = np >= 1 && np <= SC && m0 <= tile_max && ne >= 0 && ne <= SC;
    int parent = 0, digit = 0;
    if (ok && t < m0) {
        if (half) { // (digit outer, parent inner)
            digit = div_small(t, np);
            parent = t - digit * np;
        } else {    // (parent outer, digit inner)
            parent = div_small(t, d);
            digit = t - parent * d;
        }
    }
    const uint64_t pc = bperm_u64(mypc, half * 32 + parent);
    uint64_t acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const uint64_t pa = bperm_u64(mypa[q], half * 32 + parent);
        acc[q] = pa + (direct ? 0ull : SH.w[q * total + wo + digit]);
    }
    const uint64_t code = direct ? pc : (uint64_t)digit + (uint64_t)d * pc;
    bool keep = extras && ok && t < ne;
    if (extras) { // an extra whose parent is among the parents is already in the Kronecker part
        const uint64_t mybase = mypc * (uint64_t)d;
        const int np0 = __builtin_amdgcn_readlane(np, 0), np1 = __builtin_amdgcn_readlane(np, 32);
        const int npmax = np0 > np1 ? np0 : np1;
        for (int pi = 0; pi < npmax && pi < SC; ++pi) {
            const uint64_t b0 = readlane_u64(mybase, pi), b1 = readlane_u64(mybase, 32 + pi);
            const uint64_t base = half ? b1 : b0;
            if (pi < np && (xc - base) < (uint64_t)d) keep = false;
        }
    }
    const unsigned long long km = __ballot(keep);
    const unsigned kmh = half ? (unsigned)(km >> 32) : (unsigned)km;
    const int nkeep = __builtin_popcount(kmh);
    const int pos = m0 + __builtin_popcount(kmh & ((1u << t) - 1u));
    const int tot = m0 + nkeep;
    ok = ok && tot <= tile_max;
    if (__ballot(!ok) != 0ull) return false;
    if (t < m0) {
        SH.lcode[half * 32 + t] = code;
#pragma unroll
        for (int q = 0; q < K; ++q) SH.lacc[(half * 32 + t) * KS + q] = acc[q];
    }
    if (keep) {
        SH.lcode[half * 32 + pos] = xc;
#pragma unroll
        for (int q = 0; q < K; ++q) SH.lacc[(half * 32 + pos) * KS + q] = xa[q];
    }
    M = __builtin_amdgcn_readlane(tot, 0);
    N = __builtin_amdgcn_readlane(tot, 32);
    wsync();
    return true;
}

// ---- candidate matrix + full-pivot rrLU in registers ---------------------------------------------------------------------------------
// E entries per lane: position p = lane + 64 e = i + MR j.  Returns the number of pivots (-1: non-finite values).  Outputs:
//   ptab: lane k = pivot row of step k, lane 32 + k = pivot column of step k;  pvabs: lane k = sqrt(pivot * pivot) of step k;
//   error = RrLU::error (matrixlu.rs:758, :811);  rpos_out: position of row `lane` in the reference's permuted order;
//   with `fact` the factored matrix (L scaled below the pivots in the pivot columns, U in the pivot rows) goes to SH.Af[i + MR j].
template <int K, int E, bool LEFT>
__device__ __forceinline__ int small_bond(const bool stamps, const bool fact, int M, int N, int max_bond_dim, double rel_tol, double abs_tol, double& am_out,
                                          int& ptab, double& pvabs, double& error_out, int& rpos_out)
{
    constexpr int MR = E == 1 ? 8 : (E == 4 ? 16 : 32);
    constexpr int LOG_MR = E == 1 ? 3 : (E == 4 ? 4 : 5);
    constexpr int LPC = 64 / MR; // columns per register plane
    const int lane = threadIdx.x & 63;
    const int i = lane & (MR - 1), jb = lane >> LOG_MR;
    double a[E];
    int cpos[E];
    unsigned inb = 0u;
    // ---- Pi[i, j] = f(row i + column j) (tensorci2.rs:1859-1893), max sqrt(v * v) (:2009-2014) ----
    {
        uint64_t racc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) racc[q] = SH.lacc[(i < M ? i : 0) * KS + q];
        double amax = 0.0;
        bool bad = false;
        if (E > 1) { // (rolled: one call site of the functor; the values pass through the LDS block of the factored matrix)
#pragma unroll 1
            for (int e = 0; e < E; ++e) {
                const int j = jb + LPC * e;
                const bool in = i < M && j < N;
                double v = 0.0;
                if (__ballot(in) != 0ull) {
                    uint64_t acc[2] = {0, 0};
#pragma unroll
                    for (int q = 0; q < K; ++q) acc[q] = racc[q] + SH.lacc[(32 + (j < N ? j : 0)) * KS + q];
                    v = small_fn_value(acc[0], acc[1]);
                }
                SH.Af[lane + 64 * e] = v;
            }
            wsync();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = jb + LPC * e;
            cpos[e] = j;
            const bool in = i < M && j < N;
            double v;
            if (E == 1) {
                uint64_t acc[2] = {0, 0};
#pragma unroll
                for (int q = 0; q < K; ++q) acc[q] = racc[q] + SH.lacc[(32 + (j < N ? j : 0)) * KS + q];
                v = small_fn_value(acc[0], acc[1]);
            } else {
                v = SH.Af[lane + 64 * e];
            }
            v = in ? v : 0.0;
            if (in) inb |= 1u << e;
            bad |= !((v - v) == 0.0);
            amax = vmax_abs(amax, v);
            a[e] = v;
        }
        if (__ballot(bad) != 0ull) return -1;
        const double m = wave_max_f64(amax);
        const double am = hi_mid((int)hi32(m)) ? m : uniform_f64(sqrt(m * m));
        am_out = am;
    }
    small_stamp(stamps, 1);
    int rpos = i;
    int npiv = 0;
    double max_error = 0.0, error = __builtin_nan("");
    const int mn = M < N ? M : N;
    const int max_steps = max_bond_dim < mn ? max_bond_dim : mn;
    const double min_pivot_abs = (rel_tol == 0.0 && abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    ptab = 0;
    pvabs = 0.0;
    bool gave_up = false;
    for (int kn = 0; kn < max_steps; ++kn) {
        // ---- the pivot: first strict maximum of v * v in column-major order of the permuted trailing block (matrixlu.rs:480-519) ----
        int mhi = -1;
        int cnt_l = 0, be = 0;
        double xa = 0.0;
        const bool rowlive = rpos >= kn;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool live = ((inb >> e) & 1u) && rowlive && cpos[e] >= kn;
            const int h = live ? (int)(hi32(a[e]) & 0x7FFFFFFFu) : -1;
            if (h > mhi) {
                mhi = h;
                cnt_l = 1;
                be = e;
                xa = a[e];
            } else if (h == mhi && h >= 0) {
                cnt_l += 1;
            }
        }
        const int whi = wave_max_i32(mhi);
        if (whi < 0) break; // (cannot happen while kn < min(M, N))
        if (((whi >> 20) & 0x7FF) == 0x7FF) {
            gave_up = true;
            break;
        }
        int hl, es;
        double wval, pivot_abs;
        const unsigned long long at_max = __ballot(mhi == whi);
        const unsigned long long multi = E == 1 ? 0ull : __ballot(mhi == whi && cnt_l > 1);
        if (hi_mid(whi) && multi == 0ull && __builtin_popcountll(at_max) == 1) { // distinct high words: distinct squares, all normal
            hl = (int)__builtin_ctzll(at_max);
            es = E == 1 ? 0 : __builtin_amdgcn_readlane(be, hl);
            wval = readlane_f64(xa, hl);
            pivot_abs = __builtin_fabs(wval);
        } else {
            double bs = -1.0;
            unsigned bp = XNOPOS;
            be = 0;
            xa = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool live = ((inb >> e) & 1u) && rowlive && cpos[e] >= kn;
                const double sq = a[e] * a[e];
                const unsigned pos = ((unsigned)cpos[e] << 10) | (unsigned)rpos;
                if (live && (sq > bs || (sq == bs && pos < bp))) {
                    bs = sq;
                    bp = pos;
                    be = e;
                    xa = a[e];
                }
            }
            const double wsq = wave_max_f64(bs);
            const unsigned mine = (bs == wsq) ? bp : XNOPOS;
            const unsigned wp = (unsigned)ui((int)wave_min_u32(mine));
            if (wp == XNOPOS) break;
            hl = (int)__builtin_ctzll(__ballot(mine == wp));
            es = E == 1 ? 0 : __builtin_amdgcn_readlane(be, hl);
            wval = readlane_f64(xa, hl);
            pivot_abs = uniform_f64(sqrt(wval * wval));
        }
        // ---- stop rules, in the reference's order (matrixlu.rs:757-781) ----
        error = pivot_abs;
        if (kn > 0 && (pivot_abs < rel_tol * max_error || pivot_abs < abs_tol)) break;
        if (pivot_abs <= min_pivot_abs) break;
        max_error = fmax(max_error, pivot_abs);
        const int pr = hl & (MR - 1), pc = (hl >> LOG_MR) + LPC * es;
        if ((lane & 31) == kn) {
            ptab = (lane >> 5) ? pc : pr;
            pvabs = pivot_abs;
        }
        // ---- pivot column to the lanes of every row, pivot row to the lanes of every column ----
        double asel = a[0];
#pragma unroll
        for (int e = 1; e < E; ++e)
            if (es == e) asel = a[e];
        const double xcol = bperm_f64(asel, i + MR * (hl >> LOG_MR));
        double urow[E];
#pragma unroll
        for (int e = 0; e < E; ++e) urow[e] = bperm_f64(a[e], pr + MR * jb);
        const int prp = __builtin_amdgcn_readlane(rpos, hl);
        int csel = cpos[0];
#pragma unroll
        for (int e = 1; e < E; ++e)
            if (es == e) csel = cpos[e];
        const int pcp = __builtin_amdgcn_readlane(csel, hl);
        const double rp = refined_rcp(wval);
        const bool pmid = exp_mid(wval);
        // scale_column_tail (:562-577) / scale_row_tail (:579-591): the bitwise IEEE quotient
        double l = xcol;
        if (LEFT) l = small_div(xcol, wval, rp, pmid);
        const bool row_rest = rowlive && i != pr; // rows behind the new pivot row in the permuted order
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = jb + LPC * e;
            const bool in = (inb >> e) & 1u;
            const bool col_rest = cpos[e] >= kn && j != pc;
            double u = urow[e];
            if (!LEFT) u = small_div(urow[e], wval, rp, pmid);
            const double prod = l * u;
            const double upd = a[e] - prod; // separately rounded (matrixlu.rs:593-612)
            double nv = a[e];
            if (in && row_rest && col_rest) nv = upd;
            if (LEFT && in && row_rest && j == pc) nv = l;
            if (!LEFT && in && col_rest && i == pr) nv = u;
            a[e] = nv;
            // swap_cols as a position table (:541): the column at position kn takes the pivot column's old position
            if (cpos[e] == kn) cpos[e] = pcp;
            else if (j == pc) cpos[e] = kn;
        }
        if (rpos == kn) rpos = prp;
        else if (i == pr) rpos = kn;
        npiv = kn + 1;
    }
    if (gave_up) return -1;
    if (npiv >= mn) error = 0.0; // matrixlu.rs:811-813
    error_out = error;
    rpos_out = rpos;
    if (fact) {
#pragma unroll
        for (int e = 0; e < E; ++e) SH.Af[lane + 64 * e] = a[e];
    }
    return npiv;
}

// Tiles beyond 8 x 8 run out of line (their register needs — sixteen entries, positions and masks per lane — stay out of the
// allocation of the 8 x 8 path, which is the one a small problem lives on); per-lane results travel through the LDS.
template <int K, int E, bool LEFT>
__device__ __attribute__((noinline)) int small_bond_big(int fact_, int M_, int N_, int max_bond_dim_, double rel_tol, double abs_tol)
{
    int ptab, rpos;
    double pvabs, error = 0.0, am = 0.0;
    const int r = small_bond<K, E, LEFT>(false, ui(fact_) != 0, ui(M_), ui(N_), ui(max_bond_dim_), uniform_f64(rel_tol), uniform_f64(abs_tol), am, ptab, pvabs, error, rpos);
    const int lane = threadIdx.x & 63;
    if (r >= 0) {
        SH.o_ptab[lane] = ptab;
        SH.o_rpos[lane] = rpos;
        SH.o_pvabs[lane] = pvabs;
        if (lane == 0) {
            SH.o_error = error;
            SH.o_am = am;
        }
    }
    wsync();
    return r;
}

// ---- PivotSearchStrategy::Rook (tensorci2.rs:1904-1929; matrixluci/block_rook.rs:71-190) for a candidate matrix that fits the tile ----
// The lazy evaluator exists to avoid evaluating f; a built-in functor costs nothing, so the matrix is materialised in the LDS (SH.Af,
// leading dimension 32) and ONE wavefront runs factorize_lazy on it with the arithmetic of rook_dense_kernel (rook.hip), operation for
// operation: per accepted pivot the LU of the k x k pivot block with partial pivoting (first maximum of |a_ik|) and X = P^-1 A[I, :]
// by column-oriented substitutions (separately rounded multiply and subtract), the alternating column / row arg-max of the residual
// with j-ascending sums and first strict maxima.  max_sample_value only sees the rows and columns the lazy evaluator would have
// visited (tensorci2.rs:2035-2142).  Returns the rank, -1 for non-finite values, -2 for a singular pivot block (handed back).
// Results: SH.o_ptab (lane k: row of pivot k, lane 32 + k: column), SH.o_pvabs (accepted pivot errors), SH.o_error (pivot_errors.back()).
template <int K>
__device__ __attribute__((noinline)) int small_bond_rook(int M_, int N_, int max_bond_dim_, double rel_tol, double abs_tol)
{
    const int lane = threadIdx.x & 63;
    const int M = ui(M_), N = ui(N_);
    rel_tol = uniform_f64(rel_tol);
    abs_tol = uniform_f64(abs_tol);
    constexpr int LD = SMALL_TILE; // leading dimension of A in the LDS
    double* const A = SH.Af;
    // ---- Pi, compactly: entry p = i + M j of the M N entries in lane p % 64 of pass p / 64 ----
    {
        bool bad = false;
        const int total = M * N;
        for (int p0 = 0; p0 < total; p0 += 64) {
            const int p = p0 + lane;
            const bool in = p < total;
            const int j = div_small(in ? p : 0, M), i = (in ? p : 0) - j * M;
            uint64_t acc[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = SH.lacc[i * KS + q] + SH.lacc[(32 + j) * KS + q];
            const double v = small_fn_value(acc[0], acc[1]);
            if (in) {
                A[i + LD * j] = v;
                bad |= !((v - v) == 0.0);
            }
        }
        if (__ballot(bad) != 0ull) return -1;
        wsync();
    }
    double* const P = SH.wk[0].As;   // k x k, leading dimension SC
    double* const X = SH.wk[0].Bs;   // k x N, leading dimension SC
    double* const bv = SH.xs;        // k
    double* const yv = SH.xs + 64;   // k
    int* const I = SH.pp;            // selected rows
    int* const J = SH.pp + 16;       // selected columns
    int* const piv = SH.pp + 32;
    const int full_rank = M < N ? M : N;
    const int max_bond = ui(max_bond_dim_) < full_rank ? ui(max_bond_dim_) : full_rank;
    unsigned rowsel = 0u, colsel = 0u, seen_r = 0u, seen_c = 0u;
    const unsigned rmask = M >= 32 ? 0xFFFFFFFFu : ((1u << M) - 1u), cmask = N >= 32 ? 0xFFFFFFFFu : ((1u << N) - 1u);
    int k = 0, visits = 0;
    double max_error = 0.0, last_error = __builtin_nan("");
    double accepted = 0.0; // lane q: accepted pivot error q
    int ptab = 0;
    while (k < max_bond && k < SC) {
        const unsigned rem_r = ~rowsel & rmask, rem_c = ~colsel & cmask;
        if (rem_r == 0u || rem_c == 0u) break;
        const int first_row = __builtin_ctz(rem_r), first_col = __builtin_ctz(rem_c);
        const int n_rem_rows = __builtin_popcount(rem_r), n_rem_cols = __builtin_popcount(rem_c);
        if (k > 0) {
            // factor_step: P = A[I, J] -> LU in place (row swaps applied to X as well); X = P^-1 A[I, :]
            for (int e = lane; e < k * k; e += 64) {
                const int c = div_small(e, k), r = e - c * k;
                P[c * SC + r] = A[I[r] + LD * J[c]];
            }
            for (int e = lane; e < k * N; e += 64) {
                const int c = div_small(e, k), r = e - c * k;
                X[c * SC + r] = A[I[r] + LD * c];
            }
            wsync();
            int info = 0;
            for (int c = 0; c < k; ++c) {
                double bvv = -1.0;
                int bi = 0x7fffffff;
                if (lane >= c && lane < k) {
                    bvv = fabs(P[c * SC + lane]);
                    bi = lane;
                }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) { // (k <= 16)
                    const double ov = __shfl_xor(bvv, off);
                    const int oi = __shfl_xor(bi, off);
                    if (ov > bvv || (ov == bvv && oi < bi)) {
                        bvv = ov;
                        bi = oi;
                    }
                }
                const int p = __builtin_amdgcn_readlane(bi, 0);
                const double gv = readlane_f64(bvv, 0);
                if (lane == 0) piv[c] = p;
                if (!(gv > 0.0) && info == 0) info = c + 1;
                if (p != c && p < k) {
                    if (lane < k) {
                        const double t0 = P[lane * SC + c];
                        P[lane * SC + c] = P[lane * SC + p];
                        P[lane * SC + p] = t0;
                    }
                    if (lane < N) {
                        const double t0 = X[lane * SC + c];
                        X[lane * SC + c] = X[lane * SC + p];
                        X[lane * SC + p] = t0;
                    }
                }
                wsync();
                const double pv = P[c * SC + c];
                if (pv == 0.0 || pv != pv) continue;
                if (lane > c && lane < k) P[c * SC + lane] = P[c * SC + lane] / pv;
                wsync();
                const int rem = k - c - 1;
                for (int e = lane; e < rem * rem; e += 64) {
                    const int qq = div_small(e, rem);
                    const int i = c + 1 + (e - qq * rem), q = c + 1 + qq;
                    const double prod = P[c * SC + i] * P[q * SC + c];
                    P[q * SC + i] = P[q * SC + i] - prod;
                }
                wsync();
            }
            if (info != 0) return -2;
            // L X' = P_swap A[I, :], then U X = X': a lane per right-hand side, every element sees its updates in trsm_left_kernel's order
            if (lane < N) {
                double* x = X + lane * SC;
                for (int kk = 0; kk < k; ++kk) {
                    const double xk = x[kk];
                    for (int i = kk + 1; i < k; ++i) {
                        const double prod = P[kk * SC + i] * xk;
                        x[i] = x[i] - prod;
                    }
                }
                for (int kk = k - 1; kk >= 0; --kk) {
                    const double xk = x[kk] / P[kk * SC + kk];
                    x[kk] = xk;
                    for (int i = 0; i < kk; ++i) {
                        const double prod = P[kk * SC + i] * xk;
                        x[i] = x[i] - prod;
                    }
                }
            }
            wsync();
        }
        // rook_pivot (:71-118)
        int cur_col = first_col, cur_row = first_row;
        double pivot_abs = 0.0;
        const int max_steps = n_rem_rows + n_rem_cols + 1;
        for (int it = 0; it < max_steps; ++it) {
            ++visits;
            seen_c |= 1u << cur_col;
            if (k > 0) {
                if (lane < k) bv[lane] = A[I[lane] + LD * cur_col];
                wsync();
                if (lane == 0)
                    for (int j = 0; j < k; ++j) {
                        const int p = piv[j];
                        if (p != j) {
                            const double t0 = bv[j];
                            bv[j] = bv[p];
                            bv[p] = t0;
                        }
                    }
                wsync();
                // unit lower, then upper, column-oriented (rook_trsm_vec)
                for (int step = 0; step < k; ++step) {
                    const double bk = bv[step];
                    if (lane > step && lane < k) {
                        const double prod = P[step * SC + lane] * bk;
                        bv[lane] = bv[lane] - prod;
                    }
                    wsync();
                }
                for (int step = k - 1; step >= 0; --step) {
                    if (lane == 0) bv[step] = bv[step] / P[step * SC + step];
                    wsync();
                    const double bk = bv[step];
                    if (lane < step) {
                        const double prod = P[step * SC + lane] * bk;
                        bv[lane] = bv[lane] - prod;
                    }
                    wsync();
                }
            }
            {   // residual of column cur_col over the remaining rows: first maximum of |r|
                double v = -1.0;
                if (lane < M && !((rowsel >> lane) & 1u)) {
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j) {
                        const double prod = A[lane + LD * J[j]] * bv[j];
                        acc = acc + prod;
                    }
                    const double ac = A[lane + LD * cur_col];
                    const double r = k > 0 ? ac - acc : ac;
                    v = fabs(r);
                }
                double g = v;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) g = fmax(g, __shfl_xor(g, off));
                const double gv = readlane_f64(g, 0);
                const unsigned long long hit = __ballot(v == gv && v >= 0.0);
                cur_row = (gv < 0.0 || hit == 0ull) ? first_row : (int)__builtin_ctzll(hit);
            }
            seen_r |= 1u << cur_row;
            if (k > 0) {
                if (lane < k) yv[lane] = A[cur_row + LD * J[lane]];
                wsync();
            }
            int next_col;
            {
                double v = -1.0;
                if (lane < N && !((colsel >> lane) & 1u)) {
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j) {
                        const double prod = yv[j] * X[lane * SC + j];
                        acc = acc + prod;
                    }
                    const double arc = A[cur_row + LD * lane];
                    const double r = k > 0 ? arc - acc : arc;
                    v = fabs(r);
                }
                double g = v;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) g = fmax(g, __shfl_xor(g, off));
                const double gv = readlane_f64(g, 0);
                const unsigned long long hit = __ballot(v == gv && v >= 0.0);
                next_col = (gv < 0.0 || hit == 0ull) ? first_col : (int)__builtin_ctzll(hit);
                pivot_abs = gv < 0.0 ? 0.0 : gv;
            }
            if (next_col == cur_col) break;
            cur_col = next_col;
        }
        // factorize_lazy stop rules (:158-176)
        last_error = pivot_abs;
        if (k > 0 && (pivot_abs < rel_tol * max_error || pivot_abs < abs_tol)) break;
        if (pivot_abs < 2.220446049250313e-16) break;
        max_error = fmax(max_error, pivot_abs);
        if (lane == 0) {
            I[k] = cur_row;
            J[k] = cur_col;
        }
        if (lane == k) {
            accepted = pivot_abs;
            ptab = cur_row;
        }
        if (lane == 32 + k) ptab = cur_col;
        rowsel |= 1u << cur_row;
        colsel |= 1u << cur_col;
        ++k;
        wsync();
    }
    // the pivot block and X are sized for SMALL_CAP pivots: a search that would go on beyond them is not representable here — the
    // iteration is handed back (conservative: the reference might have stopped at the next pivot by its tolerance)
    if (k >= SC && k < max_bond) return -3;
    if (k >= full_rank) last_error = 0.0;                                       // :178-184
    else if (k == max_bond && k > 0) last_error = readlane_f64(accepted, k - 1);
    // what the lazy evaluator looked at: the visited rows and columns (max sqrt(v * v) over them)
    {
        double mx = 0.0;
        for (int p0 = 0; p0 < M * N; p0 += 64) {
            const int p = p0 + lane;
            if (p < M * N) {
                const int j = div_small(p, M), i = p - j * M;
                if (((seen_r >> i) & 1u) || ((seen_c >> j) & 1u)) {
                    const double v = A[i + LD * j];
                    const double av = sqrt(v * v);
                    if (av > mx) mx = av;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        const double am = readlane_f64(mx, 0);
        if (lane == 0) SH.o_am = am;
    }
    SH.o_ptab[lane] = ptab;
    SH.o_rpos[lane] = 0;
    SH.o_pvabs[lane] = accepted;
    if (lane == 0) {
        SH.o_error = last_error;
        SH.rook_bonds += 1;
        SH.rook_visits += visits;
    }
    wsync();
    return k;
}

// pivots -> I_{b+1} (lanes 0..31) and J_b (lanes 32..63) (tensorci2.rs:1934-1940 with non_empty_or_first :1813-1819)
template <int K>
__device__ __forceinline__ bool small_gather(int b, int r, int ptab)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
    const int cnt = r > 0 ? r : 1;
    if (cnt > SC) return false;
    const int site = half ? b : b + 1;
    const int fp = half * SH.n + site;
    if (t < cnt) {
        const int src = half * 32 + (r > 0 ? ptab : 0);
        SH.tab[0].code[fp * SC + t] = SH.lcode[src];
#pragma unroll
        for (int q = 0; q < K; ++q) SH.tab[0].acc[(fp * SC + t) * KS + q] = SH.lacc[src * KS + q];
    }
    if (t == 0) SH.tab[0].cnt[fp] = cnt;
    wsync();
    return true;
}

// One bond of a half-sweep / of the final 1-site sweep.  Returns 0, or the reason the iteration is handed back.
template <int K>
__device__ __forceinline__ int small_update(const int n, const int total, const int tile_max, const bool stamps, const bool rook, const int left_, const int one_, int hist_, int use_hist_,
                                            int b_, int max_bond_dim_, double rel_tol, double abs_tol, double& msv, double* core)
{
    const int lane = threadIdx.x & 63;
    // (an out-of-line function receives its arguments in vector registers: back to scalars)
    const bool left = ui(left_) != 0, one = ui(one_) != 0, use_hist = ui(use_hist_) != 0;
    const int hist = ui(hist_), b = ui(b_), max_bond_dim = ui(max_bond_dim_);
    rel_tol = uniform_f64(rel_tol);
    abs_tol = uniform_f64(abs_tol);
    int M, N;
    if (!small_lists<K>(n, total, tile_max, hist, use_hist, b, one, M, N)) return 1;
    small_stamp(stamps, 0);
    int ptab, rpos, r;
    double pvabs, error, am = 0.0;
    const int tile = (M <= 8 && N <= 8) ? 0 : ((M <= 16 && N <= 16) ? 1 : 2);
    if (!one && rook) {
        r = ui(small_bond_rook<K>(M, N, max_bond_dim, rel_tol, abs_tol));
        if (r == -2) return 5;
        if (r == -3) return 3;
        ptab = SH.o_ptab[lane];
        rpos = 0;
        pvabs = SH.o_pvabs[lane];
        error = uniform_f64(SH.o_error);
        am = uniform_f64(SH.o_am);
    } else if (tile == 0) {
        if (left) r = small_bond<K, 1, true>(stamps, one, M, N, max_bond_dim, rel_tol, abs_tol, am, ptab, pvabs, error, rpos);
        else r = small_bond<K, 1, false>(stamps, one, M, N, max_bond_dim, rel_tol, abs_tol, am, ptab, pvabs, error, rpos);
    } else {
        if (tile == 1) r = left ? small_bond_big<K, 4, true>(one, M, N, max_bond_dim, rel_tol, abs_tol) : small_bond_big<K, 4, false>(one, M, N, max_bond_dim, rel_tol, abs_tol);
        else r = left ? small_bond_big<K, 16, true>(one, M, N, max_bond_dim, rel_tol, abs_tol) : small_bond_big<K, 16, false>(one, M, N, max_bond_dim, rel_tol, abs_tol);
        r = ui(r);
        ptab = SH.o_ptab[lane];
        rpos = SH.o_rpos[lane];
        pvabs = SH.o_pvabs[lane];
        error = uniform_f64(SH.o_error);
        am = uniform_f64(SH.o_am);
    }
    if (r < 0) return 2;
    if (am > msv) msv = am; // update_max_sample_value (tensorci2.rs:2009-2014)
    small_stamp(stamps, 2);
    const int L_b = b == 0 ? 1 : ui(SH.tab[0].cnt[b]); // (before the gather: I_b is not touched by bond b)
    if (!small_gather<K>(b, r, ptab)) return 3;
    if (lane == 0) SH.bond[b] = error; // (pivot_errors.back(), tensorci2.rs:1942-1949 / :1998)
    small_stamp(stamps, 3);
    if (!one) {
        if (lane == 0) {
            SH.shapes[3 * b] = M;
            SH.shapes[3 * b + 1] = N;
            SH.shapes[3 * b + 2] = r;
        }
        return 0;
    }
    // ---- 1-site sweep with update_tensors (tensorci2.rs:991-1017): pivot_errors (:801-808), site tensor b = LUCI left factor ----
    {   // update_pivot_errors(factors.pivot_errors): [sqrt(d * d) of the r pivots] + [error]
        const double mine = lane < r ? pvabs : error;
        const int n_pe = ui(SH.n_pe);
        if (lane <= r) {
            const double old = lane < n_pe ? SH.pe[lane] : 0.0;
            SH.pe[lane] = fmax(old, mine);
        }
        if (r + 1 > n_pe && lane == 0) SH.n_pe = r + 1;
    }
    // left = P_row^T [I ; L21 L11^-1] (matrix_luci.rs:206-229): row at permuted position p < r is the unit vector e_p, the others
    // solve x L11 = l_row with L11 unit lower triangular (the scaled pivot columns of the pivot rows)
    const int MR = tile == 0 ? 8 : (tile == 1 ? 16 : 32);
    if ((lane & 31) < r) SH.pp[lane] = ptab; // pp[k] pivot row k, pp[32 + k] pivot column k
    wsync();
    const int S = SH.ldim[b];
    const int R = r > 0 ? r : 1;
    if (lane < M) {
        if (r == 0) {
            SH.xs[lane] = 0.0;
        } else if (rpos < r) {
            for (int k = 0; k < r; ++k) SH.xs[k * 64 + lane] = (k == rpos) ? 1.0 : 0.0;
        } else {
            for (int k = r - 1; k >= 0; --k) {
                const int pck = SH.pp[32 + k];
                double s = SH.Af[lane + MR * pck];
                for (int t = k + 1; t < r; ++t) {
                    const double prod = SH.xs[t * 64 + lane] * SH.Af[SH.pp[t] + MR * pck];
                    s = s - prod;
                }
                SH.xs[k * 64 + lane] = s;
            }
        }
        // t(l, s, k) = left(l S + s, k), column-major [l, s, k] (tensorci2.rs:994-1004)
        const int l = div_small(lane, S), s_ = lane - l * S;
        for (int k = 0; k < R; ++k) core[l + L_b * (s_ + S * k)] = SH.xs[k * 64 + lane];
    }
    if (lane == 0) {
        SH.cdims[3 * b] = L_b;
        SH.cdims[3 * b + 1] = S;
        SH.cdims[3 * b + 2] = R;
    }
    wsync();
    small_stamp(stamps, 4);
    return 0;
}

// fill_site_tensors of site b (tensorci2.rs:1065-1186) from table copy `tb`, one wavefront, workspace `wk`; the partial-pivot LU and
// the substitutions of fill_small_kernel (kernels_pi.hip), operation for operation.  Returns false: not representable here / singular.
// cdims != nullptr: the tensor's dimensions are recorded.
template <int K>
__device__ __attribute__((noinline)) int small_fill_site(int w_, int tb_, int b_, double* core, int record_dims_)
{
    const int lane = threadIdx.x & 63;
    const int tb = ui(tb_), b = ui(b_);
    SmallWork& wk = SH.wk[ui(w_)];
    int* const cdims = ui(record_dims_) ? SH.cdims : nullptr;
    const int n = ui(SH.n);
    const int Lb = ui(SH.tab[tb].cnt[b]), S = SH.ldim[b], wo = SH.woff[b];
    const int nj = ui(SH.tab[tb].cnt[n + b]);
    const int ni = Lb * S;
    if (ni > 64 || ni < 1 || nj < 1 || nj > SC) return false;
    uint64_t* ka = wk.fl;                 // [64][K] kron(I_b, d_b)
    uint64_t* ja = wk.fl + 64 * KS;        // [SC][K] J_b
    uint64_t* ia = wk.fl + (64 + SC) * KS; // [SC][K] I_{b+1}
    if (lane < ni) {
        const int parent = div_small(lane, S), digit = lane - parent * S;
#pragma unroll
        for (int q = 0; q < K; ++q) ka[lane * KS + q] = SH.tab[tb].acc[(b * SC + parent) * KS + q] + SH.w[q * SH.total + wo + digit];
    }
    if (lane < nj) {
#pragma unroll
        for (int q = 0; q < K; ++q) ja[lane * KS + q] = SH.tab[tb].acc[((n + b) * SC + lane) * KS + q];
    }
    const bool last = b == n - 1;
    const int np = last ? 0 : ui(SH.tab[tb].cnt[b + 1]);
    if (!last) {
        if (np != nj) return false;
        if (lane < np) {
#pragma unroll
            for (int q = 0; q < K; ++q) ia[lane * KS + q] = SH.tab[tb].acc[((b + 1) * SC + lane) * KS + q];
        }
    }
    wsync();
    const int left_dim = b == 0 ? 1 : Lb;
    if (last) { // :1109-1128: t(l, s, 0) = Pi1(l S + s, 0)
        uint64_t acc[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = ka[(lane < ni ? lane : 0) * KS + q] + ja[q];
        const double v = small_fn_value(acc[0], acc[1]);
        if (lane < ni) {
            const int l = div_small(lane, S), s_ = lane - l * S;
            core[l + left_dim * s_] = v;
        }
        if (cdims && lane == 0) {
            cdims[3 * b] = left_dim;
            cdims[3 * b + 1] = S;
            cdims[3 * b + 2] = 1;
        }
        return true;
    }
    const int nn = nj, nrhs = ni;
    double* As = wk.As; // P^T, nn x nn column-major: As[col * nn + i] = f(I_{b+1}[col], J_b[i])
    double* Bs = wk.Bs; // Pi1^T, nn x nrhs:          Bs[col * nn + i] = f(kron[col], J_b[i])
    double pm = 0.0;
    bool bad = false;
    // one pass over both matrices: entry e < nn * nn belongs to P^T, the rest to Pi1^T (ONE call of the functor per 64 entries)
    const int nA = nn * nn, nAll = nA + nn * nrhs;
    for (int e0 = 0; e0 < nAll; e0 += 64) {
        const int e = e0 + lane;
        const bool in = e < nAll;
        const bool isA = e < nA;
        const int eb = in ? (isA ? e : e - nA) : 0;
        const int col = div_small(eb, nn), i = eb - col * nn;
        uint64_t acc[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = (isA ? ia[col * KS + q] : ka[col * KS + q]) + ja[i * KS + q];
        const double v = small_fn_value(acc[0], acc[1]);
        if (in) {
            if (isA) {
                As[eb] = v;
                const double av = sqrt(v * v);
                if (av > pm) pm = av;
            } else {
                Bs[eb] = v;
            }
            bad |= !((v - v) == 0.0);
        }
    }
    if (__ballot(bad) != 0ull) return false;
    wsync();
    const double pmax = wave_max_f64(pm);
    const bool zero = pmax < 2.220446049250313e-16; // every |p| < EPS: zero core (:1154-1157)
    if (!zero) {
        int info = 0;
        for (int col = 0; col < nn; ++col) {
            double bv = -1.0;
            int bi = 0x7fffffff;
            if (lane >= col && lane < nn) {
                bv = fabs(As[col * nn + lane]);
                bi = lane;
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) { // (nn <= 16: lanes 0..15)
                const double ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            const int p = __builtin_amdgcn_readlane(bi, 0);
            const double pvv = readlane_f64(bv, 0);
            if (!(pvv > 0.0) && info == 0) info = col + 1;
            if (p != col && p < nn) {
                if (lane < nn) {
                    const double t0 = As[lane * nn + col];
                    As[lane * nn + col] = As[lane * nn + p];
                    As[lane * nn + p] = t0;
                }
                if (lane < nrhs) {
                    const double t0 = Bs[lane * nn + col];
                    Bs[lane * nn + col] = Bs[lane * nn + p];
                    Bs[lane * nn + p] = t0;
                }
            }
            wsync();
            const double piv = As[col * nn + col];
            if (piv == 0.0 || piv != piv) continue;
            if (lane > col && lane < nn) As[col * nn + lane] = As[col * nn + lane] / piv;
            wsync();
            const int rem = nn - col - 1;
            for (int e = lane; e < rem * rem; e += 64) {
                const int qq = div_small(e, rem);
                const int i = col + 1 + (e - qq * rem), q = col + 1 + qq;
                const double prod = As[col * nn + i] * As[q * nn + col];
                As[q * nn + i] = As[q * nn + i] - prod;
            }
            wsync();
        }
        if (info != 0) return false; // singular pivot matrix: the general path reports the reference's error
        // unit-lower, then upper; a lane per right-hand side
        if (lane < nrhs) {
            double* x = Bs + lane * nn;
            for (int kk = 0; kk < nn; ++kk) {
                const double xk = x[kk];
                for (int i = kk + 1; i < nn; ++i) {
                    const double prod = As[kk * nn + i] * xk;
                    x[i] = x[i] - prod;
                }
            }
            for (int kk = nn - 1; kk >= 0; --kk) {
                const double xk = x[kk] / As[kk * nn + kk];
                x[kk] = xk;
                for (int i = 0; i < kk; ++i) {
                    const double prod = As[kk * nn + i] * xk;
                    x[i] = x[i] - prod;
                }
            }
        }
        wsync();
    }
    // core[l, s, r] = X^T[r + nn (l S + s)] (:1167-1181)
    if (lane < nrhs) {
        const int l = div_small(lane, S), s_ = lane - l * S;
        for (int r = 0; r < nn; ++r) core[l + left_dim * (s_ + S * r)] = zero ? 0.0 : Bs[lane * nn + r];
    }
    if (cdims && lane == 0) {
        cdims[3 * b] = left_dim;
        cdims[3 * b + 1] = S;
        cdims[3 * b + 2] = nn;
    }
    wsync();
    return true;
}

template <int K>
__device__ __forceinline__ void small_copy_tab(int dst, int src)
{
    const int lane = threadIdx.x & 63;
    const int n = ui(SH.n);
    const int* scnt = SH.tab[src].cnt;
    for (int e = lane; e < 2 * n; e += 64) SH.tab[dst].cnt[e] = scnt[e];
    for (int e = lane; e < 2 * n * SC; e += 64) {
        const int fp = e / SC, k = e - fp * SC;
        if (k < scnt[fp]) {
            SH.tab[dst].code[e] = SH.tab[src].code[e];
#pragma unroll
            for (int q = 0; q < K; ++q) SH.tab[dst].acc[e * KS + q] = SH.tab[src].acc[e * KS + q];
        }
    }
    wsync();
}

__device__ __forceinline__ void small_export_tab(int src, int* g_cnt, uint64_t* g_code)
{
    const int lane = threadIdx.x & 63;
    const int n = ui(SH.n);
    const int* scnt = SH.tab[src].cnt;
    for (int e = lane; e < 2 * n; e += 64) g_cnt[e] = scnt[e];
    for (int e = lane; e < 2 * n * SC; e += 64) {
        const int fp = e / SC, k = e - fp * SC;
        if (k < scnt[fp]) g_code[e] = SH.tab[src].code[e];
    }
}

__device__ __forceinline__ int lds_load_acquire(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// waves 1 .. SMALL_WORKERS: fill_site_tensors (tensorci2.rs:1065-1186) of the sites w - 1, w - 1 + SMALL_WORKERS, ... for every job
// wave 0 publishes.  Job e = the fill of iteration e - 1, read from snapshot slot 1 + e % 2 (the sets at the start of iteration e, i.e.
// the sets iteration e - 1 left behind).  Nobody reads these tensors (the next iteration or the final 1-site sweep overwrites every
// site tensor; the reference invalidates them, tensorci2.rs:707-708): they go to the scratch block.
template <int K>
__device__ __forceinline__ void small_worker(double* scratch, size_t scratch_stride, int w)
{
    int next = 1;
    for (;;) {
        int job = lds_load_acquire(&SH.job);
        while (job < next) {
            if (lds_load_acquire(&SH.quit) != 0) {
                job = lds_load_acquire(&SH.job);
                if (job < next) return;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            job = lds_load_acquire(&SH.job);
        }
        const int slot = 1 + next % 2;
        const int n = ui(SH.n);
        int ok = 1;
        for (int b = w - 1; b < n && ok; b += SMALL_WORKERS) ok = small_fill_site<K>(w, slot, b, scratch + (size_t)b * scratch_stride, 0);
        if (!ok && (threadIdx.x & 63) == 0) SH.fail[w] = next;
        wsync();
        if ((threadIdx.x & 63) == 0) lds_store_release(&SH.done[w], next);
        ++next;
    }
}

// the slot about to be overwritten was last read by fill job `upto`: every worker must be through with it
__device__ __forceinline__ void small_wait_jobs(int upto)
{
    for (int w = 1; w < SMALL_WAVES; ++w)
        while (lds_load_acquire(&SH.done[w]) < upto) __builtin_amdgcn_s_sleep(2);
}

template <int K>
__global__ void __launch_bounds__(64 * SMALL_WAVES) small_optimize_kernel(SmallArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t_begin = small_clock();
    const SmallHeader* const hd = &a.h;
    const int n = ui(hd->n), total = ui(hd->total);
    const int flags = ui(hd->flags);
    // ---- input: site info, weights, the current sets; all four waves copy ----
    {
        const int* g_ldim = reinterpret_cast<const int*>(a.in + hd->o_ldim);
        const int* g_woff = reinterpret_cast<const int*>(a.in + hd->o_woff);
        const uint64_t* g_w = reinterpret_cast<const uint64_t*>(a.in + hd->o_w);
        const int* g_cnt = reinterpret_cast<const int*>(a.in + hd->o_cnt);
        const uint64_t* g_code = reinterpret_cast<const uint64_t*>(a.in + hd->o_code);
        const uint64_t* g_acc = reinterpret_cast<const uint64_t*>(a.in + hd->o_acc);
        const int cap_in = ui(hd->cap_in);
        const int tid = threadIdx.x, T = 64 * SMALL_WAVES;
        for (int e = tid; e < n; e += T) {
            SH.ldim[e] = g_ldim[e];
            SH.woff[e] = g_woff[e];
            SH.bond[e] = 0.0;
            SH.shapes[3 * e] = SH.shapes[3 * e + 1] = SH.shapes[3 * e + 2] = 0;
            SH.cdims[3 * e] = SH.cdims[3 * e + 1] = SH.cdims[3 * e + 2] = 0;
        }
        for (int e = tid; e < K * total; e += T) SH.w[e] = g_w[e];
        for (int e = tid; e < 2 * n; e += T) SH.tab[0].cnt[e] = g_cnt[e];
        for (int e = tid; e < 2 * n * cap_in; e += T) { // (every slot: no load waits for a count)
            const int fp = e / cap_in, k = e - fp * cap_in;
            SH.tab[0].code[fp * SC + k] = g_code[e];
#pragma unroll
            for (int q = 0; q < K; ++q) SH.tab[0].acc[(fp * SC + k) * KS + q] = g_acc[(size_t)e * K + q];
        }
        if (tid < T4A_FN_MAX_PARAMS) SH.params[tid] = hd->params[tid];
        if (tid == 0) {
            SH.fid = hd->fid;
            SH.n = n;
            SH.total = total;
            SH.n_pe = 0;
            SH.stamps = (flags & 8) ? 1 : 0;
            SH.rook = (flags & 16) ? 1 : 0;
            SH.rook_bonds = 0;
            SH.rook_visits = 0;
            SH.job = 0;
            SH.quit = 0;
            SH.ph_last = (flags & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
            for (int q = 0; q < SMALL_WAVES; ++q) SH.done[q] = SH.fail[q] = 0;
            for (int q = 0; q < 8; ++q) SH.ph[q] = 0ull;
        }
        __syncthreads();
    }
    if (wave > 0) {
        small_worker<K>(a.scratch, a.scratch_stride, wave);
        return;
    }
    const unsigned long long t_loaded = small_clock();
    const int max_iter = ui(hd->max_iter), ncheck = ui(hd->ncheck), strategy = ui(hd->sweep_strategy);
    const int max_bond_dim = ui(hd->max_bond_dim);
    const double tolerance = hd->tolerance;
    const bool normalize = flags & 1, strictly_nested = flags & 2, final_sweep = flags & 4;
    double* const* const cores = reinterpret_cast<double* const*>(a.in + hd->o_cores);
    int reason = 0;
    const bool stamps = (flags & 8) != 0, rook = (flags & 16) != 0;
    double msv = hd->max_sample_value;
    const int tile_max = ui(hd->tile_max);

    int iters_done = 0, converged = 0, termination = 2 /* MaxIterations */, final_done = 0, status = 1;
    int jobs = 0; // fill jobs published
    // One loop runs the iterations (tensorci2.rs:1659-1776) and then, as its last pass, the final 1-site sweep (:1781-1794): the
    // bond update has ONE call site.
    bool final_phase = false;
    for (;;) {
        if (!final_phase && (iters_done >= max_iter || converged)) {
            if (!final_sweep) break;
            final_phase = true;
        }
        const int iter = iters_done;
        const double msv0 = msv;
        const double norm = (normalize && msv0 > 0.0) ? msv0 : 1.0;
        bool forward = true;
        if (!final_phase) {
            if (strategy == 1) forward = false;
            else if (strategy == 2) forward = (iter % 2 == 0);
        }
        // extras: the sets at the start of the previous iteration (:1675-1685); then the sets as they are join the history (:1686-1689).
        // (The final sweep takes its snapshot into the slot the next iteration would have used: a failure in it hands that state back.)
        const bool use_hist = !final_phase && !strictly_nested && iter > 0;
        const int hist = 1 + (iter + 1) % 2; // (slot of iteration iter - 1)
        const int snap = 1 + iter % 2;
        small_stamp(stamps, 7);
        // this slot holds the sets at the start of iteration iter - 2, which fill job iter - 2 read
        if (iter >= 3) small_wait_jobs(iter - 2);
        small_copy_tab<K>(snap, 0);
        // fill_site_tensors of the iteration that has just ended (:1065-1186, tensorci2.rs:1727): the sets it left behind are this
        // snapshot; the other waves run it while this one walks on
        if (iter > 0) {
            jobs = iter;
            if (lane == 0) lds_store_release(&SH.job, jobs);
        }
        small_stamp(stamps, 6);
        const double rel_tol = final_phase ? 1e-14 : tolerance, abs_tol = final_phase ? tolerance * norm : 0.0;
        if (final_phase) {
            if (lane == 0) SH.n_pe = 0; // flush_pivot_errors (:900)
        } else {
            for (int e = lane; e < n; e += 64) SH.shapes[3 * e] = SH.shapes[3 * e + 1] = SH.shapes[3 * e + 2] = 0;
        }
        wsync();
        int why = 0;
        for (int step = 0; step + 1 < n && why == 0; ++step) {
            const int b = forward ? step : n - 2 - step;
            why = ui(small_update<K>(n, total, tile_max, stamps, rook, forward ? 1 : 0, final_phase ? 1 : 0, hist, use_hist ? 1 : 0, b, max_bond_dim, rel_tol, abs_tol, msv, cores[b]));
        }
        if (why == 0 && final_phase) { // fill_tensor(I_last, J_last) (:1040-1043)
            if (!ui(small_fill_site<K>(0, 0, n - 1, cores[n - 1], 1))) why = 4;
            small_stamp(stamps, 5);
        }
        if (why != 0) { // hand the state at the start of this pass back
            small_copy_tab<K>(0, snap);
            msv = msv0;
            reason = why;
            status = 2;
            break;
        }
        wsync();
        if (final_phase) {
            final_done = 1;
            break;
        }
        double error = 0.0;
        for (int b = 0; b + 1 < n; ++b) error = fmax(error, SH.bond[b]);
        int rk = 0;
        for (int p = 1; p < n; ++p) rk = max(rk, SH.tab[0].cnt[p]);
        rk = ui(rk);
        error = uniform_f64(error);
        if (lane == 0) {
            SH.err[iter] = error / norm;
            SH.rank[iter] = rk;
        }
        wsync();
        iters_done = iter + 1;
        // convergence_criterion (:1407-1437), nglobal == 0 throughout
        if (iters_done >= ncheck) {
            bool errors_converged = true, at_max = true;
            int min_rank = 0x7fffffff;
            for (int q = iters_done - ncheck; q < iters_done; ++q) {
                if (!(SH.err[q] < tolerance)) errors_converged = false;
                if (!(SH.rank[q] >= max_bond_dim)) at_max = false;
                min_rank = min(min_rank, SH.rank[q]);
            }
            const bool rank_stable = min_rank == SH.rank[iters_done - 1];
            if (at_max) {
                termination = 1;
                converged = 1;
            } else if (errors_converged && rank_stable) {
                termination = 0;
                converged = 1;
            }
            converged = ui(converged);
            termination = ui(termination);
        }
    }
    // the fill of the last completed iteration, when no later pass published it (no final sweep, or the loop handed over: `cur` was
    // restored to what that iteration left behind): snapshot the sets into the free slot first
    if (jobs < iters_done) {
        const int snap = 1 + iters_done % 2;
        if (iters_done >= 3) small_wait_jobs(iters_done - 2);
        small_copy_tab<K>(snap, 0);
        jobs = iters_done;
        if (lane == 0) lds_store_release(&SH.job, jobs);
    }
    if (lane == 0) lds_store_release(&SH.quit, 1);
    const unsigned long long t_iters = small_clock();
    small_wait_jobs(jobs);
    small_stamp(stamps, 5);
    {   // a site that could not be filled (singular pivot matrix, shapes beyond the workspace): the whole call goes to the general
        // path, which reports what the reference reports
        int any_fail = 0;
        for (int w = 1; w < SMALL_WAVES; ++w) any_fail |= SH.fail[w];
        if (ui(any_fail) != 0) {
            status = 3;
            reason = 4;
        }
    }
    // the history entry the host keeps: the sets at the start of the last completed iteration
    const int hist_slot = iters_done > 0 ? 1 + (iters_done - 1) % 2 : -1;
    // ---- results ----
    wsync();
    {
        const SmallOutLayout OL = small_out_layout(n);
        SmallOutHeader* const oh = reinterpret_cast<SmallOutHeader*>(a.out);
        double* g_err = reinterpret_cast<double*>(a.out + OL.o_err);
        int* g_rank = reinterpret_cast<int*>(a.out + OL.o_rank);
        double* g_bond = reinterpret_cast<double*>(a.out + OL.o_bond);
        double* g_pe = reinterpret_cast<double*>(a.out + OL.o_pe);
        int* g_shapes = reinterpret_cast<int*>(a.out + OL.o_shapes);
        int* g_cdims = reinterpret_cast<int*>(a.out + OL.o_cdims);
        const int n_pe = ui(SH.n_pe);
        for (int e = lane; e < iters_done; e += 64) {
            g_err[e] = SH.err[e];
            g_rank[e] = SH.rank[e];
        }
        for (int e = lane; e < n; e += 64) g_bond[e] = SH.bond[e];
        for (int e = lane; e < n_pe; e += 64) g_pe[e] = SH.pe[e];
        for (int e = lane; e < 3 * n; e += 64) {
            g_shapes[e] = SH.shapes[e];
            g_cdims[e] = SH.cdims[e];
        }
        small_export_tab(0, reinterpret_cast<int*>(a.out + OL.o_cnt), reinterpret_cast<uint64_t*>(a.out + OL.o_code));
        if (hist_slot >= 0) small_export_tab(hist_slot, reinterpret_cast<int*>(a.out + OL.o_hcnt), reinterpret_cast<uint64_t*>(a.out + OL.o_hcode));
        if (lane == 0) {
            oh->status = status;
            oh->iters_done = iters_done;
            oh->converged = converged;
            oh->termination = termination;
            oh->n_pivot_errors = n_pe;
            oh->final_done = final_done;
            oh->hist_valid = hist_slot >= 0 ? 1 : 0;
            oh->reason = reason;
            oh->max_sample_value = msv;
            oh->clocks[0] = t_loaded - t_begin;
            oh->clocks[1] = t_iters - t_loaded;
            oh->clocks[2] = small_clock() - t_iters;
            for (int q = 0; q < 8; ++q) oh->clocks[3 + q] = SH.ph[q];
            oh->clocks[11] = ((unsigned long long)(unsigned)SH.rook_bonds << 32) | (unsigned)SH.rook_visits;
        }
        __threadfence_system();
        if (lane == 0) {
            volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(a.out + OL.o_flag);
            *flag = a.token;
        }
    }
}

} // namespace

size_t small_lds_bytes(int n, int K, int total)
{
    // (everything is static LDS: the value only says whether the problem fits the fixed tables)
    if (n < 2 || n > SMALL_MAX_SITES || K < 1 || K > KS || total < 1 || K * total > SMALL_MAX_W) return 0;
    return sizeof(SmallStatic);
}

void small_optimize_launch(const SmallArgs& a, int n, int K, int total, hipStream_t stream)
{
    if (small_lds_bytes(n, K, total) == 0) throw std::runtime_error("small_optimize_launch: the problem does not fit the small-problem engine");
    if (K == 1) hipLaunchKernelGGL(small_optimize_kernel<1>, dim3(1), dim3(64 * SMALL_WAVES), 0, stream, a);
    else hipLaunchKernelGGL(small_optimize_kernel<2>, dim3(1), dim3(64 * SMALL_WAVES), 0, stream, a);
}

} // namespace t4a
