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

struct SmallLds { // byte offsets into the dynamic LDS
    size_t tab_bytes, o_cnt, o_code, o_acc; // inside one table copy
    size_t o_tab, o_params, o_w, o_ldim, o_woff, o_lcode, o_lacc, o_bond, o_shapes, o_err, o_rank, o_pe, o_As, o_Bs, o_fl, o_Af, o_xs, o_pp, o_cdims, bytes;
};
__host__ __device__ inline size_t up16(size_t v) { return (v + 15) / 16 * 16; }
__host__ __device__ inline SmallLds small_lds(int n, int K, int total)
{
    SmallLds L;
    L.o_cnt = 0;
    L.o_code = up16(sizeof(int) * 2 * (size_t)n);
    L.o_acc = L.o_code + sizeof(uint64_t) * 2 * (size_t)n * SC;
    L.tab_bytes = up16(L.o_acc + sizeof(uint64_t) * 2 * (size_t)n * SC * (size_t)K);
    size_t o = 0;
    L.o_tab = o;    o += 4 * L.tab_bytes;
    L.o_params = o; o += sizeof(double) * 16;
    L.o_w = o;      o += up16(sizeof(uint64_t) * (size_t)K * (size_t)total);
    L.o_ldim = o;   o += up16(sizeof(int) * (size_t)n);
    L.o_woff = o;   o += up16(sizeof(int) * (size_t)n);
    L.o_lcode = o;  o += sizeof(uint64_t) * 64;
    L.o_lacc = o;   o += sizeof(uint64_t) * 64 * (size_t)K;
    L.o_bond = o;   o += up16(sizeof(double) * (size_t)n);
    L.o_shapes = o; o += up16(sizeof(int) * 3 * (size_t)n);
    L.o_err = o;    o += sizeof(double) * SMALL_MAX_ITER;
    L.o_rank = o;   o += sizeof(int) * SMALL_MAX_ITER;
    L.o_pe = o;     o += up16(sizeof(double) * (SMALL_TILE + 2));
    L.o_As = o;     o += sizeof(double) * SC * SC;
    L.o_Bs = o;     o += sizeof(double) * SC * 64;
    L.o_fl = o;     o += sizeof(uint64_t) * (64 + 2 * SC) * (size_t)K;
    L.o_Af = o;     o += sizeof(double) * SMALL_TILE * SMALL_TILE;
    L.o_xs = o;     o += sizeof(double) * SC * 64;
    L.o_pp = o;     o += sizeof(int) * 64;
    L.o_cdims = o;  o += up16(sizeof(int) * 3 * (size_t)n);
    L.bytes = o;
    return L;
}

template <int K> struct Tab {
    int* cnt;
    uint64_t* code;
    uint64_t* acc;
    int n;
    __device__ __forceinline__ int& c(int f, int p) const { return cnt[f * n + p]; }
    __device__ __forceinline__ uint64_t* codes(int f, int p) const { return code + (size_t)(f * n + p) * SC; }
    __device__ __forceinline__ uint64_t* accs(int f, int p) const { return acc + (size_t)(f * n + p) * SC * K; }
};

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
__device__ __forceinline__ unsigned long long small_clock() { return __builtin_amdgcn_s_memrealtime(); }

// The scalar stage of the built-in functor (include/t4a_testfunctions.h), ONE copy per kernel: inlined at every evaluation site the
// four function bodies were most of a 200 000-line kernel.
__device__ __attribute__((noinline)) double small_fn_value(int fid, uint64_t a0, uint64_t a1, const double* params)
{
    uint64_t acc[T4A_FN_MAX_ACC] = {a0, a1, 0, 0};
    double p[T4A_FN_MAX_PARAMS];
#pragma unroll
    for (int q = 0; q < T4A_FN_MAX_PARAMS; ++q) p[q] = params[q];
    return t4a_fn_value(fid, acc, p);
}

template <int K> struct Ctx {
    int n, total, fid;
    char* tab_base;
    unsigned tab_bytes, tab_o_code, tab_o_acc;
    const double* params; // LDS copy of the functor's parameters
    __device__ __forceinline__ Tab<K> tab(int t) const
    {
        Tab<K> v;
        char* base = tab_base + (size_t)t * tab_bytes;
        v.cnt = reinterpret_cast<int*>(base);
        v.code = reinterpret_cast<uint64_t*>(base + tab_o_code);
        v.acc = reinterpret_cast<uint64_t*>(base + tab_o_acc);
        v.n = n;
        return v;
    }
    uint64_t* w;
    int *ldim, *woff;
    uint64_t *lcode, *lacc;
    double* bond;
    int* shapes;
    double *err;
    int* rank;
    double* pe;
    double *As, *Bs;
    uint64_t* fl;
    double *Af, *xs;
    int* pp;
    int* cdims;
    double msv;   // max_sample_value (wave-uniform)
    int n_pe;
    int reason;
};

// ---- the two lists of a bond: rows in list slots 0..31, columns in 32..63; lanes 0..31 build the rows, 32..63 the columns -----------
// 2-site: rows = kron(I_b, d_b) + extras(H.I[b+1]), columns = kron(J_{b+1}, d_{b+1}) + extras(H.J[b]) (tensorci2.rs:1224-1246, :1833-1846);
// 1-site forward (:918-1050): rows = kron(I_b, d_b), columns = J_b itself.
template <int K>
__device__ __forceinline__ bool small_lists(Ctx<K>& c, const Tab<K>& cur, const Tab<K>& hist, bool use_hist, int b, bool one_site, int& M, int& N)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
    const bool direct = one_site && half;
    const int ps = half ? (direct ? b : b + 1) : b;
    const int np = cur.c(half, ps);
    const int d = direct ? 1 : c.ldim[ps], wo = c.woff[ps];
    const int m0 = np * d;
    bool ok = np >= 1 && np <= SC && m0 <= SMALL_TILE;
    const uint64_t* pcode = cur.codes(half, ps);
    const uint64_t* pacc = cur.accs(half, ps);
    uint64_t code = 0, acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0;
    if (ok && t < m0) {
        int parent, digit;
        if (half) { // (digit outer, parent inner)
            digit = div_small(t, np);
            parent = t - digit * np;
        } else {    // (parent outer, digit inner)
            parent = div_small(t, d);
            digit = t - parent * d;
        }
        const uint64_t pc = pcode[parent];
        code = direct ? pc : (uint64_t)digit + (uint64_t)d * pc;
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = pacc[parent * K + q] + (direct ? 0ull : c.w[q * c.total + wo + digit]);
    }
    bool keep = false;
    uint64_t xc = 0;
    int ne = 0;
    const int hsite = half ? b : b + 1;
    if (use_hist && !one_site) {
        ne = hist.c(half, hsite);
        ok = ok && ne >= 0 && ne <= SC;
        if (ok && t < ne) {
            xc = hist.codes(half, hsite)[t];
            keep = true;
        }
        const int npmax = ui(max(__builtin_amdgcn_readlane(np, 0), __builtin_amdgcn_readlane(np, 32)));
        for (int pi = 0; pi < npmax; ++pi) {
            const uint64_t base = pcode[pi < np ? pi : 0] * (uint64_t)d;
            if (pi < np && (xc - base) < (uint64_t)d) keep = false; // its parent is among the parents: already in the Kronecker part
        }
    }
    const unsigned long long km = __ballot(keep);
    const unsigned kmh = half ? (unsigned)(km >> 32) : (unsigned)km;
    const int nkeep = __builtin_popcount(kmh);
    const int pos = m0 + __builtin_popcount(kmh & ((1u << t) - 1u));
    const int tot = m0 + nkeep;
    ok = ok && tot <= SMALL_TILE;
    if (__ballot(!ok) != 0ull) return false;
    if (t < m0) {
        c.lcode[half * 32 + t] = code;
#pragma unroll
        for (int q = 0; q < K; ++q) c.lacc[(half * 32 + t) * K + q] = acc[q];
    }
    if (keep) {
        c.lcode[half * 32 + pos] = xc;
#pragma unroll
        for (int q = 0; q < K; ++q) c.lacc[(half * 32 + pos) * K + q] = hist.accs(half, hsite)[t * K + q];
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
//   with FACT the factored matrix (L scaled below the pivots in the pivot columns, U in the pivot rows) goes to c.Af[i + MR j].
template <int K, int E>
__device__ __forceinline__ int small_bond(Ctx<K>& c, const bool LEFT, const bool FACT, int M, int N, int max_bond_dim, double rel_tol, double abs_tol, int& ptab,
                                          double& pvabs, double& error_out, int& rpos_out)
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
        for (int q = 0; q < K; ++q) racc[q] = c.lacc[(i < M ? i : 0) * K + q];
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
                    for (int q = 0; q < K; ++q) acc[q] = racc[q] + c.lacc[(32 + (j < N ? j : 0)) * K + q];
                    v = small_fn_value(c.fid, acc[0], acc[1], c.params);
                }
                c.Af[lane + 64 * e] = v;
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
                for (int q = 0; q < K; ++q) acc[q] = racc[q] + c.lacc[(32 + (j < N ? j : 0)) * K + q];
                v = small_fn_value(c.fid, acc[0], acc[1], c.params);
            } else {
                v = c.Af[lane + 64 * e];
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
        if (am > c.msv) c.msv = am;
    }
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
        const unsigned long long multi = __ballot(mhi == whi && cnt_l > 1);
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
        const double l = LEFT ? xcd_div(xcol, wval, rp, pmid) : xcol;
        const bool row_rest = rowlive && i != pr; // rows behind the new pivot row in the permuted order
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = jb + LPC * e;
            const bool in = (inb >> e) & 1u;
            const bool col_rest = cpos[e] >= kn && j != pc;
            const double u = LEFT ? urow[e] : xcd_div(urow[e], wval, rp, pmid);
            if (in && row_rest && col_rest) {
                const double prod = l * u;
                a[e] = a[e] - prod; // separately rounded (matrixlu.rs:593-612)
            } else if (LEFT && in && row_rest && j == pc) {
                a[e] = l;
            } else if (!LEFT && in && col_rest && i == pr) {
                a[e] = u;
            }
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
    if (FACT) {
#pragma unroll
        for (int e = 0; e < E; ++e) c.Af[lane + 64 * e] = a[e];
    }
    return npiv;
}

// pivots -> I_{b+1} (lanes 0..31) and J_b (lanes 32..63) (tensorci2.rs:1934-1940 with non_empty_or_first :1813-1819)
template <int K>
__device__ __forceinline__ bool small_gather(Ctx<K>& c, const Tab<K>& cur, int b, int r, int ptab)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
    const int cnt = r > 0 ? r : 1;
    if (cnt > SC) return false;
    const int site = half ? b : b + 1;
    uint64_t code = 0, acc[K];
    if (t < cnt) {
        const int src = half * 32 + (r > 0 ? ptab : 0);
        code = c.lcode[src];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = c.lacc[src * K + q];
        cur.codes(half, site)[t] = code;
#pragma unroll
        for (int q = 0; q < K; ++q) cur.accs(half, site)[t * K + q] = acc[q];
    }
    if (t == 0) cur.c(half, site) = cnt;
    wsync();
    return true;
}

// One bond of a half-sweep / of the final 1-site sweep.  Returns false: hand the iteration back.
template <int K>
__device__ __forceinline__ bool small_update(Ctx<K>& c, const bool LEFT, const bool ONE, const Tab<K>& cur, const Tab<K>& hist, bool use_hist, int b, int max_bond_dim, double rel_tol,
                                             double abs_tol, double* core)
{
    const int lane = threadIdx.x & 63;
    int M, N;
    if (!small_lists<K>(c, cur, hist, use_hist, b, ONE, M, N)) {
        c.reason = 1;
        return false;
    }
    int ptab, rpos, r;
    double pvabs, error;
    const bool FACT = ONE;
    if (M <= 8 && N <= 8) r = small_bond<K, 1>(c, LEFT, FACT, M, N, max_bond_dim, rel_tol, abs_tol, ptab, pvabs, error, rpos);
    else if (M <= 16 && N <= 16) r = small_bond<K, 4>(c, LEFT, FACT, M, N, max_bond_dim, rel_tol, abs_tol, ptab, pvabs, error, rpos);
    else r = small_bond<K, 16>(c, LEFT, FACT, M, N, max_bond_dim, rel_tol, abs_tol, ptab, pvabs, error, rpos);
    if (r < 0) {
        c.reason = 2;
        return false;
    }
    const int L_b = b == 0 ? 1 : ui(cur.c(0, b)); // (before the gather: I_b is not touched by bond b)
    if (!small_gather<K>(c, cur, b, r, ptab)) {
        c.reason = 3;
        return false;
    }
    if (lane == 0) c.bond[b] = error; // (pivot_errors.back(), tensorci2.rs:1942-1949 / :1998)
    if (!ONE) {
        if (lane == 0) {
            c.shapes[3 * b] = M;
            c.shapes[3 * b + 1] = N;
            c.shapes[3 * b + 2] = r;
        }
        return true;
    }
    // ---- 1-site sweep with update_tensors (tensorci2.rs:991-1017): pivot_errors (:801-808), site tensor b = LUCI left factor ----
    {   // update_pivot_errors(factors.pivot_errors): [sqrt(d * d) of the r pivots] + [error]
        const double mine = lane < r ? pvabs : error;
        if (lane <= r) {
            const double old = lane < c.n_pe ? c.pe[lane] : 0.0;
            c.pe[lane] = fmax(old, mine);
        }
        if (r + 1 > c.n_pe) c.n_pe = r + 1;
    }
    // left = P_row^T [I ; L21 L11^-1] (matrix_luci.rs:206-229): row at permuted position p < r is the unit vector e_p, the others
    // solve x L11 = l_row with L11 unit lower triangular (the scaled pivot columns of the pivot rows)
    const int MR = (M <= 8 && N <= 8) ? 8 : ((M <= 16 && N <= 16) ? 16 : 32);
    if ((lane & 31) < r) c.pp[lane] = ptab; // pp[k] pivot row k, pp[32 + k] pivot column k
    wsync();
    const int S = c.ldim[b];
    const int R = r > 0 ? r : 1;
    if (lane < M) {
        if (r == 0) {
            c.xs[lane] = 0.0;
        } else if (rpos < r) {
            for (int k = 0; k < r; ++k) c.xs[k * 64 + lane] = (k == rpos) ? 1.0 : 0.0;
        } else {
            for (int k = r - 1; k >= 0; --k) {
                const int pck = c.pp[32 + k];
                double s = c.Af[lane + MR * pck];
                for (int t = k + 1; t < r; ++t) {
                    const double prod = c.xs[t * 64 + lane] * c.Af[c.pp[t] + MR * pck];
                    s = s - prod;
                }
                c.xs[k * 64 + lane] = s;
            }
        }
        // t(l, s, k) = left(l S + s, k), column-major [l, s, k] (tensorci2.rs:994-1004)
        const int l = div_small(lane, S), s_ = lane - l * S;
        for (int k = 0; k < R; ++k) core[l + L_b * (s_ + S * k)] = c.xs[k * 64 + lane];
    }
    if (lane == 0) {
        c.cdims[3 * b] = L_b;
        c.cdims[3 * b + 1] = S;
        c.cdims[3 * b + 2] = R;
    }
    wsync();
    return true;
}

// fill_site_tensors of site b (tensorci2.rs:1065-1186) from table `tb`, one wavefront; the partial-pivot LU and the substitutions of
// fill_small_kernel (kernels_pi.hip), operation for operation.  Returns false: not representable here / singular.
template <int K>
__device__ __forceinline__ bool small_fill_site(Ctx<K>& c, const Tab<K>& tb, int b, double* core, int* cdims)
{
    const int lane = threadIdx.x & 63;
    const int n = c.n;
    const int Lb = ui(tb.c(0, b)), S = c.ldim[b], wo = c.woff[b];
    const int nj = ui(tb.c(1, b));
    const int ni = Lb * S;
    if (ni > 64 || ni < 1 || nj < 1 || nj > SC) return false;
    uint64_t* ka = c.fl;                 // [64][K] kron(I_b, d_b)
    uint64_t* ja = c.fl + 64 * K;        // [SC][K] J_b
    uint64_t* ia = c.fl + (64 + SC) * K; // [SC][K] I_{b+1}
    if (lane < ni) {
        const int parent = div_small(lane, S), digit = lane - parent * S;
#pragma unroll
        for (int q = 0; q < K; ++q) ka[lane * K + q] = tb.accs(0, b)[parent * K + q] + c.w[q * c.total + wo + digit];
    }
    if (lane < nj) {
#pragma unroll
        for (int q = 0; q < K; ++q) ja[lane * K + q] = tb.accs(1, b)[lane * K + q];
    }
    const bool last = b == n - 1;
    const int np = last ? 0 : ui(tb.c(0, b + 1));
    if (!last) {
        if (np != nj) return false;
        if (lane < np) {
#pragma unroll
            for (int q = 0; q < K; ++q) ia[lane * K + q] = tb.accs(0, b + 1)[lane * K + q];
        }
    }
    wsync();
    const int left_dim = b == 0 ? 1 : Lb;
    if (last) { // :1109-1128: t(l, s, 0) = Pi1(l S + s, 0)
        if (lane < ni) {
            uint64_t acc[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = ka[lane * K + q] + ja[q];
            const double v = small_fn_value(c.fid, acc[0], acc[1], c.params);
            const int l = div_small(lane, S), s_ = lane - l * S;
            core[l + left_dim * s_] = v;
        }
        if (lane == 0) {
            cdims[3 * b] = left_dim;
            cdims[3 * b + 1] = S;
            cdims[3 * b + 2] = 1;
        }
        return true;
    }
    const int nn = nj, nrhs = ni;
    double* As = c.As; // P^T, nn x nn column-major: As[col * nn + i] = f(I_{b+1}[col], J_b[i])
    double* Bs = c.Bs; // Pi1^T, nn x nrhs:          Bs[col * nn + i] = f(kron[col], J_b[i])
    double pm = 0.0;
    bool bad = false;
    for (int e = lane; e < nn * nn; e += 64) {
        const int col = div_small(e, nn), i = e - col * nn;
        uint64_t acc[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = ia[col * K + q] + ja[i * K + q];
        const double v = small_fn_value(c.fid, acc[0], acc[1], c.params);
        As[e] = v;
        bad |= !((v - v) == 0.0);
        const double av = sqrt(v * v);
        if (av > pm) pm = av;
    }
    for (int e = lane; e < nn * nrhs; e += 64) {
        const int col = div_small(e, nn), i = e - col * nn;
        uint64_t acc[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = ka[col * K + q] + ja[i * K + q];
        const double v = small_fn_value(c.fid, acc[0], acc[1], c.params);
        Bs[e] = v;
        bad |= !((v - v) == 0.0);
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
    if (lane == 0) {
        cdims[3 * b] = left_dim;
        cdims[3 * b + 1] = S;
        cdims[3 * b + 2] = nn;
    }
    wsync();
    return true;
}

template <int K>
__device__ __forceinline__ void small_copy_tab(const Tab<K>& dst, const Tab<K>& src, int n)
{
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 2 * n; e += 64) dst.cnt[e] = src.cnt[e];
    for (int e = lane; e < 2 * n * SC; e += 64) {
        const int fp = e / SC, k = e - fp * SC;
        if (k < src.cnt[fp]) {
            dst.code[e] = src.code[e];
#pragma unroll
            for (int q = 0; q < K; ++q) dst.acc[(size_t)e * K + q] = src.acc[(size_t)e * K + q];
        }
    }
    wsync();
}

template <int K>
__device__ __forceinline__ void small_export_tab(const Tab<K>& src, int n, int* g_cnt, uint64_t* g_code)
{
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 2 * n; e += 64) g_cnt[e] = src.cnt[e];
    for (int e = lane; e < 2 * n * SC; e += 64) {
        const int fp = e / SC, k = e - fp * SC;
        if (k < src.cnt[fp]) g_code[e] = src.code[e];
    }
}

template <int K>
__global__ void __launch_bounds__(64) small_optimize_kernel(SmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const unsigned long long t_begin = small_clock();
    const SmallHeader* const hd = &a.h;
    const int n = ui(hd->n), total = ui(hd->total);
    const SmallLds L = small_lds(n, K, total);
    Ctx<K> c;
    c.n = n;
    c.total = total;
    c.fid = hd->fid;
    c.tab_base = lds + L.o_tab;
    c.tab_bytes = (unsigned)L.tab_bytes;
    c.tab_o_code = (unsigned)L.o_code;
    c.tab_o_acc = (unsigned)L.o_acc;
    {
        double* pl = reinterpret_cast<double*>(lds + L.o_params);
        if (lane < T4A_FN_MAX_PARAMS) pl[lane] = hd->params[lane];
        c.params = pl;
    }
    c.w = reinterpret_cast<uint64_t*>(lds + L.o_w);
    c.ldim = reinterpret_cast<int*>(lds + L.o_ldim);
    c.woff = reinterpret_cast<int*>(lds + L.o_woff);
    c.lcode = reinterpret_cast<uint64_t*>(lds + L.o_lcode);
    c.lacc = reinterpret_cast<uint64_t*>(lds + L.o_lacc);
    c.bond = reinterpret_cast<double*>(lds + L.o_bond);
    c.shapes = reinterpret_cast<int*>(lds + L.o_shapes);
    c.err = reinterpret_cast<double*>(lds + L.o_err);
    c.rank = reinterpret_cast<int*>(lds + L.o_rank);
    c.pe = reinterpret_cast<double*>(lds + L.o_pe);
    c.As = reinterpret_cast<double*>(lds + L.o_As);
    c.Bs = reinterpret_cast<double*>(lds + L.o_Bs);
    c.fl = reinterpret_cast<uint64_t*>(lds + L.o_fl);
    c.Af = reinterpret_cast<double*>(lds + L.o_Af);
    c.xs = reinterpret_cast<double*>(lds + L.o_xs);
    c.pp = reinterpret_cast<int*>(lds + L.o_pp);
    c.cdims = reinterpret_cast<int*>(lds + L.o_cdims);
    c.n_pe = 0;
    c.reason = 0;
    c.msv = hd->max_sample_value;
    const int max_iter = ui(hd->max_iter), ncheck = ui(hd->ncheck), strategy = ui(hd->sweep_strategy), flags = ui(hd->flags);
    const int max_bond_dim = ui(hd->max_bond_dim);
    const double tolerance = hd->tolerance;
    const bool normalize = flags & 1, strictly_nested = flags & 2, final_sweep = flags & 4;
    double* const* const cores = reinterpret_cast<double* const*>(a.in + hd->o_cores);
    const SmallOutLayout OL = small_out_layout(n);
    SmallOutHeader* const oh = reinterpret_cast<SmallOutHeader*>(a.out);

    // ---- input: site info, weights, the current sets (packed: cnt[f n + p] entries at off[f n + p]) ----
    {
        const int* g_ldim = reinterpret_cast<const int*>(a.in + hd->o_ldim);
        const int* g_woff = reinterpret_cast<const int*>(a.in + hd->o_woff);
        const uint64_t* g_w = reinterpret_cast<const uint64_t*>(a.in + hd->o_w);
        const int* g_cnt = reinterpret_cast<const int*>(a.in + hd->o_cnt);
        const uint64_t* g_code = reinterpret_cast<const uint64_t*>(a.in + hd->o_code);
        const uint64_t* g_acc = reinterpret_cast<const uint64_t*>(a.in + hd->o_acc);
        const int cap_in = ui(hd->cap_in);
        for (int e = lane; e < n; e += 64) {
            c.ldim[e] = g_ldim[e];
            c.woff[e] = g_woff[e];
            c.bond[e] = 0.0;
            c.shapes[3 * e] = c.shapes[3 * e + 1] = c.shapes[3 * e + 2] = 0;
            c.cdims[3 * e] = c.cdims[3 * e + 1] = c.cdims[3 * e + 2] = 0;
        }
        for (int e = lane; e < K * total; e += 64) c.w[e] = g_w[e];
        const Tab<K> t0 = c.tab(0);
        for (int e = lane; e < 2 * n; e += 64) t0.cnt[e] = g_cnt[e];
        for (int e = lane; e < 2 * n * cap_in; e += 64) { // (every slot: no load waits for a count)
            const int fp = e / cap_in, k = e - fp * cap_in;
            t0.code[(size_t)fp * SC + k] = g_code[e];
#pragma unroll
            for (int q = 0; q < K; ++q) t0.acc[((size_t)fp * SC + k) * K + q] = g_acc[(size_t)e * K + q];
        }
        wsync();
    }
    const Tab<K> cur = c.tab(0);
    const unsigned long long t_loaded = small_clock();

    int iters_done = 0, converged = 0, termination = 2 /* MaxIterations */, final_done = 0, status = 1;
    // One loop runs the iterations (tensorci2.rs:1659-1776) and then, as its last pass, the final 1-site sweep (:1781-1794): the
    // bond update and the fill have ONE call site each (everything is inlined into this kernel).
    bool final_phase = false;
    for (;;) {
        if (!final_phase && (iters_done >= max_iter || converged)) {
            if (!final_sweep) break;
            final_phase = true;
        }
        const int iter = iters_done;
        const double norm = (normalize && c.msv > 0.0) ? c.msv : 1.0;
        bool forward = true;
        if (!final_phase) {
            if (strategy == 1) forward = false;
            else if (strategy == 2) forward = (iter % 2 == 0);
        }
        // extras: the sets at the start of the previous iteration (:1675-1685); then the sets as they are join the history (:1686-1689).
        // (The final sweep takes its snapshot into the slot the next iteration would have used: a failure in it hands that state back.)
        const bool use_hist = !final_phase && !strictly_nested && iter > 0;
        const Tab<K> hist = c.tab(1 + (iter + 2) % 3); // (slot of iteration iter - 1)
        const Tab<K> snap = c.tab(1 + iter % 3);
        small_copy_tab<K>(snap, cur, n);
        const double msv0 = c.msv;
        const double rel_tol = final_phase ? 1e-14 : tolerance, abs_tol = final_phase ? tolerance * norm : 0.0;
        if (final_phase) c.n_pe = 0; // flush_pivot_errors (:900)
        else
            for (int e = lane; e < n; e += 64) c.shapes[3 * e] = c.shapes[3 * e + 1] = c.shapes[3 * e + 2] = 0;
        bool ok = true;
        for (int step = 0; step + 1 < n && ok; ++step) {
            const int b = forward ? step : n - 2 - step;
            ok = small_update<K>(c, forward, final_phase, cur, hist, use_hist, b, max_bond_dim, rel_tol, abs_tol, cores[b]);
        }
        // fill_site_tensors (:1065-1186).  Its tensors are read by nobody when the final 1-site sweep follows (it overwrites every
        // site tensor): they go to the scratch block then, to the handle's site tensors otherwise.  The final sweep itself only
        // evaluates the last site's tensor (fill_tensor, :1040-1043).
        for (int b = final_phase ? n - 1 : 0; b < n && ok; ++b) {
            double* dst = (final_phase || !final_sweep) ? cores[b] : a.scratch + (size_t)b * a.scratch_stride;
            ok = small_fill_site<K>(c, cur, b, dst, c.cdims);
            if (!ok) c.reason = 4;
        }
        if (!ok) { // hand the state at the start of this pass back
            small_copy_tab<K>(cur, snap, n);
            c.msv = msv0;
            status = 2;
            break;
        }
        wsync();
        if (final_phase) {
            final_done = 1;
            break;
        }
        double error = 0.0;
        for (int b = 0; b + 1 < n; ++b) error = fmax(error, c.bond[b]);
        int rk = 0;
        for (int p = 1; p < n; ++p) rk = max(rk, cur.cnt[p]);
        rk = ui(rk);
        error = uniform_f64(error);
        if (lane == 0) {
            c.err[iter] = error / norm;
            c.rank[iter] = rk;
        }
        wsync();
        iters_done = iter + 1;
        // convergence_criterion (:1407-1437), nglobal == 0 throughout
        if (iters_done >= ncheck) {
            bool errors_converged = true, at_max = true;
            int min_rank = 0x7fffffff;
            for (int q = iters_done - ncheck; q < iters_done; ++q) {
                if (!(c.err[q] < tolerance)) errors_converged = false;
                if (!(c.rank[q] >= max_bond_dim)) at_max = false;
                min_rank = min(min_rank, c.rank[q]);
            }
            const bool rank_stable = min_rank == c.rank[iters_done - 1];
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
    const unsigned long long t_iters = small_clock();
    const int hist_slot = iters_done > 0 ? 1 + (iters_done - 1) % 3 : -1;
    // ---- results ----
    wsync();
    {
        double* g_err = reinterpret_cast<double*>(a.out + OL.o_err);
        int* g_rank = reinterpret_cast<int*>(a.out + OL.o_rank);
        double* g_bond = reinterpret_cast<double*>(a.out + OL.o_bond);
        double* g_pe = reinterpret_cast<double*>(a.out + OL.o_pe);
        int* g_shapes = reinterpret_cast<int*>(a.out + OL.o_shapes);
        int* g_cdims = reinterpret_cast<int*>(a.out + OL.o_cdims);
        for (int e = lane; e < iters_done; e += 64) {
            g_err[e] = c.err[e];
            g_rank[e] = c.rank[e];
        }
        for (int e = lane; e < n; e += 64) g_bond[e] = c.bond[e];
        for (int e = lane; e < c.n_pe; e += 64) g_pe[e] = c.pe[e];
        for (int e = lane; e < 3 * n; e += 64) {
            g_shapes[e] = c.shapes[e];
            g_cdims[e] = c.cdims[e];
        }
        small_export_tab<K>(cur, n, reinterpret_cast<int*>(a.out + OL.o_cnt), reinterpret_cast<uint64_t*>(a.out + OL.o_code));
        if (hist_slot >= 0)
            small_export_tab<K>(c.tab(hist_slot), n, reinterpret_cast<int*>(a.out + OL.o_hcnt), reinterpret_cast<uint64_t*>(a.out + OL.o_hcode));
        if (lane == 0) {
            oh->status = status;
            oh->iters_done = iters_done;
            oh->converged = converged;
            oh->termination = termination;
            oh->n_pivot_errors = c.n_pe;
            oh->final_done = final_done;
            oh->hist_valid = hist_slot >= 0 ? 1 : 0;
            oh->reason = c.reason;
            oh->max_sample_value = c.msv;
            oh->clocks[0] = t_loaded - t_begin;
            oh->clocks[1] = t_iters - t_loaded;
            oh->clocks[2] = small_clock() - t_iters;
            oh->clocks[3] = 0ull;
        }
    }
    __threadfence_system();
    if (lane == 0) {
        volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(a.out + OL.o_flag);
        *flag = a.token;
    }
}

} // namespace

size_t small_lds_bytes(int n, int K, int total)
{
    if (n < 2 || n > SMALL_MAX_SITES || K < 1 || K > 2 || total < 1 || K * total > SMALL_MAX_W) return 0;
    const SmallLds L = small_lds(n, K, total);
    return L.bytes <= (size_t)150 * 1024 ? L.bytes : 0;
}

void small_optimize_launch(const SmallArgs& a, int n, int K, int total, hipStream_t stream)
{
    const size_t bytes = small_lds_bytes(n, K, total);
    if (bytes == 0) throw std::runtime_error("small_optimize_launch: the problem does not fit the small-problem engine");
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_optimize_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_optimize_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (K == 1) hipLaunchKernelGGL(small_optimize_kernel<1>, dim3(1), dim3(64), bytes, stream, a);
    else hipLaunchKernelGGL(small_optimize_kernel<2>, dim3(1), dim3(64), bytes, stream, a);
}

} // namespace t4a
