// kernels_rrlu_w1_body.hpp — K2 for matrices of at most 64 rows and 64 columns: ONE wavefront, no barrier, no mailbox
// ("one wave for <= 64 x 64 — lane = row, columns in registers, DPP arg-max, v_readlane pivot row").
//
// Lane l holds row l, the NC columns sit in registers (groups of eight, so that the run-time accesses — the pivot column read and
// its clearing — index inside one vector).  A pivot step is one basic block in the normal case:
//   wave maximum of the per-lane maxima the previous update left behind (integer DPP on the high words) -> the lane that holds it
//   (ballot) -> the register group and the column inside it (compares against the maximum, bit `lane` of each mask) -> stop tests
//   -> l = column / pivot through the refined reciprocal (started from |pivot| before the column is known: the refinement is
//   odd-symmetric) -> the pivot column is CLEARED in the registers (with FACTORS it goes to the LDS first, and the pivot row with
//   it) -> permutation tables (lane arrays in registers) -> pivot row by v_readlane and the un-fused rank-1 update of ALL columns of every
//   register group that still has a live column, fused with the maxima for the next step.
// There is no column mask and no row mask: a cleared column stays zero (0 - l * 0), the pivot row carries l = pivot / pivot = 1
// and leaves exact zeros, and zeros never win the normal-case search.  Ties, zero and subnormal maxima take the exact sweep on
// (v * v, position) with explicit position tests.
//
// Same contract as the other rrLU kernels: bit-identical to rrlu_mut (tensor4all-core/src/matrixlu.rs:735-819; arg-max :480-519:
// key v*v, first strict maximum in column-major order of the permuted trailing block; stop rules :757-781; un-fused update
// :593-612); a right-orthogonal factorisation runs as the left-orthogonal one of A^T with the row-major tie order.  Arguments and
// result block are RrluXcdArgs (kernels.hpp), including the bond-chain extensions (`urows` is not used: the rows of U wait in the
// LDS).  Finite matrices only: a NaN / infinity in the input or an overflow in the trailing block ends the launch with code 2 and
// the caller runs the first-generation single-XCD kernel.
//
// The body is a header because two kernels run it: the launchers of kernels_rrlu_w1.hip (one matrix per workgroup, wave 0) and the
// persistent half-sweep of kernels_walk.hip (wave 0 of the walking workgroup).
#pragma once
#include "kernels_rrlu_xcd_common.hpp"

namespace t4a {

namespace {

template <int NC, bool FACTORS> struct W1Lds {
    static constexpr int o_lc = 0;                                   // FACTORS: double [NC][64] pivot columns (un-scaled), by step and row
    static constexpr int o_ur = o_lc + (FACTORS ? NC * 64 * 8 : 0);  // FACTORS: double [NC][NC] rows of U, by step and column
    static constexpr int bytes = o_ur + (FACTORS ? NC * NC * 8 : 0) + 16;
};
// The permutation tables and the pivot values are LANE ARRAYS: entry i of a table sits in lane i of one register, read by
// v_readlane and written by v_writelane with scalar indices — no memory, no wait, no exec mask on the step's critical path.
__device__ __forceinline__ int w1_get(int table, int idx_s) { return __builtin_amdgcn_readlane(table, idx_s); }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void w1_set(int& table, int idx_s, int value_s)
{
    // (this compiler has no v_writelane builtin; two different scalar registers in one VALU instruction violate the constant-bus
    // rule, so the lane select goes through M0 — which the compiler never keeps a value in: its own uses, s_set_gpr_idx_on among
    // them, set it immediately before)
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(table) : "s"(value_s), "s"(idx_s) : "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ double w1_readlane_f64_ordered(double v, int lane_s)
{
    unsigned lo, hi;
    asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=s"(lo), "=s"(hi) : "v"(lo32(v)), "v"(hi32(v)), "s"(lane_s));
    return mk_f64(lo, hi);
}
// bit `lane` of a wave mask (scalar)
__device__ __forceinline__ unsigned w1_bit(unsigned long long mask, int lane_s) { return (unsigned)((mask >> lane_s) & 1ull); }

// phase stamps of a diagnostic build (-DT4A_XCD_STAMPS, tools/build_stamps_lib.sh): cycles (s_memtime) per phase summed over the
// steps in p.stamps[0..3] (search, stop tests + division + clearing, tables, update), launch phases in [16..18] (load + init, steps,
// write-out), and the launch in both clocks in [20] (s_memtime) and [21] (s_memrealtime, 100 MHz): their ratio is the shader clock.
#define W1STAMP(slot)                                                    \
    do {                                                                 \
        if (kXcdStamps && p.stamps) {                                    \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            st_acc[slot] += now_ - st_last;                              \
            st_last = now_;                                              \
        }                                                                \
    } while (0)

// Called by ONE wave (all 64 lanes active).  `lds`: W1Lds<NC, FACTORS>::bytes of LDS that only this wave touches.
// FACTORS: the factored matrix is written to p.Aout.  Returns the number of pivots (-1: gave up with code 2, -2: poisoned bond).
template <int NC, bool ROWMAJOR, bool FACTORS>
__device__ __forceinline__ int rrlu_w1_body(const RrluXcdArgs& p, char* lds)
{
    constexpr int GS = 8, NG = NC / GS;
    static_assert(NC % GS == 0 && NC <= 64, "columns of the one-wave kernel");
    using L = W1Lds<NC, FACTORS>;
    const int lane = threadIdx.x & 63;
    double* const lcol = reinterpret_cast<double*>(lds + L::o_lc);
    double* const urow = reinterpret_cast<double*>(lds + L::o_ur);
    const unsigned long long ts_begin = p.ts_u64 > 0 ? wall_clock64() : 0ull;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = kXcdStamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long st_t0 = st_last, st_r0 = kXcdStamps ? __builtin_amdgcn_s_memrealtime() : 0ull;

    int M = p.M, N = p.N, max_steps = p.max_steps;
    int lda = p.M;
    if (p.dims) {
        const int d0 = __builtin_amdgcn_readfirstlane(p.dims[0]), d1 = __builtin_amdgcn_readfirstlane(p.dims[1]);
        M = p.dims_swap ? d1 : d0;
        N = p.dims_swap ? d0 : d1;
        if (M > p.M || N > p.N) M = N = 0; // (cannot happen: the plan is made for upper bounds)
        const int mn = M < N ? M : N;
        max_steps = max_steps < mn ? max_steps : mn;
        if (mn <= 0) return -2; // poisoned bond: nothing to do (no completion token)
        lda = p.rowmap ? __builtin_amdgcn_readfirstlane(p.dims[3]) : M;
    }
    M = __builtin_amdgcn_readfirstlane(M);
    N = __builtin_amdgcn_readfirstlane(N);
    max_steps = __builtin_amdgcn_readfirstlane(max_steps);
    lda = __builtin_amdgcn_readfirstlane(lda);
    int posrow = lane, rowpos = lane, poscol = lane, colpos = lane; // position -> index and index -> position, rows and columns
    int pv_lo = 0, pv_hi = 0;                                       // pivot values by step

    // ---- column g GS + qq at element qq of group g; my row: lane ----
    xvec<GS> ag[NG];
    unsigned act[NG]; // bit qq: the column exists and is still in the trailing block (wave-uniform; only "group has a live column" is used)
    const int srow = p.rowmap ? p.rowmap[lane < M ? lane : 0] : lane;
    double local_absmax = 0.0;
    bool bad = false;
    const double* colptr = p.A + srow;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        act[g] = 0u;
#pragma unroll
        for (int qq = 0; qq < GS; ++qq) {
            const int c = g * GS + qq;
            if (c < N) act[g] |= 1u << qq;
            const bool ok = c < N && lane < M;
            double v = *(ok ? colptr : p.A);
            colptr += lda; // (a running per-lane pointer: column offsets as scalar constants would occupy 2 NC scalar registers)
            v = ok ? v : 0.0;
            local_absmax = vmax_abs(local_absmax, v);
            bad |= !((v - v) == 0.0);
            ag[g][qq] = v;
        }
    }
    if (__ballot(bad) != 0ull) { // NaN / infinity in the input: the caller runs the kernel that implements the NaN-incumbent rule
        if (lane == 0) {
            atomicExch(&p.iresult[1], 2);
            if (p.h_block) reinterpret_cast<volatile int*>(p.h_block)[5] = 2;
        }
        return -1;
    }
    // max sqrt(v*v) over the input (update_max_sample_value, tensorci2.rs:2009-2014): sqrt(v*v) == |v| while the square is a normal
    // number, and the maximum of the square roots is the square root of the largest square
    double in_max = wave_max_f64(local_absmax);
    if (!hi_mid((int)hi32(in_max))) in_max = uniform_f64(sqrt(in_max * in_max));

    int npiv = 0;
    double max_error = 0.0, error = __builtin_nan("");
    bool gave_up = false;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    const double rel_tol_v = p.rel_tol, abs_tol_v = p.abs_tol;

    // per-lane maxima of |a| per register group (padded and cleared entries are zeros)
    double mg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double m0 = 0.0, m1 = 0.0;
#pragma unroll
        for (int qq = 0; qq < GS; ++qq) {
            if (qq & 1) m1 = vmax_abs(m1, ag[g][qq]);
            else m0 = vmax_abs(m0, ag[g][qq]);
        }
        mg[g] = vmax(m0, m1);
    }

    W1STAMP(4);
    for (int kn = 0; kn < max_steps; ++kn) {
        const int k = kn - 1;
        // ---- the pivot: wave maximum, the lane that holds it, its register group, its column ----
        double m = mg[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) m = vmax(m, mg[g]);
        const int mhi = (int)hi32(m);
        const int whi = wave_max_i32(mhi);
        if (((whi >> 20) & 0x7FF) == 0x7FF) { // an infinite magnitude: overflow in the trailing block
            gave_up = true;
            break;
        }
        const unsigned long long whb = __ballot(mhi == whi);
        bool found = false;
        int hl = 0, pc = 0;
        double xl = 0.0, wval = 0.0, pivot_abs = 0.0, rp = 0.0;
        if (hi_mid(whi) && __builtin_popcountll(whb) == 1) { // the square is a normal number: distinct |v| <=> distinct scores
            hl = (int)__builtin_ctzll(whb);
            const double wmax = readlane_f64(m, hl);
            const double rpa = refined_rcp(wmax);
            unsigned gm = 1u; // bit g: the maximum of lane hl sits in register group g
            if (NG > 1) {
                gm = 0u;
#pragma unroll
                for (int g = 0; g < NG; ++g) gm |= w1_bit(__ballot(mg[g] == wmax), hl) << g;
            }
            if (__builtin_popcount(gm) == 1) {
                const int gs = (int)__builtin_ctz(gm);
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if (g == gs) {
                        unsigned bits = 0u; // bit GS - 1 - qq (of every lane): column qq of the group holds this lane's maximum
#pragma unroll
                        for (int qq = 0; qq < GS; ++qq)
                            asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"((double)ag[g][qq]), "s"(wmax) : "vcc");
                        const unsigned cm = (unsigned)__builtin_amdgcn_readlane((int)bits, hl);
                        if (__builtin_popcount(cm) == 1) {
                            const int qs = GS - 1 - (int)__builtin_ctz(cm);
                            asm volatile("; pivot column in group %0" ::"n"(g)); // (a real branch: no select over whole register groups)
                            xl = ag[g][qs];
                            pc = g * GS + qs;
                            found = true;
                        }
                    }
                if (found) {
                    wval = readlane_f64(xl, hl);
                    pivot_abs = wmax;
                    rp = mk_f64(lo32(rpa), hi32(rpa) | (hi32(wval) & 0x80000000u)); // refined_rcp(-p) == -refined_rcp(p) bit for bit
                }
            }
        }
        if (!found) {
            // ties, zero / subnormal maxima: exact sweep on (v*v, position) over the trailing block
            const double wm = wave_max_f64(m);
            const double sq = wm * wm;
            unsigned mypos = XNOPOS;
            double myval = 0.0;
            int myq = 0;
            const int lane_o = opaque_v(lane), M_o = opaque_s(M), N_o = opaque_s(N);
            const unsigned rp_ = (unsigned)rowpos;
            const bool row_ok = (lane_o < M_o) & ((int)rp_ > k);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (__ballot(mg[g] * mg[g] == sq) != 0ull) {
#pragma unroll
                    for (int qq = 0; qq < GS; ++qq)
                        if (g * GS + qq < N_o) {
                            const unsigned cp_ = (unsigned)w1_get(colpos, g * GS + qq);
                            const unsigned key = ROWMAJOR ? ((rp_ << 10) | cp_) : ((cp_ << 10) | rp_);
                            const double av = ag[g][qq];
                            const bool hit = row_ok & ((int)cp_ > k) & (av * av == sq);
                            if (hit && key < mypos) {
                                mypos = key;
                                myval = av;
                                myq = g * GS + qq;
                            }
                        }
                }
            }
            const unsigned wp = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(mypos));
            if (wp == XNOPOS) break; // (cannot happen while kn < min(M, N): the trailing block is not empty)
            const unsigned long long sel = __ballot(mypos == wp);
            hl = (int)__builtin_ctzll(sel);
            wval = readlane_f64(myval, hl);
            pc = __builtin_amdgcn_readlane(myq, hl);
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g == pc / GS) {
                    asm volatile("; pivot column in group %0 (exact sweep)" ::"n"(g));
                    xl = ag[g][pc - g * GS];
                }
            const double wsq = wval * wval;
            pivot_abs = __builtin_fabs(wval);
            if (!(wsq >= 2.2250738585072014e-308 && wsq < __builtin_huge_val())) pivot_abs = sqrt(mk_f64((unsigned)opaque_v((int)lo32(wsq)), hi32(wsq)));
            pivot_abs = uniform_f64(pivot_abs);
            rp = refined_rcp(wval);
        }
        W1STAMP(0);
        // stop tests on the pivot magnitude sqrt(v*v), in the reference's order (matrixlu.rs:757-781)
        error = pivot_abs;
        if (kn > 0 && (pivot_abs < rel_tol_v * max_error || pivot_abs < abs_tol_v)) break; // pivot kn is not applied
        if (pivot_abs <= min_pivot_abs) break;
        max_error = fmax(max_error, pivot_abs);
        // ---- the pivot column leaves the registers (with FACTORS: un-scaled into the LDS, and the pivot row with it) ----
        if constexpr (FACTORS) {
            lcol[kn * 64 + lane] = xl;
            if (lane == hl) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int qq = 0; qq < GS; ++qq) urow[kn * NC + g * GS + qq] = ag[g][qq];
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (NG == 1 || g == pc / GS) {
                if (NG > 1) asm volatile("; clear in group %0" ::"n"(g));
                ag[g][pc - g * GS] = 0.0;
                act[g] &= ~(1u << (pc - g * GS));
            }
        // l = column / pivot, bitwise the IEEE quotient
        double l;
        {
            const double x = xl;
            const double q0 = x * rp;
            const double qf = __builtin_fma(__builtin_fma(-wval, q0, x), rp, q0);
            l = (x == 0.0) ? q0 : qf;
            const bool slow = !exp_mid(wval) | ((__builtin_fabs(x) < 4.909093465297727e-91) & (x != 0.0)); // 2^-300 (|x| <= |pivot|: full pivoting)
            if (__ballot(slow) != 0ull) {
                if (!(exp_mid(wval) && (exp_mid(x) || x == 0.0))) l = x / wval;
            }
        }
        // ---- permutation tables (swap_rows / swap_cols of the reference as index tables) ----
        W1STAMP(1);
        {
            const int rk_ = w1_get(posrow, kn), ck_ = w1_get(poscol, kn);
            const int prp = w1_get(rowpos, hl), pcp = w1_get(colpos, pc);
            w1_set(posrow, prp, rk_);
            w1_set(posrow, kn, hl);
            w1_set(rowpos, rk_, prp);
            w1_set(rowpos, hl, kn);
            w1_set(poscol, pcp, ck_);
            w1_set(poscol, kn, pc);
            w1_set(colpos, ck_, pcp);
            w1_set(colpos, pc, kn);
            w1_set(pv_lo, kn, (int)lo32(wval));
            w1_set(pv_hi, kn, (int)hi32(wval));
        }
        W1STAMP(2);
        // ---- pivot row (finished row kn of U) by v_readlane, rank-1 update fused with the maxima for the next arg-max.  The
        // pivot row has l = 1: x - 1.0 x leaves exact zeros; rows pivoted before hold zeros in every live column, so their l is 0
        // and they stay 0; cleared columns stay 0.
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (NG > 1 && act[g] == 0u) {
                mg[g] = 0.0;
                continue;
            }
            double m0 = 0.0, m1 = 0.0;
#pragma unroll
            for (int qq = 0; qq < GS; ++qq) {
                const double uq = w1_readlane_f64_ordered(ag[g][qq], hl);
                double t = ag[g][qq];
                sub_in_place(t, l * uq);
                ag[g][qq] = t;
                if (qq & 1) m1 = vmax_abs(m1, t);
                else m0 = vmax_abs(m0, t);
            }
            mg[g] = vmax(m0, m1);
        }
        npiv = kn + 1;
        W1STAMP(3);
    }
    W1STAMP(5);

    // ---- results ----
    if (gave_up) {
        if (lane == 0) {
            atomicExch(&p.iresult[1], 2);
            if (p.h_block) reinterpret_cast<volatile int*>(p.h_block)[5] = 2;
        }
        return -1;
    }
    if (npiv >= (M < N ? M : N)) error = 0.0; // matrixlu.rs:811-813
    {
        int* const h_rp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.row_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        int* const h_cp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.col_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        if (lane < M) {
            const int v = posrow;
            p.row_perm[lane] = v;
            if (h_rp) h_rp[lane] = v;
        }
        if (lane < N) {
            const int v = poscol;
            p.col_perm[lane] = v;
            if (h_cp) h_cp[lane] = v;
        }
        unsigned long long* const h_pv = p.h_block ? p.h_block + (reinterpret_cast<const char*>(p.pivot_vals) - reinterpret_cast<const char*>(p.dresult)) / 8 : nullptr;
        if (lane < npiv) {
            const double v = mk_f64((unsigned)pv_lo, (unsigned)pv_hi);
            p.pivot_vals[lane] = v;
            if (h_pv) h_pv[lane] = (unsigned long long)__double_as_longlong(v);
        }
    }
    // factored matrix in permuted coordinates: rows of U and the pivot columns from the LDS (L scaled now: scale_column_tail,
    // matrixlu.rs:562-577), the untouched trailing block from the registers.  (l = x / pivot with |x| <= |pivot| and finite
    // operands: no NaN can appear in the factors of a launch that did not give up.)
    if constexpr (FACTORS) {
        if (p.Aout) {
            const int rp_ = rowpos;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int qq = 0; qq < GS; ++qq) {
                    const int c = g * GS + qq;
                    if (c < N) {
                        const int cp = w1_get(colpos, c);
                        const int cpc = cp < npiv ? cp : 0;
                        const double pv = mk_f64((unsigned)w1_get(pv_lo, cpc), (unsigned)w1_get(pv_hi, cpc));
                        if (lane < M) {
                            double v = ag[g][qq];
                            if (rp_ < npiv && cp >= rp_) v = urow[rp_ * NC + c];
                            else if (cp < npiv) v = xcd_div(lcol[cp * 64 + lane], pv, refined_rcp(pv), exp_mid(pv));
                            if (p.out_transposed) p.Aout[(size_t)rp_ * N + cp] = v;
                            else p.Aout[(size_t)cp * M + rp_] = v;
                        }
                    }
                }
        }
    }
    if (lane == 0) {
        p.iresult[0] = npiv;
        p.dresult[0] = error;
        const unsigned long long maxbits = (unsigned long long)__double_as_longlong(in_max > 0.0 ? in_max : 0.0);
        if (p.h_block) {
            p.h_block[0] = (unsigned long long)__double_as_longlong(error);
            p.h_block[1] = maxbits;
            ((volatile int*)p.h_block)[4] = npiv;
            ((volatile int*)p.h_block)[7] = (int)p.salt; // completion token
            if (p.ts_u64 > 0) {
                p.h_block[p.ts_u64] = ts_begin;
                p.h_block[p.ts_u64 + 1] = wall_clock64();
            }
            reinterpret_cast<unsigned long long*>(p.dresult)[1] = 0ull; // clean header for the next launch
        } else {
            reinterpret_cast<unsigned long long*>(p.dresult)[1] = maxbits; // (one wave: the header's maximum needs no atomic)
            if (p.dims && p.ts_u64 > 0) {
                unsigned long long* const blk = reinterpret_cast<unsigned long long*>(p.dresult);
                blk[p.ts_u64] = ts_begin;
                blk[p.ts_u64 + 1] = wall_clock64();
            }
        }
    }
    if (kXcdStamps && p.stamps) {
        W1STAMP(6);
        if (lane == 0) {
            for (int i = 0; i < 4; ++i) p.stamps[i] = st_acc[i];
            p.stamps[16] = st_acc[4];
            p.stamps[17] = st_acc[5];
            p.stamps[18] = st_acc[6];
            p.stamps[20] = __builtin_amdgcn_s_memtime() - st_t0;
            p.stamps[21] = __builtin_amdgcn_s_memrealtime() - st_r0;
        }
    }
    if (p.dims) { // the next kernel of a chain checks the token: everything above is out first
        __threadfence();
        if (lane == 0) p.iresult[3] = (int)p.salt;
    }
    return npiv;
}

} // namespace

} // namespace t4a
