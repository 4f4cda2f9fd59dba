// kernels_rrlu_wg.hip — K2 for matrices that fit ONE workgroup (round 4): register-resident full-pivot rrLU with the per-step
// exchange through the LDS of one compute unit.  Up to 8 waves x 64 lanes x 96 values: 64 x 512, 128 x 384.
//
// Why: the single-XCD kernels pay ~2 us per pivot step at ANY size (two L2 hand-offs between 29 compute units) and the
// single-workgroup plan of the chip-wide kernel 3 100 - 3 600 cycles; the end bonds of every chain, the whole growth phase of a
// patch and all of a rank-2 problem run at that price.  One compute unit holds these matrices
// in its registers and hands the candidate column over through its LDS: ONE barrier per pivot step, no polling, no tags.
//
// Same contract as the other rrLU kernels: bit-identical to rrlu_mut (tensor4all-core/src/matrixlu.rs:735-819; arg-max
// matrixlu.rs:480-519: key v*v, first strict maximum in column-major order of the permuted trailing block; stop rules :757-781;
// un-fused update :593-612); a right-orthogonal factorisation runs as the left-orthogonal one of A^T with the row-major tie
// order.  Arguments, result block, bond-chain extensions (device-side dimensions, row map, speculative candidate matrix of the
// next bond by the other workgroups of the launch, completion token) are those of the single-XCD kernel (RrluXcdArgs).
// Non-finite values are not handled (as in kernels_rrlu_xcd2.hip): the launch gives up with code 2 and the caller runs the
// first-generation single-XCD kernel.
//
// Structure of a step (wave w owns columns w + 8 q, lane l rows l + 64 r; columns in register groups of GS so that the two
// run-time accesses — the candidate's value and its column — index inside one vector):
//   all waves: wave maximum of |a| over the trailing block (kept per lane and register group by the update) -> position of the
//   candidate -> key {value, row, column slot} and the candidate's COLUMN into the LDS  -> barrier ->
//   all waves (redundantly, no second barrier): read the 8 keys, pick the winner, stop tests, read the winner's column, divide
//   (l of the pivot row is pivot / pivot = 1: the update zeroes that row exactly, which is all the masking there is), own copy
//   of the permutation tables, pivot row by v_readlane, rank-1 update fused with the maxima for the next step.
#include "kernels_rrlu_xcd_common.hpp"

#include <type_traits>

namespace t4a {

namespace {

constexpr int WGW = 8;         // waves per workgroup
constexpr int WGT = 64 * WGW;  // threads
template <int RPT> constexpr int wg_gs() { return 8; } // columns per register group (GS * RPT <= 16 doubles: one VGPR tuple; wider tuples — three rows per lane — spill)

template <int RPT, int CPW> struct WgLds {
    static constexpr int MP = 64 * RPT;                          // padded rows
    static constexpr int NP = WGW * CPW;                         // padded columns
    static constexpr int o_keys = 0;                             // u32x4 [2][8]: {value lo, value hi, meta, 0} per step parity and wave
    static constexpr int o_col = o_keys + 2 * WGW * 16;          // double [2][8][MP]: the candidate column of every wave
    static constexpr int o_ctl = o_col + 2 * WGW * MP * 8;       // int [16]: [0] non-finite input [8] tile of the speculative work
    static constexpr int o_pv = o_ctl + 64;                      // double [MP] pivot values
    static constexpr int o_tab = o_pv + MP * 8;                  // per wave: u16 posrow[MP] rowpos[MP] poscol[NP] colpos[NP]
    static constexpr int tab_bytes = (2 * MP + 2 * NP) * 2;
    static constexpr int bytes = o_tab + WGW * tab_bytes;
};
__device__ __forceinline__ double readlane_f64_ordered(double v, int lane_s)
{
    unsigned lo, hi;
    asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=s"(lo), "=s"(hi) : "v"(lo32(v)), "v"(hi32(v)), "s"(lane_s));
    return mk_f64(lo, hi);
}
__device__ __forceinline__ void sub_in_place_ordered(double& a, double prod)
{
    asm volatile("v_add_f64 %0, %0, -%1" : "+v"(a) : "v"(prod));
}
// key meta word: bits 0..7 row index of the candidate, 8..14 column slot q of the publishing wave, bit 16: has a candidate
constexpr unsigned WG_META_VALID = 1u << 16;

template <int RPT, int CPW, bool ROWMAJOR>
__device__ __forceinline__ void rrlu_wg_body(const RrluXcdArgs& p)
{
    constexpr int GS = wg_gs<RPT>(), NG = CPW / GS, GE = GS * RPT;
    static_assert(RPT <= 2 && CPW % GS == 0 && GE <= 16 && RPT * CPW <= 96, "plan family of the one-workgroup kernel");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = WgLds<RPT, CPW>;
    constexpr int MP = L::MP, NP = L::NP;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* const ctl = reinterpret_cast<int*>(smem_raw + L::o_ctl);
    double* const lds_pivots = reinterpret_cast<double*>(smem_raw + L::o_pv);
    unsigned short* const posrow = reinterpret_cast<unsigned short*>(smem_raw + L::o_tab + wave * L::tab_bytes);
    unsigned short* const rowpos = posrow + MP;
    unsigned short* const poscol = rowpos + MP;
    unsigned short* const colpos = poscol + NP;
    const unsigned long long ts_begin = p.ts_u64 > 0 ? wall_clock64() : 0ull;

    int M = p.M, N = p.N, max_steps = p.max_steps;
    int lda = p.M;
    if (p.dims) {
        const int d0 = __builtin_amdgcn_readfirstlane(p.dims[0]), d1 = __builtin_amdgcn_readfirstlane(p.dims[1]);
        M = p.dims_swap ? d1 : d0;
        N = p.dims_swap ? d0 : d1;
        if (M > p.M || N > p.N) M = N = 0; // (cannot happen: the plan is made for upper bounds)
        const int mn = M < N ? M : N;
        max_steps = max_steps < mn ? max_steps : mn;
        if (mn <= 0) return; // poisoned bond: nothing to do (no completion token)
        lda = p.rowmap ? __builtin_amdgcn_readfirstlane(p.dims[3]) : M;
    }
    M = __builtin_amdgcn_readfirstlane(M);
    N = __builtin_amdgcn_readfirstlane(N);
    max_steps = __builtin_amdgcn_readfirstlane(max_steps);
    lda = __builtin_amdgcn_readfirstlane(lda);
    if (tid == 0) ctl[0] = 0;
    for (int i = lane; i < M; i += 64) {
        posrow[i] = (unsigned short)i;
        rowpos[i] = (unsigned short)i;
    }
    for (int j = lane; j < N; j += 64) {
        poscol[j] = (unsigned short)j;
        colpos[j] = (unsigned short)j;
    }

    // ---- my columns: wave + 8 (g GS + qq); my rows: lane + 64 r; element (qq, r) of group g at index qq * RPT + r ----
    xvec<GE> ag[NG];
    unsigned act[NG]; // bit qq: the column exists and is still in the trailing block (wave-uniform)
    int srow[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = lane + 64 * r;
        srow[r] = i;
        if (p.rowmap) srow[r] = p.rowmap[i < M ? i : 0];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        act[g] = 0u;
#pragma unroll
        for (int qq = 0; qq < GS; ++qq) {
            const int c = wave + WGW * (g * GS + qq);
            if (c < N) act[g] |= 1u << qq;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int i = lane + 64 * r;
                const bool ok = c < N && i < M;
                ag[g][qq * RPT + r] = p.A[ok ? (size_t)c * lda + srow[r] : (size_t)0];
            }
        }
    }
    double local_sqmax = 0.0;
    bool bad = false;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int qq = 0; qq < GS; ++qq)
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int c = wave + WGW * (g * GS + qq), i = lane + 64 * r;
                const double v = (c < N && i < M) ? (double)ag[g][qq * RPT + r] : 0.0;
                const double sqv = v * v;
                if (sqv > local_sqmax) local_sqmax = sqv;
                bad |= !((v - v) == 0.0);
                ag[g][qq * RPT + r] = v;
            }
    {
        const double wm = wave_max_f64(sqrt(local_sqmax));
        if (lane == 0 && wm > 0.0) atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(wm));
    }
    __syncthreads();
    if (__ballot(bad) != 0ull && lane == 0) ctl[0] = 1;
    __syncthreads();
    if (ctl[0]) { // NaN / infinity in the input: the caller runs the kernel that implements the NaN-incumbent rule
        if (tid == 0) {
            atomicExch(&p.iresult[1], 2);
            if (p.h_block) reinterpret_cast<volatile int*>(p.h_block)[5] = 2;
        }
        return;
    }

    int npiv = 0;
    double max_error = 0.0, error = __builtin_nan("");
    bool gave_up = false;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    const double rel_tol_v = p.rel_tol, abs_tol_v = p.abs_tol;

    // per-lane maxima of |a| per register group over the columns of the trailing block (-1: none)
    double mg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double m0 = -1.0, m1 = -1.0;
#pragma unroll
        for (int qq = 0; qq < GS; ++qq)
            if (act[g] & (1u << qq)) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    if ((qq * RPT + r) & 1) m1 = vmax_abs(m1, ag[g][qq * RPT + r]);
                    else m0 = vmax_abs(m0, ag[g][qq * RPT + r]);
                }
            }
        mg[g] = vmax(m0, m1);
    }

    for (int kn = 0; kn < max_steps; ++kn) {
        const int k = kn - 1;
        const int par = kn & 1;
        // ---- my candidate: wave maximum, then its position ----
        double m = mg[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) m = vmax(m, mg[g]);
        const int whi = wave_max_i32((int)hi32(m));
        const unsigned long long whb = __ballot((int)hi32(m) == whi);
        const double wmax = (__builtin_popcountll(whb) == 1) ? readlane_f64(m, (int)__builtin_ctzll(whb)) : wave_max_f64(m);
        bool has_cand = false;
        double cval = 0.0;
        int cirow = 0, qstar = 0;
        xvec<RPT> xc; // the candidate's column (my rows)
#pragma unroll
        for (int r = 0; r < RPT; ++r) xc[r] = 0.0;
        if (wmax >= 0.0) {
            bool done = false;
            if (hi_mid(whi)) { // the square is a normal number: distinct |v| <=> distinct scores, the sweep compares |v| itself
                int nhit = 0, ghit = -1, hl = 0; // (no array of ballots: scalar registers are scarce in the wide instantiations)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const unsigned long long bgm = __ballot(mg[g] == wmax);
                    const int c = __builtin_popcountll(bgm);
                    nhit += c;
                    if (c) {
                        ghit = g;
                        hl = (int)__builtin_ctzll(bgm);
                    }
                }
                if (nhit == 1) { // one lane of one register group holds the maximum: the normal case
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        if (g == ghit) {
                            // bit GE - 1 - e of `bits`: element e of the group holds the maximum (columns outside the trailing
                            // block keep stale values: they are skipped, their bits stay 0)
                            unsigned bits = 0u;
#pragma unroll
                            for (int qq = 0; qq < GS; ++qq) {
                                if (act[g] & (1u << qq)) {
#pragma unroll
                                    for (int r = 0; r < RPT; ++r)
                                        asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"((double)ag[g][qq * RPT + r]), "s"(wmax) : "vcc");
                                } else {
                                    bits <<= RPT;
                                }
                            }
                            const unsigned hb_ = (unsigned)__builtin_amdgcn_readlane((int)bits, hl);
                            if (__builtin_popcount(hb_) == 1) {
                                const int es = GE - 1 - (int)__builtin_ctz(hb_);
                                const int qq = es / RPT, rstar = es - qq * RPT;
                                cirow = hl + 64 * rstar;
                                qstar = g * GS + qq;
                                cval = readlane_f64(ag[g][es], hl);
#pragma unroll
                                for (int r = 0; r < RPT; ++r) xc[r] = ag[g][qq * RPT + r];
                                has_cand = true;
                                done = true;
                            }
                        }
                }
            }
            if (!done) {
                // ties, zero / subnormal scores (an infinite score ends the launch at the pick): exact sweep on (v*v, position)
                const double sq = wmax * wmax;
                unsigned mypos = XNOPOS;
                double myval = 0.0;
                int myrow = 0, myq = 0;
                const int lane_o = opaque_v(lane), M_o = opaque_s(M);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const bool ghit = (mg[g] >= 0.0) & (mg[g] * mg[g] == sq);
                    if (__ballot(ghit) != 0ull) {
#pragma unroll
                        for (int qq = 0; qq < GS; ++qq)
                            if (act[g] & (1u << qq)) {
                                const unsigned cp_ = colpos[opaque_s(wave + WGW * (g * GS + qq))];
#pragma unroll
                                for (int r = 0; r < RPT; ++r) {
                                    const int i = lane_o + 64 * r;
                                    const unsigned rp_ = rowpos[i < M_o ? i : 0];
                                    const unsigned key = ROWMAJOR ? ((rp_ << 10) | cp_) : ((cp_ << 10) | rp_);
                                    const double av = ag[g][qq * RPT + r];
                                    const bool hit = ghit & (i < M_o) & ((int)rp_ > k) & (av * av == sq);
                                    if (hit && key < mypos) {
                                        mypos = key;
                                        myval = av;
                                        myrow = i;
                                        myq = g * GS + qq;
                                    }
                                }
                            }
                    }
                }
                const unsigned wp = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(mypos));
                if (wp != XNOPOS) {
                    const unsigned long long sel = __ballot(mypos == wp);
                    const int hl = (int)__builtin_ctzll(sel);
                    has_cand = true;
                    cval = readlane_f64(myval, hl);
                    cirow = __builtin_amdgcn_readlane(myrow, hl);
                    qstar = __builtin_amdgcn_readlane(myq, hl);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        if (g == qstar / GS) {
                            const int qq = qstar - g * GS;
#pragma unroll
                            for (int r = 0; r < RPT; ++r) xc[r] = ag[g][qq * RPT + r];
                        }
                }
            }
        }
        // ---- key and candidate column into the LDS ----
        {
            u32x4 kv;
            kv.x = lo32(cval);
            kv.y = hi32(cval);
            kv.z = (unsigned)cirow | ((unsigned)qstar << 8) | (has_cand ? WG_META_VALID : 0u);
            kv.w = 0u;
            if (lane == 0) *reinterpret_cast<u32x4*>(smem_raw + L::o_keys + (par * WGW + wave) * 16) = kv;
            if (has_cand) {
                double* const cb = reinterpret_cast<double*>(smem_raw + L::o_col) + (par * WGW + wave) * MP + lane;
#pragma unroll
                for (int r = 0; r < RPT; ++r) cb[64 * r] = xc[r];
            }
        }
        __syncthreads();

        // ---- every wave picks the winner itself (matrixlu.rs:480-519 across the waves) ----
        const u32x4 key = *reinterpret_cast<const u32x4*>(smem_raw + L::o_keys + (par * WGW + (lane & 7)) * 16);
        const int mh = (key.z & WG_META_VALID) ? (int)(key.y & 0x7FFFFFFFu) : -1;
        int wst = 0;
        {
            int v = mh;
            v = dpp_max_i32<0xB1>(v);  // quad_perm [1,0,3,2]
            v = dpp_max_i32<0x4E>(v);  // quad_perm [2,3,0,1]
            v = dpp_max_i32<0x141>(v); // row_half_mirror: lanes 0..7 hold the maximum over the eight keys
            const int ghi = __builtin_amdgcn_readfirstlane(v);
            const unsigned hb = (unsigned)(__ballot(mh == ghi) & 0xFFull);
            if (hi_mid(ghi) && __builtin_popcount(hb) == 1) {
                wst = (int)__builtin_ctz(hb);
            } else if (ghi >= 0 && ((ghi >> 20) & 0x7FF) == 0x7FF) {
                gave_up = true; // an infinite magnitude: overflow in the trailing block
            } else {
                // ties between waves, zero / subnormal scores: exact comparison of (v*v, position) over the eight keys; a wave
                // without candidate carries value 0 and the largest position key
                const bool in8 = lane < 8, valid = (key.z & WG_META_VALID) != 0u;
                const double kvv = mk_f64(key.x, key.y);
                const unsigned rp_ = rowpos[key.z & 255u], cp_ = colpos[min((lane & 7) + WGW * (int)((key.z >> 8) & 127u), N - 1)];
                const unsigned pkey = ROWMAJOR ? ((rp_ << 10) | cp_) : ((cp_ << 10) | rp_);
                const double sc = in8 ? kvv * kvv : -2.0;
                const unsigned pk = (in8 && valid) ? pkey : XNOPOS;
                const double gmax = wave_max_f64(sc);
                const unsigned gpos = wave_min_u32((sc == gmax) ? pk : XNOPOS);
                const unsigned long long sel = __ballot(in8 & (sc == gmax) & (pk == gpos));
                wst = sel ? (int)__builtin_ctzll(sel) : 0;
            }
        }
        if (gave_up) break;
        const double wval = mk_f64((unsigned)__builtin_amdgcn_readlane((int)key.x, wst), (unsigned)__builtin_amdgcn_readlane((int)key.y, wst));
        const unsigned wmeta = (unsigned)__builtin_amdgcn_readlane((int)key.z, wst);
        const int irow_p = (int)(wmeta & 255u);
        const int qsl = (int)((wmeta >> 8) & 127u);
        const int pc = wst + WGW * qsl; // original index of the pivot column
        // ---- the winner's column -> l = column / pivot (every wave for itself) ----
        double xl[RPT];
        {
            const double* const cb = reinterpret_cast<const double*>(smem_raw + L::o_col) + (par * WGW + wst) * MP + lane;
#pragma unroll
            for (int r = 0; r < RPT; ++r) xl[r] = cb[64 * r];
        }
        // stop tests on the pivot magnitude sqrt(v*v), in the reference's order (matrixlu.rs:757-781)
        {
            const double wsq = wval * wval;
            double pivot_abs = __builtin_fabs(wval);
            if (!(wsq >= 2.2250738585072014e-308 && wsq < __builtin_huge_val())) pivot_abs = sqrt(mk_f64((unsigned)opaque_v((int)lo32(wsq)), hi32(wsq)));
            pivot_abs = uniform_f64(pivot_abs);
            error = pivot_abs;
            bool stop = false;
            if (kn > 0 && (pivot_abs < rel_tol_v * max_error || pivot_abs < abs_tol_v)) stop = true;
            else if (pivot_abs <= min_pivot_abs) stop = true;
            else max_error = fmax(max_error, pivot_abs);
            if (stop) break; // pivot kn is not applied
        }
        const bool p_mid = exp_mid(wval);
        const double rp = refined_rcp(wval);
        xvec<RPT> l;
        {
            bool slow = !p_mid;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const double x = xl[r];
                const double q0 = x * rp;
                const double qf = __builtin_fma(__builtin_fma(-wval, q0, x), rp, q0);
                l[r] = (x == 0.0) ? q0 : qf;
                slow |= (__builtin_fabs(x) < 4.909093465297727e-91) & (x != 0.0); // 2^-300 (|x| <= |pivot|: full pivoting)
            }
            if (__ballot(slow) != 0ull) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (!(p_mid & (exp_mid(xl[r]) | (xl[r] == 0.0)))) l[r] = xl[r] / wval;
            }
        }
        // ---- my copy of the permutation tables (swap_rows / swap_cols of the reference as index tables) ----
        if (lane == 0) {
            const int rk_ = posrow[kn], ck_ = poscol[kn];
            const int prp = rowpos[irow_p], pcp = colpos[pc];
            posrow[prp] = (unsigned short)rk_;
            posrow[kn] = (unsigned short)irow_p;
            rowpos[rk_] = (unsigned short)prp;
            rowpos[irow_p] = (unsigned short)kn;
            poscol[pcp] = (unsigned short)ck_;
            poscol[kn] = (unsigned short)pc;
            colpos[ck_] = (unsigned short)pcp;
            colpos[pc] = (unsigned short)kn;
            if (wave == 0) lds_pivots[kn] = wval;
        }
        const int ls = irow_p & 63, rs = irow_p >> 6;
        if (wave == wst) {
            // the pivot column leaves the trailing block (its registers keep the un-scaled column: L is formed at the write-out)
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g == qsl / GS) act[g] &= ~(1u << (qsl - g * GS));
            if (p.urows && lane == ls) p.urows[(unsigned)(kn * N + pc)] = wval;
        }
        // ---- pivot row (finished row kn of U) by v_readlane, rank-1 update fused with the maxima for the next arg-max.  The
        // pivot row has l = 1: x - 1.0 x leaves exact zeros; rows pivoted before hold zeros in every column of the trailing
        // block, so their l is 0 and they stay 0 (no row mask anywhere).
        auto eliminate = [&](auto rs_c) {
            constexpr int RS = decltype(rs_c)::value;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double m0 = -1.0, m1 = -1.0;
#pragma unroll
                for (int qq = 0; qq < GS; ++qq)
                    if (act[g] & (1u << qq)) {
                        // (volatile: the columns stay apart — left to itself the scheduler hoists every readlane to the top of
                        // the loop and keeps 2 CPW scalar registers alive at once)
                        const double uq = readlane_f64_ordered(ag[g][qq * RPT + RS], ls);
                        if (p.urows && lane == ls) p.urows[(unsigned)(kn * N + (wave + WGW * (g * GS + qq)))] = uq;
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            double t = ag[g][qq * RPT + r];
                            if (r == 0) sub_in_place_ordered(t, l[r] * uq);
                            else sub_in_place(t, l[r] * uq);
                            ag[g][qq * RPT + r] = t;
                            if ((qq * RPT + r) & 1) m1 = vmax_abs(m1, t);
                            else m0 = vmax_abs(m0, t);
                        }
                    }
                mg[g] = vmax(m0, m1);
            }
        };
        if (rs == 0) eliminate(std::integral_constant<int, 0>{});
        if constexpr (RPT > 1) if (rs == 1) eliminate(std::integral_constant<int, 1>{});
        npiv = kn + 1;
    }

    // ---- results ----
    __syncthreads();
    if (gave_up) {
        if (tid == 0) {
            atomicExch(&p.iresult[1], 2);
            if (p.h_block) reinterpret_cast<volatile int*>(p.h_block)[5] = 2;
        }
        return;
    }
    if (npiv >= (M < N ? M : N)) error = 0.0; // matrixlu.rs:811-813
    if (tid == 0) {
        p.iresult[0] = npiv;
        p.dresult[0] = error;
    }
    {
        int* const h_rp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.row_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        int* const h_cp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.col_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        const unsigned short* const posrow0 = reinterpret_cast<const unsigned short*>(smem_raw + L::o_tab); // (all eight copies are equal)
        const unsigned short* const poscol0 = posrow0 + 2 * MP;
        for (int i = tid; i < M; i += WGT) {
            const int v = posrow0[i];
            p.row_perm[i] = v;
            if (h_rp) h_rp[i] = v;
        }
        for (int j = tid; j < N; j += WGT) {
            const int v = poscol0[j];
            p.col_perm[j] = v;
            if (h_cp) h_cp[j] = v;
        }
        unsigned long long* const h_pv = p.h_block ? p.h_block + (reinterpret_cast<const char*>(p.pivot_vals) - reinterpret_cast<const char*>(p.dresult)) / 8 : nullptr;
        for (int e = tid; e < npiv; e += WGT) {
            const double v = lds_pivots[e];
            p.pivot_vals[e] = v;
            if (h_pv) h_pv[e] = (unsigned long long)__double_as_longlong(v);
        }
    }
    // factored matrix in permuted coordinates: rows of U from the side buffer, L (scaled now: scale_column_tail,
    // matrixlu.rs:562-577) and the untouched trailing block from the registers
    int nan_seen = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int qq = 0; qq < GS; ++qq)
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int c = wave + WGW * (g * GS + qq), i = lane + 64 * r;
                if (c < N && i < M) {
                    const int cp = colpos[c], rp_ = rowpos[i];
                    const bool from_u = (rp_ < npiv) && (cp >= rp_);
                    double v = ag[g][qq * RPT + r];
                    const bool in_l = (cp < npiv) && (rp_ > cp);
                    if (in_l) {
                        const double pv = lds_pivots[cp];
                        v = xcd_div(v, pv, refined_rcp(pv), exp_mid(pv));
                        if (v != v) nan_seen = 1;
                    }
                    if (p.Aout) {
                        if (from_u)
                            v = __longlong_as_double((long long)__hip_atomic_load(
                                reinterpret_cast<const unsigned long long*>(p.urows) + ((size_t)rp_ * N + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (p.out_transposed) p.Aout[(size_t)rp_ * N + cp] = v;
                        else p.Aout[(size_t)cp * M + rp_] = v;
                    }
                }
            }
    if (nan_seen) {
        atomicExch(&p.iresult[2], 1);
        if (p.h_block) ((volatile int*)p.h_block)[6] = 1;
    }
    __syncthreads(); // (every wave's atomicMax / flag is out before the header is closed)
    if (p.h_block && tid == 0) {
        p.h_block[0] = (unsigned long long)__double_as_longlong(error);
        p.h_block[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.dresult) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((volatile int*)p.h_block)[4] = npiv;
        ((volatile int*)p.h_block)[7] = (int)p.salt; // completion token
        if (p.ts_u64 > 0) {
            p.h_block[p.ts_u64] = ts_begin;
            p.h_block[p.ts_u64 + 1] = wall_clock64();
        }
        reinterpret_cast<unsigned long long*>(p.dresult)[1] = 0ull; // clean header for the next launch
        if (p.dims) {
            __threadfence();
            p.iresult[3] = (int)p.salt;
        }
    }
    if (!p.h_block && p.dims && tid == 0) {
        if (p.ts_u64 > 0) {
            unsigned long long* const blk = reinterpret_cast<unsigned long long*>(p.dresult);
            blk[p.ts_u64] = ts_begin;
            blk[p.ts_u64 + 1] = wall_clock64();
        }
        __threadfence();
        p.iresult[3] = (int)p.salt;
    }
}

#ifndef T4A_WG_GROUP_TU
// solo launch: workgroup 0 factorises, the others (bond chain) evaluate the next bond's candidate matrix
template <int RPT, int CPW, bool ROWMAJOR>
__global__ void __launch_bounds__(WGT) __attribute__((amdgpu_waves_per_eu(2, 2))) rrlu_wg_kernel(RrluXcdArgs p)
{
    if (blockIdx.x != 0) {
        extern __shared__ __attribute__((aligned(16))) char smem_raw[];
        if (p.spec.out && p.dims) {
            const int m_spec = p.dims[2] != 0 ? 0 : (p.dims_swap ? p.dims[1] : p.dims[0]);
            if (m_spec > 0 && m_spec <= p.M)
                xcd_spec_work(reinterpret_cast<const XcdSpecArgs*>(kernarg_base() + offsetof(RrluXcdArgs, spec)), m_spec,
                              reinterpret_cast<int*>(smem_raw + WgLds<RPT, CPW>::o_ctl) + 8);
        }
        return;
    }
    rrlu_wg_body<RPT, CPW, ROWMAJOR>(p);
}
#else
// group launch: workgroup x factorises slot x (a slot with xcc = -1 is empty)
template <int RPT, int CPW, bool ROWMAJOR>
__global__ void __launch_bounds__(WGT) __attribute__((amdgpu_waves_per_eu(2, 2))) rrlu_wg_group_kernel(RrluXcdGroupArgs g)
{
    (void)g;
    const RrluXcdArgs& p = *reinterpret_cast<const RrluXcdArgs*>(kernarg_base() + (size_t)blockIdx.x * sizeof(RrluXcdArgs));
    if (p.xcc < 0) return;
    rrlu_wg_body<RPT, CPW, ROWMAJOR>(p);
}

#endif

template <int RPT, int CPW, bool ROWMAJOR> void wg_launch_one(const RrluXcdPlan& plan, const RrluXcdArgs* solo, const RrluXcdGroupArgs* group, hipStream_t stream)
{
    static std::once_flag attr_once;
#ifndef T4A_WG_GROUP_TU
    (void)group;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_wg_kernel<RPT, CPW, ROWMAJOR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_wg_kernel<RPT, CPW, ROWMAJOR>), dim3(plan.grid), dim3(WGT), plan.lds_bytes, stream, *solo);
#else
    (void)solo;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_wg_group_kernel<RPT, CPW, ROWMAJOR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_wg_group_kernel<RPT, CPW, ROWMAJOR>), dim3(8), dim3(WGT), plan.lds_bytes, stream, *group);
#endif
}
template <int RPT, int CPW> void wg_launch_rc(const RrluXcdPlan& plan, bool row_major, const RrluXcdArgs* solo, const RrluXcdGroupArgs* group, hipStream_t stream)
{
    if (row_major) wg_launch_one<RPT, CPW, true>(plan, solo, group, stream);
    else wg_launch_one<RPT, CPW, false>(plan, solo, group, stream);
}

// instantiated shapes: rows per lane x columns per wave (a plan rounds the columns up to the next one)
#ifndef T4A_WG_GROUP_TU
constexpr int kWgCpw1[] = {8, 16, 32, 64};
constexpr int kWgCpw2[] = {8, 16, 32, 48};
#endif

void wg_dispatch(const RrluXcdPlan& plan, bool row_major, const RrluXcdArgs* solo, const RrluXcdGroupArgs* group, hipStream_t stream)
{
    const int key = plan.RPT * 1000 + plan.CPT;
    switch (key) {
    case 1008: wg_launch_rc<1, 8>(plan, row_major, solo, group, stream); break;
    case 1016: wg_launch_rc<1, 16>(plan, row_major, solo, group, stream); break;
    case 1032: wg_launch_rc<1, 32>(plan, row_major, solo, group, stream); break;
    case 1064: wg_launch_rc<1, 64>(plan, row_major, solo, group, stream); break;
    case 2008: wg_launch_rc<2, 8>(plan, row_major, solo, group, stream); break;
    case 2016: wg_launch_rc<2, 16>(plan, row_major, solo, group, stream); break;
    case 2032: wg_launch_rc<2, 32>(plan, row_major, solo, group, stream); break;
    default: wg_launch_rc<2, 48>(plan, row_major, solo, group, stream); break;
    }
}

#ifndef T4A_WG_GROUP_TU
size_t wg_lds_bytes(int rpt, int cpw)
{
    const size_t MP = 64 * (size_t)rpt, NP = (size_t)WGW * cpw;
    return 2 * WGW * 16 + 2 * WGW * MP * 8 + 64 + MP * 8 + WGW * (2 * MP + 2 * NP) * 2;
}
static_assert(WgLds<2, 48>::bytes == 2 * WGW * 16 + 2 * WGW * 128 * 8 + 64 + 128 * 8 + WGW * (2 * 128 + 2 * 384) * 2, "plan.lds_bytes must cover the layout");
#endif

} // namespace

#ifndef T4A_WG_GROUP_TU
// One-workgroup plan for an M x N matrix (upper bounds in a bond chain), or false when it does not fit.  spec_blocks: workgroups
// beside the factorising one that evaluate the next bond's candidate matrix (bond chain, solo launch).
bool rrlu_wg_make_plan(int M, int N, RrluXcdPlan* out, int spec_blocks)
{
    static const bool off = std::getenv("T4A_NO_WG") != nullptr;
    if (off || M < 1 || N < 1 || M > 128) return false;
    const int rpt = (M + 63) / 64;
    const int need = (N + WGW - 1) / WGW;
    const int* list = rpt == 1 ? kWgCpw1 : kWgCpw2;
    const int nlist = 4;
    int cpw = -1;
    for (int i = 0; i < nlist; ++i)
        if (need <= list[i]) {
            cpw = list[i];
            break;
        }
    if (cpw < 0) return false;
    // beyond 16 values per lane (64 x 128, 128 x 64) the update (2 readlanes + 3 RPT vector instructions per owned column, all on ONE compute unit)
    // costs more than the single-XCD kernel's two L2 hand-offs (measured: tools/probe_wg.py)
    static const int max_values = diag_env("T4A_WG_MAXV") ? std::atoi(diag_env("T4A_WG_MAXV")) : 16;
    if (rpt * cpw > max_values) return false;
    RrluXcdPlan plan;
    plan.W = 1;
    plan.RPT = rpt;
    plan.CPT = cpw;
    plan.grid = 1 + (spec_blocks > 0 ? spec_blocks : 0);
    plan.lds_bytes = wg_lds_bytes(rpt, cpw);
    plan.wg = 1;
    *out = plan;
    return true;
}

void rrlu_wg_launch(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream) { wg_dispatch(plan, a.tie_row_major != 0, &a, nullptr, stream); }
#else
// (this half of the file is compiled as its own translation unit: kernels_rrlu_wg_group.hip)
void rrlu_wg_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool tie_row_major, hipStream_t stream)
{
    wg_dispatch(plan, tie_row_major, nullptr, &a, stream);
}
#endif

} // namespace t4a
