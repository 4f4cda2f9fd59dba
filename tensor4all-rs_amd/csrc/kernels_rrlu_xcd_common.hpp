// kernels_rrlu_xcd_common.hpp — helpers shared by the two generations of the single-XCD rrLU kernel (kernels_rrlu_xcd.hip,
// kernels_rrlu_xcd2.hip): wave reductions through DPP, tagged 16-byte granules, the bitwise-IEEE division through a shared
// refined reciprocal, the speculative candidate-matrix work of the pass-through workgroups.  Everything sits in an anonymous
// namespace: each translation unit gets its own copy (the kernels are templates instantiated per translation unit anyway).
#pragma once
#include "kernels.hpp"

#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <stdexcept>

namespace t4a {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned XNOPOS = 0xFFFFFFFFu;
#ifndef T4A_XCD_WAVES
#define T4A_XCD_WAVES 8
#endif
constexpr int XWAVES = T4A_XCD_WAVES; // agents (waves) per workgroup.  Measured: 4 (one wave per SIMD) runs every phase 1.5 - 2x slower — a single wave issues one f64 instruction per 8 cycles, two waves per SIMD reach the 4-cycle rate
constexpr int XT = 64 * XWAVES;       // threads per workgroup
constexpr int XCD_MAX_CPT = 4;       // (the key carries the column slot in two bits)
#ifndef T4A_XCD_MAXV
#define T4A_XCD_MAXV 64
#endif
constexpr int XCD_MAX_VALUES = XWAVES == 4 ? 80 : T4A_XCD_MAXV; // matrix entries per thread: beyond this the register file of a 512-thread workgroup spills
constexpr int BUF_SC1 = 16;  // aux bits of the raw buffer loads: sc1 (L1 bypass, served by the XCD's L2)
// The RE-loads of the polling loops sit behind a compiler barrier (xcd_poll_again): a raw buffer load is an ordinary memory read to
// the optimiser, and a loop that only re-reads one address until a tag matches is a loop-invariant load to it — hoisted, the loop
// spins on its first answer until the bounded poll gives up.  (The intrinsic's compiler-only volatile bit, aux bit 31, also keeps
// the loads in place but selects system-scope loads, sc0 sc1: the barrier keeps them the same sc1 loads as the first ones.)
__device__ __forceinline__ void xcd_poll_again() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
// maxNum(a, |b|): a NaN operand drops out (matrixlu.rs:506: a NaN score never replaces the incumbent)
__device__ __forceinline__ double vmax_abs(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL>((int)(b & 0xFFFFFFFFll));
    const int hi = dpp_i32<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// DPP move whose lanes without a source read 0 (bound_ctrl): no "old" value, so no copy in front of every move.  Only good
// where those lanes do not matter.
template <int CTRL> __device__ __forceinline__ double dppz_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_f64(double v) // maxNum over the 64 lanes (uniform result, in scalar registers)
{
    v = vmax(v, dppz_f64<0xB1>(v));  // quad_perm [1,0,3,2]
    v = vmax(v, dppz_f64<0x4E>(v));  // quad_perm [2,3,0,1]
    v = vmax(v, dppz_f64<0x141>(v)); // row_half_mirror
    v = vmax(v, dppz_f64<0x140>(v)); // row_mirror: every lane of a row holds the row maximum
    // the two broadcasts have no source for row 0 (and row 1): those lanes see 0 and go wrong, but lane 63, the only one read,
    // is fed by lanes 15, 31 and 47, which are right: 15 after the row stages, 31 and 47 after row_bcast15
    v = vmax(v, dppz_f64<0x142>(v)); // row_bcast15: rows 1..3 see lane 15 of the previous row
    v = vmax(v, dppz_f64<0x143>(v)); // row_bcast31: rows 2, 3 see lane 31 -> lane 63 holds the wave maximum
    return readlane_f64(v, 63);
}
// maximum of a signed 32-bit value over the 64 lanes, in lane 63 (DPP row reductions folded into v_max_i32: lanes without a
// source keep INT_MIN, the identity).  Non-negative doubles order like their high words first: the reductions of the step loop
// run on the high word and fall back to the 64-bit comparison only when it does not single out one lane.
template <int CTRL> __device__ __forceinline__ int dpp_max_i32(int v)
{
    const int o = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xF, 0xF, false);
    return o > v ? o : v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = dpp_max_i32<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_max_i32<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_max_i32<0x141>(v); // row_half_mirror
    v = dpp_max_i32<0x140>(v); // row_mirror: every lane of a row holds the row maximum
    v = dpp_max_i32<0x142>(v); // row_bcast15
    v = dpp_max_i32<0x143>(v); // row_bcast31 -> lane 63 holds the wave maximum
    return __builtin_amdgcn_readlane(v, 63);
}
// biased exponent of a non-negative double's high word within [600, 1500]: the square is a normal number with room to spare
__device__ __forceinline__ bool hi_mid(int hi) { return (unsigned)((hi >> 20) - 600) <= 900u; }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    unsigned o;
    o = (unsigned)dpp_i32<0xB1>((int)v);
    v = o < v ? o : v;
    o = (unsigned)dpp_i32<0x4E>((int)v);
    v = o < v ? o : v;
    o = (unsigned)dpp_i32<0x141>((int)v);
    v = o < v ? o : v;
    o = (unsigned)dpp_i32<0x140>((int)v);
    v = o < v ? o : v;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ unsigned hi32(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32); }
__device__ __forceinline__ unsigned lo32(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ double mk_f64(unsigned lo, unsigned hi)
{
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Reciprocal of p refined exactly like the f64 division expansion of the compiler (v_rcp_f64 + two Newton steps); with it
// x / p == fma(fma(-p, x * r, x), r, x * r) bitwise whenever the hardware sequence would not rescale its operands
// (v_div_scale is the identity while both exponents are far from the ends of the range), see xcd_div below.
__device__ __forceinline__ double refined_rcp(double p)
{
    const double r0 = __builtin_amdgcn_rcp(p);
    const double e0 = __builtin_fma(-p, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-p, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
// biased exponent within [723, 1323] (|v| in 2^-300 .. 2^300): no rescaling, no special case in the division sequence
__device__ __forceinline__ bool exp_mid(double v) { return (((hi32(v) >> 20) & 0x7FFu) - 723u) <= 600u; }
__device__ __forceinline__ double xcd_div(double x, double p, double rp, bool p_mid)
{
    if (p_mid && (exp_mid(x) || x == 0.0)) {
        const double q0 = x * rp;
        const double res = __builtin_fma(-p, q0, x);
        return (x == 0.0) ? q0 : __builtin_fma(res, rp, q0);
    }
    return x / p; // denormals, huge ratios, non-finite values: the full IEEE sequence
}

__host__ __device__ constexpr int xcd_lstr(int rpt) // doubles per lane in the l buffer: even, and odd in units of 16 bytes
{
    return ((rpt + 1) / 2) % 2 == 1 ? ((rpt + 1) / 2) * 2 : ((rpt + 1) / 2) * 2 + 2;
}

// per-phase cycle stamps of rank 0 / thread 0 (T4A_RRLU_STAMPS=1), kept in LDS.  Diagnostic builds only
// (T4A_EXTRA_FLAGS=-DT4A_XCD_STAMPS): the sixteen uniform branches and the live timestamp cost scalar registers and
// instructions in every step of the production kernel otherwise.
#ifdef T4A_XCD_STAMPS
constexpr bool kXcdStamps = true;
#else
constexpr bool kXcdStamps = false;
#endif
#define XSTAMP(slot)                                                      \
    do {                                                                  \
        if (stamp_on) {                                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            lds_stamps[slot] += now_ - stamp_last;                        \
            stamp_last = now_;                                            \
        }                                                                 \
    } while (0)

__device__ __forceinline__ double uniform_f64(double v) // a wave-uniform value moved into scalar registers
{
    return mk_f64((unsigned)__builtin_amdgcn_readfirstlane((int)lo32(v)), (unsigned)__builtin_amdgcn_readfirstlane((int)hi32(v)));
}

// LDS layout of one workgroup: every offset is a compile-time constant (tables sized for the largest matrix of the plan
// family), so no address of it lives in a scalar register across the step loop
template <int RPT> struct XcdLds {
    static constexpr int LSTR = xcd_lstr(RPT);
    static constexpr int o_l = 0;                        // double [64][LSTR]: l of row lane + 64 r at lane * LSTR + r
    static constexpr int o_wd = o_l + 64 * LSTR * 8;     // (16 bytes, unused: the key relay slot of the first versions)
    static constexpr int o_wi = o_wd + 16;               // int [16]: [0,1] winner value bits [2] meta [3] agent | rk << 8 | ck << 18 | stop << 28 [4] abort [5] next diagonal element: row | column << 10 [6] rank
    static constexpr int o_pp = o_wi + 64;               // u64 [2]: iresult / h_block pointers for the give-up paths
    static constexpr int o_st = o_pp + 16;               // u64 [16] phase stamps (diagnostic builds)
    static constexpr int o_pv = o_st + 128;              // double [1024] pivot values of this launch
    static constexpr int o_pr = o_pv + 1024 * 8;         // u16 [1024] position -> row index
    static constexpr int o_rp = o_pr + 1024 * 2;         // u16 [1024] row index -> position
    static constexpr int o_pc = o_rp + 1024 * 2;         // u16 [1024] position -> column index
    static constexpr int bytes = o_pc + 1024 * 2;
};
constexpr size_t xcd_lds_total(int rpt) { return (size_t)64 * xcd_lstr(rpt) * 8 + 16 + 64 + 16 + 128 + 1024 * 8 + 4 * 1024 * 2; } // (the fourth table, column index -> position, belongs to the second-generation kernel)
// second generation, round 5: plans beyond 1024 rows (RPT > 16) or 1024 columns (agents on several XCDs) carry tables of 2048 entries
__host__ __device__ constexpr int xcd2_tbl(int rpt, int kx = 1) { return (rpt > 16 || kx > 1) ? 2048 : 1024; }
constexpr size_t xcd2_lds_total(int rpt, int kx = 1) { return (size_t)64 * xcd_lstr(rpt) * 8 + 16 + 64 + 16 + 128 + (size_t)xcd2_tbl(rpt, kx) * 8 + 4 * (size_t)xcd2_tbl(rpt, kx) * 2; }

// a value the optimiser must treat as unknown: keeps loop-invariant masks / addresses of RARE paths from being hoisted out of
// the step loop into scalar registers (the kernel is bound by its scalar register file: every hoisted lane mask is a pair)
__device__ __forceinline__ int opaque_s(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ int opaque_v(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}


// key meta word: bits 0..19 position key (10 + 10 bits, tie order), 20..29 row index of the candidate, 30..31 column slot
// of the publishing agent.  An agent without a candidate publishes value 0 with the largest position key.
constexpr unsigned XKEY_NONE = 0xFFFFFu;

// The rank-1 update must run IN PLACE: written as plain C++ the register allocator gives every updated entry a new register
// and moves the whole block back at the loop's back edge (dozens of v_mov_b64 per step).  Tied operands leave it no choice.
__device__ __forceinline__ void sub_in_place(double& a, double prod) // a = a - prod (one rounding, like the reference's un-fused update)
{
    asm("v_add_f64 %0, %0, -%1" : "+v"(a) : "v"(prod));
}
template <int N> using xvec = double __attribute__((ext_vector_type(N)));

// The row slots of one owned column (second generation): one register tuple up to 16 slots.  A tuple of 24 doubles (48 registers) is
// not an indexable register class — the compiler copies such a block to scratch memory and loads the element back, in every step —
// so larger columns are TWO tuples of RPT / 2 slots; a run-time slot index picks the tuple with a (wave-uniform) select.
template <int RPT> struct XSlab {
    static constexpr int NV = RPT > 16 ? 2 : 1, HV = RPT / NV;
    static_assert(RPT == NV * HV, "row slots split evenly over the tuples");
    xvec<HV> h[NV];
    __device__ __forceinline__ double get(int r) const { return h[r / HV][r % HV]; } // (r is a constant after unrolling)
    __device__ __forceinline__ void set(int r, double v) { h[r / HV][r % HV] = v; }
    __device__ __forceinline__ double dyn(int idx) const // wave-uniform run-time slot
    {
        if constexpr (NV == 1) {
            return h[0][idx];
        } else {
            const double lo = h[0][idx < HV ? idx : 0], hi = h[1][idx >= HV ? idx - HV : 0];
            return idx < HV ? lo : hi;
        }
    }
};
template <int Q, int RPT, int AUX = 0>
__device__ __forceinline__ void xcd_publish_column(const XSlab<RPT>& col, __amdgpu_buffer_rsrc_t mail, int myslot, unsigned tag, int rstride = 1024)
{
    asm volatile("; column slot %0" ::"n"(Q));
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const double av = col.get(r);
        u32x4 gv;
        gv.x = lo32(av);
        gv.y = hi32(av);
        gv.z = 0u;
        gv.w = tag ^ gv.x ^ gv.y;
        __builtin_amdgcn_raw_buffer_store_b128(gv, mail, myslot + r * rstride, 0, AUX);
    }
}

// Publication of one owned column: 16-byte granules {lo, hi, 0, tag ^ lo ^ hi}, one per slot row (rows beyond M hold zeros:
// every slot row is written, no row mask).  One instance per column slot (the marker keeps the instances apart: merged, the
// compiler would first copy the selected column into a common block of registers).
template <int Q, int RPT, int AUX = 0> // AUX: cache bits of the stores (BUF_SC1: write-through, for readers on other XCDs)
__device__ __forceinline__ void xcd_publish_column(const xvec<RPT>& col, __amdgpu_buffer_rsrc_t mail, int myslot, unsigned tag, int rstride = 1024)
{
    asm volatile("; column slot %0" ::"n"(Q));
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const double av = col[r];
        u32x4 gv;
        gv.x = lo32(av);
        gv.y = hi32(av);
        gv.z = 0u;
        gv.w = tag ^ gv.x ^ gv.y;
        __builtin_amdgcn_raw_buffer_store_b128(gv, mail, myslot + r * rstride, 0, AUX);
    }
}

// Bond chain: what the pass-through workgroups do instead of returning at once (XcdSpecArgs in kernels.hpp).  Tiles of XT
// candidates x SPEC_TJ independent entries are handed out through a global counter, so it does not matter how many workgroups
// there are or where they run.  M = rows of this launch's matrix (= entries of the dependent list).
constexpr int SPEC_TJ = 16;
__device__ __forceinline__ const char* kernarg_base() // the kernel-argument segment (the by-value RrluXcdArgs sits at its start)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (const char*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}
// (not inlined, and handed a pointer into the kernel-argument segment rather than a reference to the by-value argument: the
// step loop of the kernel is bound by its scalar registers, nothing of this path may leak into its register allocation)
__device__ __forceinline__ void xcd_spec_work(const XcdSpecArgs* spp, int M, int* lds_tile)
{
    const XcdSpecArgs& sp = *spp;
    const int tid = threadIdx.x;
    const int K = sp.fn.n_acc;
    const int ne = sp.ext_cnt ? *sp.ext_cnt : 0;
    const int ni = *sp.ind_cnt;
    const int nkron = M * sp.d;
    const int lda = nkron + ne;
    if (ni <= 0 || lda <= 0) return;
    const int tiles_c = (lda + XT - 1) / XT, tiles_j = (ni + SPEC_TJ - 1) / SPEC_TJ;
    const int n_tiles = tiles_c * tiles_j;
    for (;;) {
        __syncthreads();
        if (tid == 0) *lds_tile = (int)atomicAdd(sp.tile_counter, 1u);
        __syncthreads();
        const int t = *lds_tile;
        if (t >= n_tiles) break;
        const int cand = (t % tiles_c) * XT + tid;
        const int j0 = (t / tiles_c) * SPEC_TJ;
        if (cand >= lda) continue;
        uint64_t racc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
        if (cand < nkron) {
            const int par = cand / sp.d, dg = cand % sp.d;
            for (int k = 0; k < K; ++k) racc[k] = sp.dep_acc[(size_t)par * K + k] + sp.w_site[(size_t)k * sp.total + dg];
        } else {
            const int e = cand - nkron;
            for (int k = 0; k < K; ++k) racc[k] = sp.ext_acc[(size_t)e * K + k];
        }
        const int j1 = j0 + SPEC_TJ < ni ? j0 + SPEC_TJ : ni;
        for (int j = j0; j < j1; ++j) {
            uint64_t acc[T4A_FN_MAX_ACC] = {0, 0, 0, 0};
            for (int k = 0; k < K; ++k) acc[k] = racc[k] + sp.ind_acc[(size_t)j * K + k];
            sp.out[(size_t)j * lda + cand] = t4a_fn_value(sp.fn.fid, acc, sp.fn.params);
        }
    }
}

} // namespace

} // namespace t4a
