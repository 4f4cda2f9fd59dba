// kernels_rrlu_reg.hip — K2 fast path: register-resident full-pivot rank-revealing LU for gfx950.
//
// Same contract as kernels_rrlu.hip (bit-identical to rrlu_mut, tensor4all-core/src/matrixlu.rs:735-819),
// restricted to the LEFT-orthogonal elimination; the engine runs a right-orthogonal factorisation as the
// left-orthogonal one of A^T with `tie_row_major = 1` (x*y == y*x bitwise, so every value is identical; only
// the arg-max tie order — column-major of A == row-major of A^T — has to follow, matrixlu.rs:480-519).
//
// Design notes (every item below is a measured fix, see profiles/ and DESIGN.md):
//   * the slab lives in REGISTERS: thread (tr, tc) of a TR x TC thread grid owns rows tr + TR*r (r < RPT) of
//     columns w + W*(tc + TC*q) (q < CPT); the rank-1 update is pure VALU on RPT*CPT independent elements
//     (the LDS-resident kernel spent 2.5 us/step in dependent LDS round trips);
//   * arg-max = v_max_f64 chain (NaN scores drop out of maxNum for free) + one equality sweep for the
//     smallest position; wave reductions use DPP row all-reduce + v_readlane (ds_bpermute chains cost
//     0.9 us/step);
//   * the permutation replica keeps only position->index tables in LDS; index->position lives in the
//     owning threads' registers; stop tests are evaluated redundantly by every thread;
//   * inter-workgroup exchange, two hops per pivot step, no drain barrier, form R2 of
//     cdna_hip_programming.md §6 Guideline 16 (the data is the flag: every granule is one aligned 8-byte
//     atomic sc1 store / sc1 load carrying a launch-salted 32-bit tag, consumers re-read until every tag
//     matches; spins are bounded):
//       hop 1 (arg-max all-gather): each workgroup stores ONE 16-byte key {value, position} into a shared table;
//         one wave per workgroup sweeps the table (sc1 loads of unchanged lines are L2 hits, ~270 cycles —
//         tools/ld_bench.hip).  Measured alternatives (tools/xchg_bench.hip, W = 86, cycles per round):
//         shared table 3 860, per-workgroup inboxes 5 690; speculatively publishing every workgroup's
//         candidate column instead of hop 2: 5.4 us/step.
//       hop 2 (pivot column broadcast): every workgroup has a slot for its candidate column; a workgroup whose
//         candidate score is >= spec_frac * (previous pivot)^2 — ~55 of 230 at the mid bond — stores it together with
//         its key, so the winner's column is normally already in flight while the keys are gathered and the readers issue
//         their column loads straight after the gather; a winner that did not speculate publishes afterwards (the tagged
//         granules make both cases one reader loop).
//   * no stream operation besides the launch: built-in functors are evaluated straight into the slab (p.fused),
//     results are mirrored into pinned host memory by workgroup 0, which also resets the device header and clears
//     the key table of the next launch (two alternating tables).
// Barriers per pivot step: 3.
// Thresholded speculative publication of the pivot column is compiled in (measured on MI355X: 42.1 -> 40.8 ms of rrLU per
// sweep at d = 30, chi = 256; selecting the mode through RrluRegArgs::spec at run time costs the default path 2 ms of
// code-generation noise, so it is a build-time choice).  Remove the define to get the run-time switch back.
#define T4A_RRLU_SPEC2 1
#include "kernels.hpp"

#include <mutex>

#include <algorithm>
#include <cstdlib>

namespace t4a {

namespace {

constexpr unsigned NOPOS = 0xFFFFFFFFu;

__device__ __forceinline__ void st_u64_sc1(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_u64_sc1(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte write-through store / agent-scope loads (inline asm: hipcc neither counts nor pads these, so the
// loads carry their own s_waitcnt and the store its s_nop — cdna_hip_programming.md §5.7).
__device__ __forceinline__ void st_b128_sc1(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// one pivot-column row = two tagged 8-byte granules {value lo | tag, value hi | tag}, stored as ONE 16-byte write-through
// store (a single fabric write; measured against two 8-byte stores with T4A_RRLU_COL_ST8 builds)
__device__ __forceinline__ void st_col_row(unsigned long long* dst, unsigned long long tagbits, unsigned long long vb)
{
#ifdef T4A_RRLU_COL_ST8
    st_u64_sc1(dst, tagbits | (vb & 0xFFFFFFFFull));
    st_u64_sc1(dst + 1, tagbits | (vb >> 32));
#else
    u32x4 v;
    v.x = (unsigned)vb;
    v.y = (unsigned)(tagbits >> 32);
    v.z = (unsigned)(vb >> 32);
    v.w = (unsigned)(tagbits >> 32);
    st_b128_sc1(dst, v);
#endif
}
template <int N> struct Load16;
template <> struct Load16<1> {
    static __device__ __forceinline__ void run(const void* const* p, u32x4* o)
    {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o[0]) : "v"(p[0]) : "memory");
    }
};
template <> struct Load16<2> {
    static __device__ __forceinline__ void run(const void* const* p, u32x4* o)
    {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1])
                     : "v"(p[0]), "v"(p[1])
                     : "memory");
    }
};
template <> struct Load16<3> {
    static __device__ __forceinline__ void run(const void* const* p, u32x4* o)
    {
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                     "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2])
                     : "memory");
    }
};
template <> struct Load16<4> {
    static __device__ __forceinline__ void run(const void* const* p, u32x4* o)
    {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
                     : "memory");
    }
};

// maxNum without the sNaN-quieting canonicalisation hipcc adds to fmax (both operands are arithmetic results)
__device__ __forceinline__ double vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- DPP helpers: all-reduce inside a row of 16 lanes, then combine the 4 rows through v_readlane ----
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
__device__ __forceinline__ double wave_max_f64(double v) // maxNum over the 64 lanes (uniform result)
{
    v = fmax(v, dpp_f64<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmax(v, dpp_f64<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmax(v, dpp_f64<0x141>(v)); // row_half_mirror
    v = fmax(v, dpp_f64<0x140>(v)); // row_mirror
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    return fmax(fmax(a, b), fmax(c, d));
}
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

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) / 16 * 16; }

struct RegSmem {
    double* urow;            // TC*CPT pivot-row entries of the owned columns
    double* lcol;            // M (single-workgroup mode only): raw pivot column
    double* red_sc;          // 16
    unsigned* red_pos;       // 16
    double* red_val;         // 16
    double* win_d;           // [0] value
    int* win_i;              // [0] winner wg  [1] position key  [2] abort flag
    unsigned short* posrow;  // M
    unsigned short* poscol;  // N
};

__host__ __device__ inline size_t reg_smem_layout(int M, int N, int cols_per_wg, bool single, RegSmem* s, char* base)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align16(off + bytes);
        return o;
    };
    const size_t o_urow = take((size_t)cols_per_wg * 8);
    const size_t o_lcol = take(single ? (size_t)M * 8 : 8);
    const size_t o_rsc = take(16 * 8);
    const size_t o_rpos = take(16 * 4);
    const size_t o_rval = take(16 * 8);
    const size_t o_wd = take(2 * 8);
    const size_t o_wi = take(4 * 4);
    const size_t o_pr = take((size_t)M * 2);
    const size_t o_pc = take((size_t)N * 2);
    if (s) {
        s->urow = (double*)(base + o_urow);
        s->lcol = (double*)(base + o_lcol);
        s->red_sc = (double*)(base + o_rsc);
        s->red_pos = (unsigned*)(base + o_rpos);
        s->red_val = (double*)(base + o_rval);
        s->win_d = (double*)(base + o_wd);
        s->win_i = (int*)(base + o_wi);
        s->posrow = (unsigned short*)(base + o_pr);
        s->poscol = (unsigned short*)(base + o_pc);
    }
    return off;
}

// thresholded speculative column publication compiled in (T4A_RRLU_SPEC2) or selected at run time (RrluRegArgs::spec)
#ifdef T4A_RRLU_SPEC2
#define T4A_SPEC(p) 2
#else
#define T4A_SPEC(p) ((p).spec)
#endif
#define T4A_RSTAMP(slot)                                                  \
    do {                                                                  \
        if (stamp_on) {                                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            stamp_acc[slot] += now_ - stamp_last;                         \
            stamp_last = now_;                                            \
        }                                                                 \
    } while (0)

// UNI: every wave lies inside one column group (TR % 64 == 0), so the column state is wave-uniform and the
// per-column tests become scalar branches.
// ROWMAJOR: ties of the arg-max break row-major (the right-orthogonal factorisation runs on the transposed matrix).
template <int RPT, int CPT, bool SINGLE, bool UNI, bool ROWMAJOR>
__global__ void __attribute__((amdgpu_flat_work_group_size(64, (RPT * CPT > 24) ? 256 : 512)))
rrlu_reg_kernel(RrluRegArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    RegSmem s;
    reg_smem_layout(p.M, p.N, p.TC * CPT, SINGLE, &s, smem_raw);
    // bond chain (single-workgroup launches): the real dimensions come from device memory; the launch (thread grid, LDS
    // layout) was planned for the upper bounds p.M x p.N
    int M = p.M, N = p.N, max_steps = p.max_steps;
    if (SINGLE && p.dims) {
        const int d0 = p.dims[0], d1 = p.dims[1];
        M = p.dims_swap ? d1 : d0;
        N = p.dims_swap ? d0 : d1;
        if (M > p.M || N > p.N) M = N = 0;
        const int mn = M < N ? M : N;
        max_steps = max_steps < mn ? max_steps : mn;
        if (mn <= 0) return; // poisoned bond
    }
    const unsigned long long ts_begin = (SINGLE && p.ts_u64 > 0) ? wall_clock64() : 0ull;

    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = T >> 6;
    const int w = SINGLE ? 0 : (int)blockIdx.x;
    const int tr = tid % p.TR;
    const int tc = tid / p.TR;
    constexpr bool rowmajor = ROWMAJOR; // tie order of the arg-max: a run-time flag here costs the loop 1.5 %

    // ---- my rows / columns ----
    int irow[RPT], rpos[RPT];
    int ccol[CPT], cpos[CPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = tr + p.TR * r;
        irow[r] = i < M ? i : -1;
        rpos[r] = i < M ? i : -1;
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int c = w + p.W * (tc + p.TC * q);
        ccol[q] = c < N ? c : -1;
        cpos[q] = c < N ? c : -1;
    }
    double a[CPT][RPT];
    double local_absmax = 0.0;
    if (RPT * CPT <= RRLU_FUSED_MAX_VALUES && p.fused) { // (larger slabs spill around the inlined functor)
        // the candidate matrix is built straight into the registers (tensorci2.rs:1859-1893): no Π in memory at all
        const int K = p.fn.n_acc;
        uint64_t racc[RPT][T4A_FN_MAX_ACC];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int k = 0; k < T4A_FN_MAX_ACC; ++k)
                racc[r][k] = (irow[r] >= 0 && k < K) ? p.rowacc[(size_t)irow[r] * K + k] : 0ull;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            uint64_t cacc[T4A_FN_MAX_ACC];
#pragma unroll
            for (int k = 0; k < T4A_FN_MAX_ACC; ++k)
                cacc[k] = (ccol[q] >= 0 && k < K) ? p.colacc[(size_t)ccol[q] * K + k] : 0ull;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                double v = 0.0;
                if (ccol[q] >= 0 && irow[r] >= 0) {
                    uint64_t acc[T4A_FN_MAX_ACC];
#pragma unroll
                    for (int k = 0; k < T4A_FN_MAX_ACC; ++k) acc[k] = racc[r][k] + cacc[k];
                    v = t4a_fn_value(p.fn.fid, acc, p.fn.params);
                    const double av = sqrt(v * v);
                    if (av > local_absmax) local_absmax = av;
                }
                a[q][r] = v;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < CPT; ++q)
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                double v = 0.0;
                if (ccol[q] >= 0 && irow[r] >= 0) {
                    v = (SINGLE && p.rowmap) ? p.A[(size_t)ccol[q] * p.dims[3] + p.rowmap[irow[r]]] : p.A[(size_t)ccol[q] * M + irow[r]];
                    const double av = sqrt(v * v);
                    if (av > local_absmax) local_absmax = av;
                }
                a[q][r] = v;
            }
    }
    for (int i = tid; i < M; i += T) s.posrow[i] = (unsigned short)i;
    for (int j = tid; j < N; j += T) s.poscol[j] = (unsigned short)j;
    if (tid == 0) s.win_i[2] = 0;
    {
        const double wm = wave_max_f64(local_absmax);
        if (lane == 0 && wm > 0.0)
            atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(wm));
    }
    __syncthreads();

#ifdef T4A_RRLU_TRACE
    if (p.trace && tid == 0 && !SINGLE) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.trace[w] = xcc & 0xF;
    }
#endif
    const bool stamp_on = (p.stamps != nullptr) && w == 0 && tid == 0;
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;

    int npiv = 0;
    double max_error = 0.0;
    double error = __builtin_nan("");
    bool timed_out = false;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;

    double l[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) l[r] = 0.0;
    double prev_sq = __builtin_huge_val(); // square of the previous pivot (thresholded speculation: nobody speculates first)
    const double spec_frac = p.spec_frac;

    for (int k = -1; k < max_steps; ++k) {
        // =====================================================================================
        // (C) rank-1 update of step k (k >= 0) fused with the candidate search of step k+1
        // =====================================================================================
        double m = -1.0;
        {
            double u[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) u[q] = (k >= 0) ? s.urow[tc + p.TC * q] : 0.0; // issued back to back
            int cps[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) cps[q] = UNI ? __builtin_amdgcn_readfirstlane(cpos[q]) : cpos[q];
#ifdef T4A_RRLU_FLAT_PASS // measured: the pass itself gets 15 % shorter, the step does not (the exchange dominates)
            {
                // branch-free form: every element is updated speculatively and selected by its activity masks
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const bool ract = rpos[r] > k;
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const bool act = ract && (cps[q] > k);
                        double t = a[q][r];
                        if (k >= 0) {
                            const double prod = l[r] * u[q]; // update_trailing_submatrix (matrixlu.rs:593-612)
                            const double upd = t - prod;
                            const bool piv = ract && (cps[q] == k); // scale_column_tail: owners store l_i
                            t = act ? upd : (piv ? l[r] : t);
                            a[q][r] = t;
                        }
                        const double sc = vmax(m, t * t); // maxNum drops NaN scores (matrixlu.rs:506)
                        m = act ? sc : m;
                    }
                }
            }
#else
            {
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    if (rpos[r] > k) { // EXEC mask: rows already pivoted keep their U entries untouched
#pragma unroll
                        for (int q = 0; q < CPT; ++q) {
                            if (cps[q] > k) {
                                if (k >= 0) {
                                    const double prod = l[r] * u[q]; // update_trailing_submatrix (matrixlu.rs:593-612)
                                    a[q][r] = a[q][r] - prod;
                                }
                                m = vmax(m, a[q][r] * a[q][r]); // maxNum drops NaN scores (matrixlu.rs:506)
                            } else if (k >= 0 && cps[q] == k) {
                                a[q][r] = l[r]; // scale_column_tail (matrixlu.rs:562-577): owners store l_i
                            }
                        }
                    }
                }
            }
#endif
        }
        npiv = k + 1;
        if (k + 1 >= max_steps) break; // the reference stops before another arg-max (matrixlu.rs:747)
        const int kn = k + 1;
        const unsigned diagkey = ((unsigned)kn << 16) | (unsigned)kn;
        // a NaN sitting on the next diagonal element wins outright (it is the reference's initial incumbent)
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (rpos[r] == kn) {
#pragma unroll
                for (int q = 0; q < CPT; ++q)
                    if (cpos[q] == kn && a[q][r] != a[q][r]) m = __builtin_huge_val();
            }
        T4A_RSTAMP(0);
#ifdef T4A_RRLU_TRACE
        if (p.trace && tid == 0 && !SINGLE) p.trace[p.W + ((size_t)kn * p.W + w) * 4 + 3] = wall_clock64(); // pass of step k done
#endif
        // ---- workgroup reduction: (max score, smallest position among the maxima, value there) ----
        const double wmax = wave_max_f64(m);
        unsigned mypos = NOPOS;
        double myval = 0.0;
        if (m == wmax) { // only lanes holding the wave maximum look for its position
            if (m == __builtin_huge_val()) {
                // +inf scores: a NaN on the diagonal (incumbent) or genuine infinities
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const bool act = (rpos[r] > k) && (cpos[q] > k);
                        const unsigned key = rowmajor ? (((unsigned)rpos[r] << 16) | (unsigned)cpos[q])
                                                      : (((unsigned)cpos[q] << 16) | (unsigned)rpos[r]);
                        const double sc = a[q][r] * a[q][r];
                        const bool hit = act && ((sc == m) || (key == diagkey && sc != sc));
                        if (hit && key < mypos) {
                            mypos = key;
                            myval = a[q][r];
                        }
                    }
            } else {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const bool act = (rpos[r] > k) && (cpos[q] > k);
                        const unsigned key = rowmajor ? (((unsigned)rpos[r] << 16) | (unsigned)cpos[q])
                                                      : (((unsigned)cpos[q] << 16) | (unsigned)rpos[r]);
                        const bool hit = act && (a[q][r] * a[q][r] == m) && (key < mypos);
                        mypos = hit ? key : mypos;
                        myval = hit ? a[q][r] : myval;
                    }
            }
        }
        const unsigned wpos = wave_min_u32(mypos);
        if (mypos == wpos && wpos != NOPOS) s.red_val[wave] = myval; // unique lane of the wave
        if (wpos == NOPOS && lane == 0) s.red_val[wave] = 0.0;
        if (lane == 0) {
            s.red_sc[wave] = wmax;
            s.red_pos[wave] = wpos;
        }
        T4A_RSTAMP(7);
        __syncthreads(); // (D)
        // all LDS reads of the cross-wave reduction are issued together (<= 16 waves)
        constexpr int NWMAX = (RPT * CPT > 24) ? 4 : 8; // = maximum workgroup size / 64
        double rsc[NWMAX], rvl[NWMAX];
        unsigned rps[NWMAX];
#pragma unroll
        for (int qv = 0; qv < NWMAX; ++qv) {
            if (qv < nwaves) {
                rsc[qv] = s.red_sc[qv];
                rps[qv] = s.red_pos[qv];
                rvl[qv] = s.red_val[qv];
            }
        }
        double bsc = rsc[0];
        unsigned bpos = rps[0];
        double bval = rvl[0];
#pragma unroll
        for (int qv = 1; qv < NWMAX; ++qv) {
            if (qv < nwaves) {
                const bool better = rsc[qv] > bsc || (rsc[qv] == bsc && rps[qv] < bpos);
                bsc = better ? rsc[qv] : bsc;
                bpos = better ? rps[qv] : bpos;
                bval = better ? rvl[qv] : bval;
            }
        }
        if (bsc < 0.0) {
            bpos = NOPOS; // no candidate in this workgroup
            bval = 0.0;
        }
        // which column group owns the candidate column
        const unsigned bcol = bpos == NOPOS ? NOPOS : (rowmajor ? (bpos & 0xFFFFu) : (bpos >> 16));
        int qstar = -1;
#pragma unroll
        for (int q = 0; q < CPT; ++q)
            if (cpos[q] >= 0 && (unsigned)cpos[q] == bcol) qstar = q;
        T4A_RSTAMP(1);

        // =====================================================================================
        // exchange: global winner of step kn
        // =====================================================================================
        double wval;
        unsigned wkey;
        int ww = 0;
        double colv[RPT];
        bool early_pub = false;
        if (SINGLE) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < CPT; ++q)
                    if (q == qstar) v = a[q][r];
                colv[r] = v;
            }
            if (qstar >= 0) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) s.lcol[irow[r]] = colv[r];
            }
            __syncthreads(); // (A')
            wval = bval;
            wkey = bpos;
        } else {
            const int par = kn & 1;
            const unsigned tag = p.salt * 65536u + ((unsigned)kn % 65535u + 1u);
            const unsigned long long tagbits = (unsigned long long)tag << 32;
            // hop 1 (arg-max all-gather): ONE 16-byte key per workgroup in a shared table keys[par][W][2]:
            //   g0 = tag16 | value bits 63..16,  g1 = tag16 | value bits 15..0 | position (32 bits).
            // The table is zeroed before every launch, tag16 = step + 1 (never 0).  Measured
            // (tools/xchg_bench.hip, W = 86): shared table 3 860 cycles/round, per-workgroup inboxes 5 690.
            const int poll_wave = nwaves > 1 ? 1 : 0; // not the storing wave: its loads would queue behind its stores
            const unsigned long long tag16 = (unsigned long long)((unsigned)kn % 65535u + 1u);
            if (wave == 0 && lane == 0) {
#ifdef T4A_RRLU_TRACE
                if (p.trace) p.trace[p.W + ((size_t)kn * p.W + w) * 4] = wall_clock64();
#endif
                const unsigned long long vb = (unsigned long long)__double_as_longlong(bval);
                unsigned long long* kd = p.keys + ((size_t)par * p.W + w) * 2;
                const unsigned long long k0 = (tag16 << 48) | (vb >> 16);
                const unsigned long long k1 = (tag16 << 48) | ((vb & 0xFFFFull) << 32) | (unsigned long long)bpos;
#ifdef T4A_RRLU_KEY_ST8 // the two 8-byte stores of the first versions (A/B builds only)
                st_u64_sc1(kd + 0, k0);
                st_u64_sc1(kd + 1, k1);
#else
                u32x4 kv; // one 16-byte write-through store: a single fabric write instead of two
                kv.x = (unsigned)k0;
                kv.y = (unsigned)(k0 >> 32);
                kv.z = (unsigned)k1;
                kv.w = (unsigned)(k1 >> 32);
                st_b128_sc1(kd, kv);
#endif
            }
            // everybody else (and the pusher afterwards) prepares the candidate column while the keys travel
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < CPT; ++q)
                    if (q == qstar) v = a[q][r];
                colv[r] = v;
            }
            // speculative mode: EVERY workgroup publishes its candidate column together with its key, so the winner's
            // column is already in flight while the keys are gathered (one hand-off per pivot step instead of two)
            // thresholded mode (spec == 2): only workgroups whose candidate is within a factor of the previous pivot publish
            // early (a handful per step instead of all W), the others catch up after the gather if they win
            early_pub = T4A_SPEC(p) == 1 || (T4A_SPEC(p) == 2 && bsc >= spec_frac * prev_sq);
            if (early_pub && qstar >= 0) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) {
                        const unsigned long long vb = (unsigned long long)__double_as_longlong(colv[r]);
                        unsigned long long* dst = p.cols + (((size_t)par * p.W + w) * (size_t)M + irow[r]) * 2;
                        st_col_row(dst, tagbits, vb);
                    }
            }
            T4A_RSTAMP(2);
            // one wave sweeps the shared key table until every tag matches
            if (wave == poll_wave) {
                for (int d = 0; d < p.poll_delay; ++d) __builtin_amdgcn_s_sleep(1); // skip sweeps that would certainly fail
                const unsigned long long* kb = p.keys + (size_t)par * p.W * 2;
                unsigned spins = 0;
                bool giveup = false;
                double csc = -1.0, cval = 0.0;
                unsigned cpk = NOPOS;
                int cw = -1;
                // all loads of one sweep are issued back to back (one memory round trip per sweep)
                constexpr int KPL = 4; // keys per lane: W <= 256 (rrlu_reg_make_plan clamps the plan)
                static_assert(KPL * 64 == 256, "engine.hip reserves 256 key slots per table");
                unsigned long long g[KPL][2];
                for (;;) {
                    bool ok = true;
#ifdef T4A_RRLU_KEY_LD8 // two 8-byte loads per key (A/B builds only); a run-time switch here costs the loop 1 %
                    const bool key_ld16 = false;
#else
                    const bool key_ld16 = true;
#endif
                    if (key_ld16) { // one 16-byte load per key instead of two 8-byte ones
                        const void* ptrs[KPL];
                        u32x4 got[KPL];
#pragma unroll
                        for (int j = 0; j < KPL; ++j) {
                            const int qw = lane + 64 * j;
                            ptrs[j] = kb + 2 * (size_t)(qw < p.W ? qw : 0);
                        }
                        Load16<KPL>::run(ptrs, got);
#pragma unroll
                        for (int j = 0; j < KPL; ++j) {
                            g[j][0] = ((unsigned long long)got[j].y << 32) | got[j].x;
                            g[j][1] = ((unsigned long long)got[j].w << 32) | got[j].z;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < KPL; ++j) {
                            const int qw = lane + 64 * j;
                            if (qw < p.W) {
                                g[j][0] = ld_u64_sc1(kb + 2 * (size_t)qw);
                                g[j][1] = ld_u64_sc1(kb + 2 * (size_t)qw + 1);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const int qw = lane + 64 * j;
                        if (qw < p.W) ok &= ((g[j][0] >> 48) == tag16) && ((g[j][1] >> 48) == tag16);
                    }
                    if (__all(ok)) break;
                    if (++spins > p.spin_limit) {
                        giveup = true;
                        break;
                    }
                }
                if (!giveup) {
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const int qw = lane + 64 * j;
                        if (qw < p.W) {
                            const unsigned pk = (unsigned)(g[j][1] & 0xFFFFFFFFull);
                            if (pk != NOPOS) {
                                const unsigned long long vb = (g[j][0] << 16) | ((g[j][1] >> 32) & 0xFFFFull);
                                const double v = __longlong_as_double((long long)vb);
                                double sc = v * v;
                                if (sc != sc) sc = (pk == diagkey) ? __builtin_huge_val() : -1.0;
                                if (sc > csc || (sc == csc && pk < cpk)) {
                                    csc = sc;
                                    cval = v;
                                    cpk = pk;
                                    cw = qw;
                                }
                            }
                        }
                    }
                }
                if (p.stamps != nullptr && w == 0 && lane == 0) p.stamps[5] += spins;
                if (giveup) {
                    if (lane == 0) {
                        s.win_i[2] = 1;
                        atomicExch(&p.iresult[1], 1);
                        if (p.h_block) ((volatile int*)p.h_block)[5] = 1;
                    }
                } else {
                    const double gmax = wave_max_f64(csc);
                    const unsigned gpos = wave_min_u32((csc == gmax) ? cpk : NOPOS);
                    if (csc == gmax && cpk == gpos && gpos != NOPOS) { // unique lane
                        s.win_d[0] = cval;
                        s.win_i[0] = cw;
                        s.win_i[1] = (int)cpk;
                    }
                }
            }
            __syncthreads(); // (B)
#ifdef T4A_RRLU_TRACE
            if (p.trace && tid == 0) p.trace[p.W + ((size_t)kn * p.W + w) * 4 + 1] = wall_clock64();
#endif
            if (s.win_i[2]) {
                timed_out = true;
                break;
            }
            wval = s.win_d[0];
            wkey = (unsigned)s.win_i[1];
            ww = s.win_i[0];
            // hop 2: only the winner's owning column group publishes the pivot column: one 16-byte store of two
            // tagged granules per row, replicated into `ncopy` copies so that at most W/ncopy readers share a line
            if (T4A_SPEC(p) == 2 && !early_pub && ww == w && qstar >= 0) { // the winner did not speculate: publish into its slot now
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) {
                        const unsigned long long vb = (unsigned long long)__double_as_longlong(colv[r]);
                        unsigned long long* dst = p.cols + (((size_t)par * p.W + w) * (size_t)M + irow[r]) * 2;
                        st_col_row(dst, tagbits, vb);
                    }
            }
            if (!T4A_SPEC(p) && ww == w && qstar >= 0) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) {
                        const unsigned long long vb = (unsigned long long)__double_as_longlong(colv[r]);
                        for (int c = 0; c < p.ncopy; ++c) {
                            unsigned long long* dst = p.cols + (((size_t)par * p.ncopy + c) * (size_t)M + irow[r]) * 2;
                            st_u64_sc1(dst, tagbits | (vb & 0xFFFFFFFFull));
                            st_u64_sc1(dst + 1, tagbits | (vb >> 32));
                        }
                    }
            }
        }
        const bool need_fetch = !SINGLE && !(ww == w && qstar >= 0);
        const unsigned long long* colsrc =
            SINGLE ? nullptr
                   : (T4A_SPEC(p) ? p.cols + ((size_t)((k + 1) & 1) * p.W + ww) * (size_t)M * 2
                             : p.cols + ((size_t)((k + 1) & 1) * p.ncopy + (w % p.ncopy)) * (size_t)M * 2);
        unsigned long long cg0[RPT], cg1[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            cg0[r] = 0ull;
            cg1[r] = 0ull;
        }
        if (need_fetch && T4A_SPEC(p) == 2) { // the winner most likely published with its key: the column is already there
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (irow[r] >= 0) {
                    cg0[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r]);
                    cg1[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r] + 1);
                }
        }
        T4A_RSTAMP(3);

        // ---- stop tests, every thread (matrixlu.rs:757-781) ----
        const double pivot_abs = sqrt(wval * wval);
        error = pivot_abs;
        if (kn > 0 && (pivot_abs < p.rel_tol * max_error || pivot_abs < p.abs_tol)) break;
        if (pivot_abs <= min_pivot_abs) break;
        max_error = fmax(max_error, pivot_abs);
        prev_sq = wval * wval;

        // ---- permutation bookkeeping: position -> index from the LDS tables, index -> position in registers ----
        const int prp = (int)(rowmajor ? (wkey >> 16) : (wkey & 0xFFFFu));
        const int pcp = (int)(rowmajor ? (wkey & 0xFFFFu) : (wkey >> 16));
        const int pr = s.posrow[prp];
        const int rk = s.posrow[kn];
        const int pc = s.poscol[pcp];
        const int ck = s.poscol[kn];

        // the owner of row pr publishes the pivot-row entries of its column group
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (irow[r] >= 0 && irow[r] == pr) {
#pragma unroll
                for (int q = 0; q < CPT; ++q) s.urow[tc + p.TC * q] = a[q][r];
            }
        }
        // index -> position updates (swap positions kn <-> prp, kn <-> pcp)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (irow[r] >= 0) {
                if (irow[r] == pr) rpos[r] = kn;
                else if (irow[r] == rk) rpos[r] = prp;
            }
        }
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            if (ccol[q] >= 0) {
                if (ccol[q] == pc) cpos[q] = kn;
                else if (ccol[q] == ck) cpos[q] = pcp;
            }
        }
        if (w == 0 && tid == 0) p.pivot_vals[kn] = wval;

        // issue the pivot-column loads (issuing them before the bookkeeping measured worse: more first-sweep misses; in the
        // thresholded speculative mode they were already issued straight after the gather)
        if (need_fetch && T4A_SPEC(p) != 2) {
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (irow[r] >= 0) {
                    cg0[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r]);
                    cg1[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r] + 1);
                }
        }

        // ---- pivot column -> l (scaled) ----
        if (SINGLE) {
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (irow[r] >= 0) l[r] = s.lcol[irow[r]] / wval;
        } else if (ww == w && qstar >= 0) {
            // the winner's owning column group already holds the column
#pragma unroll
            for (int r = 0; r < RPT; ++r) l[r] = colv[r] / wval;
        } else {
            const unsigned tag = p.salt * 65536u + ((unsigned)kn % 65535u + 1u);
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) ok &= ((unsigned)(cg0[r] >> 32) == tag) && ((unsigned)(cg1[r] >> 32) == tag);
                if (__all(ok)) break;
                if (++spins > p.spin_limit) {
                    atomicExch(&p.iresult[1], 1);
                    if (p.h_block) ((volatile int*)p.h_block)[5] = 1;
                    s.win_i[2] = 1; // observed by everybody after the next barrier
                    break;
                }
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (irow[r] >= 0) {
                        cg0[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r]);
                        cg1[r] = ld_u64_sc1(colsrc + 2 * (size_t)irow[r] + 1);
                    }
            }
            if (stamp_on) stamp_acc[6] += spins;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (irow[r] >= 0) {
                    const double raw = __longlong_as_double(
                        (long long)(((cg1[r] & 0xFFFFFFFFull) << 32) | (cg0[r] & 0xFFFFFFFFull)));
                    l[r] = raw / wval;
                }
        }
#ifdef T4A_RRLU_TRACE
        if (p.trace && tid == 0 && !SINGLE) p.trace[p.W + ((size_t)kn * p.W + w) * 4 + 2] = wall_clock64(); // column of step kn in registers
#endif
        __syncthreads(); // (C): urow visible, everybody has read the position tables
        if (!SINGLE && s.win_i[2]) {
            timed_out = true;
            break;
        }
        if (tid == 0) { // next read of these entries happens after barrier (D) of the next step
            s.posrow[kn] = (unsigned short)pr;
            s.posrow[prp] = (unsigned short)rk;
            s.poscol[kn] = (unsigned short)pc;
            s.poscol[pcp] = (unsigned short)ck;
        }
        T4A_RSTAMP(4);
    }

    // ---- results ----
    if (npiv >= (M < N ? M : N)) error = 0.0; // matrixlu.rs:811-813
    if (w == 0 && tid == 0) {
        p.iresult[0] = npiv;
        p.dresult[0] = error;
    }
    if (stamp_on)
        for (int qv = 0; qv < 8; ++qv)
            if (qv != 5) p.stamps[qv] = stamp_acc[qv];
    if (timed_out) return;
    __syncthreads();
    if (w == 0) {
        for (int i = tid; i < M; i += T) p.row_perm[i] = s.posrow[i];
        for (int j = tid; j < N; j += T) p.col_perm[j] = s.poscol[j];
    }
    int nan_seen = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (ccol[q] >= 0 && irow[r] >= 0) {
                const int cp = cpos[q], rp = rpos[r];
                const double v = a[q][r];
                const bool in_l = (cp < npiv) && (rp >= cp);
                const bool in_u = (rp < npiv) && (cp >= rp);
                if ((in_l || in_u) && v != v) nan_seen = 1;
                if (p.Aout) {
                    if (p.out_transposed)
                        p.Aout[(size_t)rp * N + cp] = v;
                    else
                        p.Aout[(size_t)cp * M + rp] = v;
                }
            }
        }
    if (nan_seen) {
        atomicExch(&p.iresult[2], 1);
        if (p.h_block) ((volatile int*)p.h_block)[6] = 1;
    }
    // host-visible mirror of the packed result block (everything but the two flag words, which their setters write)
    if (p.h_block && w == 0) {
        __syncthreads(); // this workgroup's writes to the device block (perms, pivot values, npiv, error) are visible
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.dresult);
        for (int e = tid; e < p.block_u64; e += T) {
            if (e == 2 || e == 3) continue;
            p.h_block[e] = (e == 1) ? ld_u64_sc1(src + 1) : src[e]; // [1] = max |a| bits: atomics of all workgroups
        }
        if (tid == 0) ((volatile int*)p.h_block)[4] = npiv;
        if (SINGLE && p.dims && tid == 0) p.iresult[3] = (int)p.dev_token; // bond chain: seen by the next preparation kernel
        if (SINGLE && p.done_token != 0u) { // everything this (only) workgroup sends to the host is out: completion token
            __threadfence_system();
            __syncthreads();
            if (tid == 0) ((volatile unsigned*)p.h_block)[7] = p.done_token;
        }
        // leave the device side clean for the next launch: the max|a| word is only ever raised by the atomics of the
        // load phase (all long done), and nobody touches the other key table during this launch
        __syncthreads();
        if (SINGLE && p.ts_u64 > 0 && tid == 0) { // bond chain: device-side start / end time of the factorisation (100 MHz)
            p.h_block[p.ts_u64] = ts_begin;
            p.h_block[p.ts_u64 + 1] = wall_clock64();
        }
        if (tid == 0) reinterpret_cast<unsigned long long*>(p.dresult)[1] = 0ull;
        for (int e = tid; e < p.keys_next_u64; e += T) p.keys_next[e] = 0ull;
    }
    // bond chain without per-launch host mirror: the device block is copied to the host once, behind the whole chain; it keeps
    // max |a| and gets the time stamps and the completion token here
    if (SINGLE && !p.h_block && p.dims) {
        __syncthreads();
        if (tid == 0) {
            if (p.ts_u64 > 0) {
                unsigned long long* const blk = reinterpret_cast<unsigned long long*>(p.dresult);
                blk[p.ts_u64] = ts_begin;
                blk[p.ts_u64 + 1] = wall_clock64();
            }
            __threadfence();
            p.iresult[3] = (int)p.dev_token;
        }
        for (int e = tid; e < p.keys_next_u64; e += T) p.keys_next[e] = 0ull;
    }
}

template <int RPT, int CPT, bool SINGLE, bool UNI, bool ROWMAJOR>
void launch_tie(const RrluRegPlan& plan, const RrluRegArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once; // (launches come from several host threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_reg_kernel<RPT, CPT, SINGLE, UNI, ROWMAJOR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_reg_kernel<RPT, CPT, SINGLE, UNI, ROWMAJOR>), dim3(SINGLE ? 1 : plan.W), dim3(plan.T), plan.lds_bytes,
                       stream, a);
}

template <int RPT, int CPT, bool SINGLE, bool UNI>
void launch_one(const RrluRegPlan& plan, const RrluRegArgs& a, hipStream_t stream)
{
    if (a.tie_row_major) launch_tie<RPT, CPT, SINGLE, UNI, true>(plan, a, stream);
    else launch_tie<RPT, CPT, SINGLE, UNI, false>(plan, a, stream);
}

template <int RPT, int CPT> void launch_rc(const RrluRegPlan& plan, const RrluRegArgs& a, hipStream_t stream)
{
    const bool uni = (plan.TR % 64) == 0;
    if (plan.W == 1) {
        if (uni) launch_one<RPT, CPT, true, true>(plan, a, stream);
        else launch_one<RPT, CPT, true, false>(plan, a, stream);
    } else {
        if (uni) launch_one<RPT, CPT, false, true>(plan, a, stream);
        else launch_one<RPT, CPT, false, false>(plan, a, stream);
    }
}

template <int RPT> void launch_r(const RrluRegPlan& plan, const RrluRegArgs& a, hipStream_t stream)
{
    switch (plan.CPT) {
    case 1: launch_rc<RPT, 1>(plan, a, stream); break;
    case 2: launch_rc<RPT, 2>(plan, a, stream); break;
    case 3: launch_rc<RPT, 3>(plan, a, stream); break;
    case 4: launch_rc<RPT, 4>(plan, a, stream); break;
    case 5: launch_rc<RPT, 5>(plan, a, stream); break;
    case 6: launch_rc<RPT, 6>(plan, a, stream); break;
    default: launch_rc<RPT, 8>(plan, a, stream); break;
    }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }
int norm_cpt(int c) { return c <= 1 ? 1 : (c <= 6 ? c : 8); }

} // namespace

bool rrlu_reg_make_plan(int M, int N, int num_cus, RrluRegPlan* out)
{
    RrluRegPlan plan;
    const char* ew = diag_env("T4A_RRLU_W");
    const char* et = diag_env("T4A_RRLU_T");
    const char* ec = diag_env("T4A_RRLU_CPT");
    // (the key-table poller reads 4 keys per lane and the engine reserves 256 slots: never plan more workgroups than that)
    const int maxw = std::min(num_cus > 16 ? num_cus - 8 : num_cus, 256);
    const long long elems = (long long)M * N;
    static const long long single_max = diag_env("T4A_RRLU_SINGLE_MAX") ? std::atoll(diag_env("T4A_RRLU_SINGLE_MAX")) : 64 * 64;
    bool single = elems <= single_max;
    if (ew) single = std::atoi(ew) == 1;
    bool found = false;
    if (single) {
        // one workgroup: TR x TC thread grid with <= 4 x 8 elements per thread; minimise the per-thread work,
        // then the thread count
        int best_cost = 1 << 30;
        for (int T = 64; T <= 512; T *= 2) {
            if (et && T != std::atoi(et)) continue;
            for (int TR = 16; TR <= T; TR *= 2) {
                const int TC = T / TR;
                const int RPT = (M + TR - 1) / TR;
                const int CPT = norm_cpt((N + TC - 1) / TC);
                if (RPT > 4 || (long long)TC * CPT < N) continue;
                if (T > ((RPT * CPT > 24) ? 256 : 512)) continue;
                const int cost = RPT * CPT * 64 + T / 64;
                if (cost < best_cost) {
                    best_cost = cost;
                    plan.W = 1;
                    plan.T = T;
                    plan.TR = TR;
                    plan.TC = TC;
                    plan.RPT = RPT;
                    plan.CPT = CPT;
                    found = true;
                }
            }
        }
    }
    if (!found) {
        int RPT = (M + 511) / 512; // two waves per SIMD hide the f64 issue latency (measured: T=512 beats 256)
        if (RPT > 4) RPT = 4;
        int TR = round_up((M + RPT - 1) / RPT, 64);
        int TC = 1;
        if (TR < 256) TC = 256 / TR;
        if (et) {
            int T = round_up(std::atoi(et), 64);
            if (T > 1024) T = 1024;
            if (T >= round_up(M, 64)) {
                RPT = 1;
                TR = round_up(M, 64);
                TC = T / TR;
            } else {
                RPT = (M + T - 1) / T;
                TR = round_up((M + RPT - 1) / RPT, 64);
                TC = 1;
            }
        }
        if (const char* etr = diag_env("T4A_RRLU_TR")) { // experiment: explicit thread grid TR x TC
            TR = round_up(std::atoi(etr), 64);
            RPT = (M + TR - 1) / TR;
            TC = diag_env("T4A_RRLU_TC") ? std::atoi(diag_env("T4A_RRLU_TC")) : 1;
            if (TC < 1) TC = 1;
        }
        if (RPT > 4 || TR > 1024) return false; // beyond the register budget: LDS kernel
        // more, thinner workgroups win once the key table is shared (measured: 3 columns per thread and 230 workgroups
        // beat 4 / 172 by 3.5 % at 685 x 688); fall back to 4 and 8 when that would need more workgroups than CUs
        int CPT = ec ? norm_cpt(std::atoi(ec)) : 3;
        int W = (N + TC * CPT - 1) / (TC * CPT);
        if (!ec && W > maxw) {
            CPT = 4;
            W = (N + TC * CPT - 1) / (TC * CPT);
        }
        if (ew && std::atoi(ew) > 1) {
            W = std::atoi(ew);
            CPT = norm_cpt((N + W * TC - 1) / (W * TC));
        }
        while (W > maxw && CPT < 8) {
            CPT = CPT < 6 ? CPT + 1 : 8; // 3 -> 4 -> 5 -> 6 -> 8 columns per thread
            W = (N + TC * CPT - 1) / (TC * CPT);
        }
        if (W < 1) W = 1;
        if (W > maxw || (long long)W * TC * CPT < N) return false;
        if (TR * TC > ((RPT * CPT > 24) ? 256 : 512)) return false;
        plan.W = W;
        plan.T = TR * TC;
        plan.TR = TR;
        plan.TC = TC;
        plan.RPT = RPT;
        plan.CPT = CPT;
    }
    plan.lds_bytes = reg_smem_layout(M, N, plan.TC * plan.CPT, plan.W == 1, nullptr, nullptr);
    if (plan.lds_bytes > 160 * 1024) return false;
    if (plan.W > 1 && plan.lds_bytes < 84 * 1024) plan.lds_bytes = 84 * 1024; // one workgroup per CU
    *out = plan;
    return true;
}

size_t rrlu_reg_keys_bytes(const RrluRegPlan& plan)
{
    return (size_t)2 * plan.W * 2 * sizeof(unsigned long long);
}
size_t rrlu_reg_cols_bytes(const RrluRegPlan& plan, int M)
{
    const size_t slots = (size_t)(plan.W > RRLU_MAX_COPIES ? plan.W : RRLU_MAX_COPIES); // speculative mode: one per workgroup
    return (size_t)2 * slots * (size_t)M * 2 * sizeof(unsigned long long);
}

void rrlu_reg_launch(const RrluRegPlan& plan, const RrluRegArgs& a, hipStream_t stream, bool keys_zeroed)
{
    // the key table carries 16-bit step tags and must start zeroed: normally the previous launch cleared it
    if (plan.W > 1 && !keys_zeroed) (void)hipMemsetAsync(a.keys, 0, rrlu_reg_keys_bytes(plan), stream);
    switch (plan.RPT) {
    case 1: launch_r<1>(plan, a, stream); break;
    case 2: launch_r<2>(plan, a, stream); break;
    case 3: launch_r<3>(plan, a, stream); break;
    default: launch_r<4>(plan, a, stream); break;
    }
}

} // namespace t4a
