// kernels_rrlu.hip — K2: full-pivot rank-revealing LU for gfx950.
//
// Replaces rrlu_mut (tensor4all-core/src/matrixlu.rs:735-819) with BIT-IDENTICAL pivot choice and
// L/U values:
//   * argmax key = abs_sq = v*v, first strict maximum in column-major order of the PERMUTED trailing
//     block (matrixlu.rs:480-519)  ->  here: max score, ties to the smallest (colpos, rowpos);
//     a NaN never replaces the incumbent but a NaN sitting at (k,k) stays the maximum;
//   * stop tests on p = sqrt(v*v) in the reference's order (matrixlu.rs:757-779);
//   * elimination t = t - x*y with separately rounded multiply and subtract (no FMA: this file is built
//     with -ffp-contract=off), IEEE division for the column (left-orth) / row (right-orth) scaling.
//
// MI355X design (the step chain is latency-bound: 2 flop / 16 B, <= min(M,N) dependent steps):
//   * the matrix is distributed over W workgroups by COLUMN (workgroup w owns columns c = w (mod W)),
//     each slab lives in LDS for the whole factorisation; HBM is touched once on load and once on
//     write-out;
//   * rows and columns are never moved: every workgroup keeps an identical replica of the permutation
//     (rowpos/posrow/colpos/poscol) in LDS and ties are broken on permuted positions;
//   * the rank-1 update of step k is fused with the arg-max search of step k+1;
//   * ONE inter-workgroup exchange per pivot step: every workgroup speculatively publishes its best
//     local candidate {value, position} together with that candidate's whole column into a
//     double-buffered mailbox (write-through sc1 stores, drained, then tagged 8-byte key granules);
//     every workgroup sweeps the W keys (sc1 loads), picks the global winner deterministically and
//     fetches only the winner's column.  The pivot ROW needs no exchange: a workgroup owns entire
//     columns, so u_kj for its columns is local.
//   * hand-off protocol: cdna_hip_programming.md §6 Guideline 16, form R1 (sc1 payload -> every storing
//     wave s_waitcnt vmcnt(0) -> workgroup barrier -> tagged granules by one wave) with all consumer
//     loads of handed-off bytes being agent-scope (sc1) loads; one workgroup per CU (LDS > 80 KiB);
//     every spin is bounded and reports T4A_GPU_KERNEL_TIMEOUT instead of hanging.
#include "kernels.hpp"

#include <atomic>

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

// "a beats b": larger score, ties to the smaller permuted column-major position.
__device__ __forceinline__ bool beats(double sa, unsigned pa, double sb, unsigned pb)
{
    return (sa > sb) || (sa == sb && pa < pb);
}

// score of a value at permuted position pos for the search of step `knext`
__device__ __forceinline__ double score_of(double a, unsigned pos, int knext)
{
    double sc = a * a;
    if (sc != sc) {
        const unsigned diag = ((unsigned)knext << 16) | (unsigned)knext;
        sc = (pos == diag) ? __builtin_huge_val() : -1.0;
    }
    return sc;
}

// Diagnostic phase stamps (only when args.stamps != nullptr; thread 0 of workgroup 0 accumulates shader-clock
// cycles per phase).  Never enabled in timed runs.
#define T4A_STAMP(slot)                                                   \
    do {                                                                  \
        if (stamp_on) {                                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            stamp_acc[slot] += now_ - stamp_last;                         \
            stamp_last = now_;                                            \
        }                                                                 \
    } while (0)

struct Smem {
    double* slab;          // cpw * Mld
    double* lcol;          // Mpad: pivot column of the current step (scaled when left-orthogonal)
    double* urow;          // cpw : pivot row entries of the owned columns
    double* red_sc;        // 16
    double* red_val;       // 16
    double* win_d;         // [0] pivot value  [1] max_error  [2] error
    unsigned* red_pos;     // 16
    int* red_jl;           // 16
    int* win_i;            // [0] winner wg [1] pr (orig row) [2] pc (orig col) [3] stop flag [4] blk jl [5] blk has cand
                           // [6] abort [7] prp [8] pcp
    unsigned* keybuf;      // 4*W payloads of the last key sweep
    unsigned short* rowpos; // M
    unsigned short* posrow; // M
    unsigned short* colpos; // N
    unsigned short* poscol; // N
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__host__ __device__ inline size_t smem_layout(int M, int N, int W, int cpw, int Mld, Smem* s, char* base)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 16);
        return o;
    };
    const size_t Mpad = align_up((size_t)M, 2);
    size_t o_slab = take((size_t)cpw * Mld * 8);
    size_t o_lcol = take(Mpad * 8);
    size_t o_urow = take((size_t)(cpw > 0 ? cpw : 1) * 8);
    size_t o_rsc = take(16 * 8);
    size_t o_rval = take(16 * 8);
    size_t o_wind = take(4 * 8);
    size_t o_rpos = take(16 * 4);
    size_t o_rjl = take(16 * 4);
    size_t o_wini = take(12 * 4);
    size_t o_key = take((size_t)4 * W * 4);
    size_t o_rp = take((size_t)M * 2);
    size_t o_pr = take((size_t)M * 2);
    size_t o_cp = take((size_t)N * 2);
    size_t o_pc = take((size_t)N * 2);
    if (s) {
        s->slab = (double*)(base + o_slab);
        s->lcol = (double*)(base + o_lcol);
        s->urow = (double*)(base + o_urow);
        s->red_sc = (double*)(base + o_rsc);
        s->red_val = (double*)(base + o_rval);
        s->win_d = (double*)(base + o_wind);
        s->red_pos = (unsigned*)(base + o_rpos);
        s->red_jl = (int*)(base + o_rjl);
        s->win_i = (int*)(base + o_wini);
        s->keybuf = (unsigned*)(base + o_key);
        s->rowpos = (unsigned short*)(base + o_rp);
        s->posrow = (unsigned short*)(base + o_pr);
        s->colpos = (unsigned short*)(base + o_cp);
        s->poscol = (unsigned short*)(base + o_pc);
    }
    return off;
}

// Thread-local candidate
struct Cand {
    double sc;
    double val;
    unsigned pos;
    int jl;
};

// One pass over the owned slab: optional rank-1 update of step k, fused with the candidate search for
// step k+1 over the domain {rowpos > k, colpos > k}.
template <bool UPDATE>
__device__ __forceinline__ Cand slab_pass(const RrluArgs& p, const Smem& s, int w, int k, int pc, bool left)
{
    Cand best;
    best.sc = -1.0;
    best.val = 0.0;
    best.pos = NOPOS;
    best.jl = -1;
    const int T = blockDim.x;
    const int tid = threadIdx.x;
    const int knext = k + 1;
    for (int jl = 0, c = w; c < p.N; ++jl, c += p.W) {
        const int cp = s.colpos[c];
        double* col = s.slab + (size_t)jl * p.Mld;
        if (cp <= k) {
            if (UPDATE && left && c == pc) {
                // scale_column_tail (matrixlu.rs:562-577): the owner stores l_i = a_ik / pivot
                for (int i = tid; i < p.M; i += T)
                    if ((int)s.rowpos[i] > k) col[i] = s.lcol[i];
            }
            continue;
        }
        const double u = UPDATE ? s.urow[jl] : 0.0;
        const unsigned cpos = (unsigned)cp << 16;
        for (int i = tid; i < p.M; i += T) {
            const int rp = s.rowpos[i];
            if (rp <= k) continue;
            double a = col[i];
            if (UPDATE) {
                const double prod = s.lcol[i] * u; // update_trailing_submatrix (matrixlu.rs:593-612)
                a = a - prod;
                col[i] = a;
            }
            const unsigned pos = cpos | (unsigned)rp;
            const double sc = score_of(a, pos, knext);
            if (beats(sc, pos, best.sc, best.pos)) {
                best.sc = sc;
                best.val = a;
                best.pos = pos;
                best.jl = jl;
            }
        }
    }
    return best;
}

// Workgroup reduction of the thread candidates. Result (identical in every thread): block winner.
__device__ __forceinline__ Cand block_reduce(Cand c, const Smem& s)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double osc = __shfl_xor(c.sc, off);
        const double oval = __shfl_xor(c.val, off);
        const unsigned opos = (unsigned)__shfl_xor((int)c.pos, off);
        const int ojl = __shfl_xor(c.jl, off);
        if (beats(osc, opos, c.sc, c.pos)) {
            c.sc = osc;
            c.val = oval;
            c.pos = opos;
            c.jl = ojl;
        }
    }
    if (lane == 0) {
        s.red_sc[wave] = c.sc;
        s.red_val[wave] = c.val;
        s.red_pos[wave] = c.pos;
        s.red_jl[wave] = c.jl;
    }
    __syncthreads();
    Cand b;
    b.sc = s.red_sc[0];
    b.val = s.red_val[0];
    b.pos = s.red_pos[0];
    b.jl = s.red_jl[0];
    for (int q = 1; q < nwaves; ++q) {
        const double osc = s.red_sc[q];
        const unsigned opos = s.red_pos[q];
        if (beats(osc, opos, b.sc, b.pos)) {
            b.sc = osc;
            b.val = s.red_val[q];
            b.pos = opos;
            b.jl = s.red_jl[q];
        }
    }
    return b;
}

// Stop tests + permutation swap for step k, executed by ONE thread; fills the win record.
// (matrixlu.rs:757-791)
__device__ __forceinline__ void decide_step(const RrluArgs& p, const Smem& s, int k, int winner_wg, double val,
                                            unsigned pos, double max_error)
{
    const double pivot_abs = sqrt(val * val);
    s.win_d[0] = val;
    s.win_d[2] = pivot_abs; // lu.error = pivot_abs (:758)
    int stop = 0;
    if (k > 0 && (pivot_abs < p.rel_tol * max_error || pivot_abs < p.abs_tol)) stop = 1;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    if (!stop && pivot_abs <= min_pivot_abs) stop = 1;
    s.win_i[3] = stop;
    s.win_i[0] = winner_wg;
    if (!stop) {
        s.win_d[1] = fmax(max_error, pivot_abs);
        const int prp = (int)(pos & 0xFFFFu);
        const int pcp = (int)(pos >> 16);
        const int pr = s.posrow[prp];
        const int pc = s.poscol[pcp];
        // swap rows k <-> prp, cols k <-> pcp in the replicated permutation
        const int rk = s.posrow[k];
        s.posrow[k] = (unsigned short)pr;
        s.posrow[prp] = (unsigned short)rk;
        s.rowpos[rk] = (unsigned short)prp;
        s.rowpos[pr] = (unsigned short)k;
        const int ck = s.poscol[k];
        s.poscol[k] = (unsigned short)pc;
        s.poscol[pcp] = (unsigned short)ck;
        s.colpos[ck] = (unsigned short)pcp;
        s.colpos[pc] = (unsigned short)k;
        s.win_i[1] = pr;
        s.win_i[2] = pc;
    }
}

template <bool SINGLE>
__global__ void __launch_bounds__(1024) rrlu_kernel(RrluArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem s;
    smem_layout(p.M, p.N, p.W, p.cpw, p.Mld, &s, smem_raw);

    const int tid = threadIdx.x;
    const int T = blockDim.x;
    const int w = SINGLE ? 0 : (int)blockIdx.x;
    const bool left = p.left_orth != 0;

    // ---- load the owned columns (coalesced, one HBM pass) and init the permutation replicas ----
    double local_absmax = 0.0;
    for (int jl = 0, c = w; c < p.N; ++jl, c += p.W) {
        const double* src = p.A + (size_t)c * p.M;
        double* col = s.slab + (size_t)jl * p.Mld;
        for (int i = tid; i < p.M; i += T) {
            const double v = src[i];
            col[i] = v;
            const double av = sqrt(v * v);
            if (av > local_absmax) local_absmax = av;
        }
    }
    for (int i = tid; i < p.M; i += T) {
        s.rowpos[i] = (unsigned short)i;
        s.posrow[i] = (unsigned short)i;
    }
    for (int j = tid; j < p.N; j += T) {
        s.colpos[j] = (unsigned short)j;
        s.poscol[j] = (unsigned short)j;
    }
    if (tid == 0) {
        s.win_i[6] = 0;
        s.win_d[1] = 0.0;
        s.win_d[2] = __builtin_nan("");
    }
    // max |a| of the input (max_sample_value bookkeeping, tensorci2.rs:2009-2014); non-negative doubles
    // order like their bit patterns, so an integer atomicMax is exact.
    {
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(local_absmax, off);
            if (o > local_absmax) local_absmax = o;
        }
        if ((tid & 63) == 0 && local_absmax > 0.0)
            atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(local_absmax));
    }
    __syncthreads();

    const bool stamp_on = (p.stamps != nullptr) && w == 0 && tid == 0;
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;

    int npiv = 0;
    double max_error = 0.0;
    double error = __builtin_nan("");
    bool timed_out = false;

    if (p.max_steps > 0) {
        // ---- candidate for step 0 ----
        Cand mine = slab_pass<false>(p, s, w, -1, -1, left);
        Cand blk = block_reduce(mine, s);

        for (int k = 0; k < p.max_steps; ++k) {
            // ================= exchange: find the global winner of step k =================
            if (SINGLE) {
                if (tid == 0) decide_step(p, s, k, 0, blk.val, blk.pos, max_error);
                __syncthreads();
            } else {
                const int par = k & 1;
                const unsigned tag = (unsigned)k + 1u;
                // (D) publish my candidate column, drain, then the tagged key granules
                if (blk.pos != NOPOS) {
                    const double* col = s.slab + (size_t)blk.jl * p.Mld;
                    unsigned long long* dst = p.cols + ((size_t)par * p.W + w) * p.M;
                    for (int i = tid; i < p.M; i += T) st_u64_sc1(dst + i, (unsigned long long)__double_as_longlong(col[i]));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains (R1)
                __syncthreads();
                T4A_STAMP(0);
                if (tid < 4) {
                    const unsigned long long vb = (unsigned long long)__double_as_longlong(blk.val);
                    unsigned payload;
                    if (tid == 0) payload = (unsigned)(vb & 0xFFFFFFFFull);
                    else if (tid == 1) payload = (unsigned)(vb >> 32);
                    else if (tid == 2) payload = blk.pos;
                    else payload = (blk.pos != NOPOS) ? 1u : 0u;
                    st_u64_sc1(p.keys + ((size_t)par * p.W + w) * 4 + tid, ((unsigned long long)tag << 32) | payload);
                }
                // (A) wave 0 sweeps all W keys until every tag matches
                if (tid < 64) {
                    const unsigned long long* kb = p.keys + (size_t)par * p.W * 4;
                    const int total = 4 * p.W;
                    unsigned spins = 0;
                    bool giveup = false;
                    for (;;) {
                        bool ok = true;
                        for (int idx = tid; idx < total; idx += 64) {
                            const unsigned long long x = ld_u64_sc1(kb + idx);
                            ok &= ((unsigned)(x >> 32) == tag);
                            s.keybuf[idx] = (unsigned)x;
                        }
                        if (__all(ok)) break;
                        if (++spins > p.spin_limit) {
                            giveup = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (giveup) {
                        if (tid == 0) {
                            s.win_i[6] = 1;
                            atomicExch(&p.iresult[1], 1);
                        }
                    } else {
                        // winner among W candidates (identical computation in every workgroup)
                        double bsc = -1.0, bval = 0.0;
                        unsigned bpos = NOPOS;
                        int bw = -1;
                        for (int q = tid; q < p.W; q += 64) {
                            const unsigned lo = s.keybuf[4 * q + 0], hi = s.keybuf[4 * q + 1];
                            const unsigned pos = s.keybuf[4 * q + 2];
                            const unsigned has = s.keybuf[4 * q + 3];
                            if (has) {
                                const double v = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                                const double sc = score_of(v, pos, k);
                                if (beats(sc, pos, bsc, bpos)) {
                                    bsc = sc;
                                    bval = v;
                                    bpos = pos;
                                    bw = q;
                                }
                            }
                        }
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) {
                            const double osc = __shfl_xor(bsc, off);
                            const double oval = __shfl_xor(bval, off);
                            const unsigned opos = (unsigned)__shfl_xor((int)bpos, off);
                            const int ow = __shfl_xor(bw, off);
                            if (beats(osc, opos, bsc, bpos)) {
                                bsc = osc;
                                bval = oval;
                                bpos = opos;
                                bw = ow;
                            }
                        }
                        if (tid == 0) decide_step(p, s, k, bw, bval, bpos, max_error);
                    }
                }
                __syncthreads();
                T4A_STAMP(1);
                if (s.win_i[6]) {
                    timed_out = true;
                    break;
                }
            }

            // ================= every thread: read the decision =================
            error = s.win_d[2];
            if (s.win_i[3]) break; // stop tests fired (matrixlu.rs:761-779)
            max_error = s.win_d[1];
            const double pivot = s.win_d[0];
            const int pr = s.win_i[1];
            const int pc = s.win_i[2];
            const int ww = s.win_i[0];

            // (B) fetch the winner's column -> lcol (scaled for left-orth), pivot-row entries -> urow
            if (SINGLE) {
                const int jlw = (pc - w) / p.W;
                const double* col = s.slab + (size_t)jlw * p.Mld;
                for (int i = tid; i < p.M; i += T) {
                    const double raw = col[i];
                    s.lcol[i] = left ? raw / pivot : raw;
                }
            } else {
                const unsigned long long* src = p.cols + ((size_t)(k & 1) * p.W + ww) * p.M;
                for (int i = tid; i < p.M; i += T) {
                    const double raw = __longlong_as_double((long long)ld_u64_sc1(src + i));
                    s.lcol[i] = left ? raw / pivot : raw;
                }
            }
            for (int jl = tid, c = w + tid * p.W; c < p.N; jl += T, c += T * p.W) {
                if ((int)s.colpos[c] > k) {
                    double u = s.slab[(size_t)jl * p.Mld + pr];
                    if (!left) { // scale_row_tail (matrixlu.rs:579-591)
                        u = u / pivot;
                        s.slab[(size_t)jl * p.Mld + pr] = u;
                    }
                    s.urow[jl] = u;
                }
            }
            if (w == 0 && tid == 0) p.pivot_vals[k] = pivot;
            __syncthreads();
            T4A_STAMP(2);

            // (C) rank-1 update of step k fused with the candidate search of step k+1
            npiv = k + 1;
            mine = slab_pass<true>(p, s, w, k, pc, left);
            T4A_STAMP(3);
            blk = block_reduce(mine, s);
            T4A_STAMP(4);
        }
    }

    // ---- results ----
    if (npiv >= (p.M < p.N ? p.M : p.N)) error = 0.0; // matrixlu.rs:811-813
    if (w == 0 && tid == 0) {
        p.iresult[0] = npiv;
        p.dresult[0] = error;
    }
    if (stamp_on)
        for (int q = 0; q < 8; ++q) p.stamps[q] = stamp_acc[q];
    if (timed_out) return;
    __syncthreads();
    if (w == 0) {
        for (int i = tid; i < p.M; i += T) p.row_perm[i] = s.posrow[i];
        for (int j = tid; j < p.N; j += T) p.col_perm[j] = s.poscol[j];
    }
    // write-out in permuted coordinates + NaN check of the L / U regions (matrixlu.rs:653-662)
    int nan_seen = 0;
    for (int jl = 0, c = w; c < p.N; ++jl, c += p.W) {
        const int cp = s.colpos[c];
        const double* col = s.slab + (size_t)jl * p.Mld;
        for (int rp = tid; rp < p.M; rp += T) {
            const double v = col[s.posrow[rp]];
            const bool in_l = (cp < npiv) && (rp >= cp);
            const bool in_u = (rp < npiv) && (cp >= rp);
            // A diagonal NaN is hidden in the factor whose diagonal is forced to 1 but shows in the other
            // one (matrixlu.rs:643-662), so any NaN inside L u U is an error.
            if ((in_l || in_u) && v != v) nan_seen = 1;
            if (p.Aout) p.Aout[(size_t)cp * p.M + rp] = v;
        }
    }
    if (nan_seen) atomicExch(&p.iresult[2], 1);
}

} // namespace

RrluPlan rrlu_make_plan(int M, int N, int num_cus)
{
    RrluPlan plan;
    const char* ew = diag_env("T4A_RRLU_W");
    const char* et = diag_env("T4A_RRLU_T");
    int T = 256;
    if (et) T = std::atoi(et);
    if (T < 64) T = 64;
    if (T > 1024) T = 1024;
    T = (T / 64) * 64;
    const size_t elems = (size_t)M * (size_t)N;
    int W;
    if (ew) {
        W = std::atoi(ew);
    } else if (elems <= 96 * 96) {
        W = 1;
    } else {
        // aim at ~6 columns (<= ~36 KiB) per workgroup; the exchange cost grows slowly with W while the
        // per-step update time shrinks as 1/W
        W = (N + 5) / 6;
        const int maxw = num_cus > 16 ? num_cus - 8 : num_cus;
        if (W > maxw) W = maxw;
    }
    if (W < 1) W = 1;
    if (W > N) W = N > 0 ? N : 1;
    if (W > num_cus) W = num_cus;
    // LDS capacity: grow W until the slab fits
    for (;;) {
        plan.W = W;
        plan.cpw = (N + W - 1) / W;
        plan.Mld = (int)(((size_t)M + 1) / 2 * 2);
        plan.lds_bytes = smem_layout(M, N, W, plan.cpw, plan.Mld, nullptr, nullptr);
        if (plan.lds_bytes <= 160 * 1024 || W >= num_cus || W >= N) break;
        W = W * 2 > num_cus ? num_cus : W * 2;
        if (W > N) W = N;
    }
    if (plan.W > 1 && plan.lds_bytes < 84 * 1024) plan.lds_bytes = 84 * 1024; // one workgroup per CU
    if (plan.W == 1 && !et) {
        // single workgroup: use more threads for bigger slabs
        T = elems >= 4096 ? 1024 : (elems >= 1024 ? 512 : 256);
    }
    plan.T = T;
    return plan;
}

size_t rrlu_keys_bytes(const RrluPlan& plan) { return (size_t)2 * plan.W * 4 * sizeof(unsigned long long); }
size_t rrlu_cols_bytes(const RrluPlan& plan, int M) { return (size_t)2 * plan.W * (size_t)M * sizeof(unsigned long long); }

void rrlu_launch(const RrluPlan& plan, const RrluArgs& a, hipStream_t stream)
{
    static std::atomic<bool> attr_set{false}; // (launches come from several host threads; setting the attribute twice is harmless)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (plan.W == 1) {
        hipLaunchKernelGGL(rrlu_kernel<true>, dim3(1), dim3(plan.T), plan.lds_bytes, stream, a);
    } else {
        (void)hipMemsetAsync(a.keys, 0, rrlu_keys_bytes(plan), stream);
        hipLaunchKernelGGL(rrlu_kernel<false>, dim3(plan.W), dim3(plan.T), plan.lds_bytes, stream, a);
    }
}

} // namespace t4a
