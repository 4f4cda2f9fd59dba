// kernels_rrlu_xcd.hip — K2 fast path, round 2: register-resident full-pivot rank-revealing LU whose workgroups all sit on
// ONE XCD of the MI355X, so that the per-pivot exchange runs through that XCD's own L2 (plain stores keep the line there,
// `sc1` loads bypass the reader's L1) instead of write-through stores that leave the XCD.
//
// Same contract as kernels_rrlu_reg.hip: bit-identical to rrlu_mut (tensor4all-core/src/matrixlu.rs:735-819) for the
// left-orthogonal elimination; a right-orthogonal factorisation runs as the left-orthogonal one of A^T with the row-major tie
// order (ROWMAJOR).  Arg-max semantics: matrixlu.rs:480-519 (key v*v, first strict maximum in column-major order of the
// permuted trailing block, NaN never replaces the incumbent, a NaN at (k,k) stays).
//
// Measured background (tools/xcd_bench.hip, MI355X): workgroups b and b + 8 of a grid land on the same XCD (strict round robin,
// 1024 of 1024 blocks); an all-gather of 32 keys through one XCD's L2 costs ~360 cycles against ~1 500 - 2 300 for the
// write-through all-gather over the whole chip, a pivot column hand-off 300 - 400 cycles against ~1 800.
//
// Structure:
//   * grid = 8 * W workgroups of 512 threads; only the W workgroups whose HW_REG_XCC_ID equals `xcc` take part (a ticket gives
//     them their rank), the others return at once.  If the placement assumption ever fails the bounded spins give up and
//     the host re-runs the factorisation with the chip-wide kernel.
//   * every WAVE is an agent that owns whole columns: agent g = rank * 8 + wave holds columns g + NW * q (q < CPT) with rows
//     lane + 64 * r (r < RPT) in registers, so the candidate search, the pivot-row broadcast (v_readlane) and the speculative
//     publication of the candidate column need no workgroup-level synchronisation at all;
//   * per pivot step: rank-1 update (in place) fused with per-column maxima of |a| -> wave maximum (DPP) -> EARLY key {candidate
//     magnitude} -> position of the candidate (the early keys travel meanwhile) -> FULL key {value, position, row index, column
//     slot} and, when the candidate is within `spec_frac` of the previous pivot, its column (16 bytes per row) -> wave 0 of
//     every workgroup gathers the NW early keys (sc1 loads, L2 hits), picks the winner from the magnitudes and reads only the
//     winner's full key (ties and special values: exact comparison over all full keys), stop rules, record -> barrier (B) ->
//     all 512 threads fetch their rows of the winner's column, divide by the pivot (shared refined reciprocal, bitwise the
//     IEEE quotient) and park l in LDS -> barrier (C).  Two barriers per step, no LDS traffic on the search side.
//   * the kernel is bound by its instruction stream and its scalar registers, not by arithmetic: compile-time LDS layout, one
//     buffer resource for the whole mailbox, column slots padded to 64 * RPT rows (no row masks), clamped key addresses (no
//     agent masks), rare paths fed through opaque registers so that nothing of them is hoisted into the step loop, tied
//     operands for the update (DESIGN.md 5.1 has the measurements behind each of these).
//   * granules carry a launch-salted tag folded with their payload (d3 = tag ^ d0 ^ d1 ^ d2), so a torn or stale granule
//     never passes the check and no buffer has to be cleared between launches.
#include "kernels_rrlu_xcd_common.hpp"

namespace t4a {

namespace {

template <int RPT, int CPT, bool ROWMAJOR>
__device__ __forceinline__ void rrlu_xcd_body(const RrluXcdArgs& p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = XcdLds<RPT>;
    constexpr int LSTR = L::LSTR;
    constexpr int MP = 64 * RPT; // rows of a published column slot (rows beyond M carry zeros)
    double* const lbuf = reinterpret_cast<double*>(smem_raw + L::o_l);
    int* const win_i = reinterpret_cast<int*>(smem_raw + L::o_wi);
    unsigned long long* const lds_ptrs = reinterpret_cast<unsigned long long*>(smem_raw + L::o_pp);
    unsigned long long* const lds_stamps = reinterpret_cast<unsigned long long*>(smem_raw + L::o_st);
    double* const lds_pivots = reinterpret_cast<double*>(smem_raw + L::o_pv);
    unsigned short* const posrow = reinterpret_cast<unsigned short*>(smem_raw + L::o_pr);
    unsigned short* const rowpos = reinterpret_cast<unsigned short*>(smem_raw + L::o_rp);
    unsigned short* const poscol = reinterpret_cast<unsigned short*>(smem_raw + L::o_pc);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t_entry = (kXcdStamps && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- election: only the workgroups that landed on the wanted XCD take part ----
    if (tid == 0) {
        int rank = -1;
        if ((int)xcc_id() == p.xcc) {
            const unsigned t = atomicAdd(p.ticket, 1u) - p.ticket_base;
            if (t < (unsigned)p.W) rank = (int)t;
        }
        win_i[6] = rank;
        win_i[4] = 0;
        win_i[5] = 0; // the first diagonal element is (row 0, column 0)
        lds_ptrs[0] = (unsigned long long)p.iresult;
        lds_ptrs[1] = (unsigned long long)p.h_block;
        for (int e = 0; e < 16; ++e) lds_stamps[e] = 0ull;
    }
    __syncthreads();
    const int rank = __builtin_amdgcn_readfirstlane(win_i[6]);
    if (rank < 0) {
        // pass-through workgroup (another XCD).  Bond chain: it evaluates its share of the NEXT bond's candidate matrix first
        if (p.spec.out && p.dims) {
            const int m_spec = p.dims[2] != 0 ? 0 : (p.dims_swap ? p.dims[1] : p.dims[0]);
            if (m_spec > 0 && m_spec <= p.M)
                xcd_spec_work(reinterpret_cast<const XcdSpecArgs*>(kernarg_base() + offsetof(RrluXcdArgs, spec)), m_spec, win_i + 8);
        }
        return;
    }
    const unsigned long long ts_begin = p.ts_u64 > 0 ? wall_clock64() : 0ull;
    const unsigned long long t_elected = (kXcdStamps && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
    const int NW = p.W * XWAVES;
    const int g = rank * XWAVES + wave; // agent id
    // bond chain: the real dimensions come from device memory (the launch was planned for the upper bounds p.M x p.N: rows
    // beyond M are padding zeros like those beyond p.M always were, columns beyond N have no owner)
    int M = p.M, N = p.N, max_steps = p.max_steps;
    int lda = p.M; // leading dimension of the source matrix
    if (p.dims) {
        const int d0 = __builtin_amdgcn_readfirstlane(p.dims[0]), d1 = __builtin_amdgcn_readfirstlane(p.dims[1]);
        M = p.dims_swap ? d1 : d0;
        N = p.dims_swap ? d0 : d1;
        if (M > p.M || N > p.N) M = N = 0; // (cannot happen: the plan is made for upper bounds; never index out of the plan)
        const int mn = M < N ? M : N;
        max_steps = max_steps < mn ? max_steps : mn;
        if (mn <= 0) return; // poisoned bond: nothing to do (no completion token: the next preparation kernel sees that)
        lda = p.rowmap ? __builtin_amdgcn_readfirstlane(p.dims[3]) : M;
    }
    M = __builtin_amdgcn_readfirstlane(M); // (wave-uniform by construction; tell the compiler)
    N = __builtin_amdgcn_readfirstlane(N);
    max_steps = __builtin_amdgcn_readfirstlane(max_steps);
    lda = __builtin_amdgcn_readfirstlane(lda);

    // ---- my columns (per wave): g + NW q; my rows (per lane): lane + 64 r ----
    int cpos[CPT]; // current position of column g + NW q (-1: beyond N); wave-uniform
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int c = g + NW * q;
        cpos[q] = c < N ? c : -1;
    }
    xvec<RPT> a[CPT]; // ext vectors: the two run-time row-slot accesses become s_set_gpr_idx moves
    double local_sqmax = 0.0;
    // every load is issued before the first one is consumed (clamped addresses instead of branches): the whole matrix is
    // one round trip to memory per lane, not RPT * CPT dependent ones
    // (row validity as a per-lane bit mask, tested again for every column behind a compiler barrier: RPT x CPT lane masks in scalar
    // register pairs are what spilled the scalar register file of the wide instantiations, see kernels_rrlu_xcd2.hip)
    int srow[RPT]; // source row of my slot rows (bond chain: through the row map of the speculative candidate matrix); 0 beyond M
    unsigned rowmask = 0u;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = lane + 64 * r;
        const bool rok = i < M;
        rowmask |= rok ? (1u << r) : 0u;
        srow[r] = rok ? i : 0;
        if (p.rowmap) srow[r] = p.rowmap[rok ? i : 0];
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const bool cok = cpos[q] >= 0; // (wave-uniform)
        const double* const colp = p.A + (cok ? (size_t)(g + NW * q) * lda : (size_t)0);
        const unsigned rm = (unsigned)opaque_v((int)rowmask);
#pragma unroll
        for (int r = 0; r < RPT; ++r) a[q][r] = colp[(cok && ((rm >> r) & 1u)) ? srow[r] : 0];
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const bool cok = cpos[q] >= 0;
        const unsigned rm = (unsigned)opaque_v((int)rowmask);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool ok = cok && ((rm >> r) & 1u);
            const double v = ok ? a[q][r] : 0.0;
            const double sqv = v * v; // max sqrt(v*v) == sqrt(max v*v): one square root per lane below
            if (sqv > local_sqmax) local_sqmax = sqv; // (NaN never enters, like the branchy form)
            a[q][r] = v;
        }
    }
    for (int i = tid; i < M; i += XT) {
        posrow[i] = (unsigned short)i;
        rowpos[i] = (unsigned short)i;
    }
    for (int j = tid; j < N; j += XT) poscol[j] = (unsigned short)j;
    for (int e = tid; e < 64 * LSTR; e += XT) lbuf[e] = 0.0;
    {
        const double wm = wave_max_f64(sqrt(local_sqmax));
        if (lane == 0 && wm > 0.0)
            atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(wm));
    }
    __syncthreads();

    const bool stamp_on = kXcdStamps && (p.stamps != nullptr) && rank == 0 && tid == 0;
    unsigned long long stamp_last = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_loop = stamp_last;

    // one mailbox: [2][NW] early keys (candidate magnitude only), [2][NW] full keys, then [2][NW][MP] column rows (one buffer
    // resource for every store and load of the exchange)
    const unsigned k2_base = 2u * (unsigned)NW * 16u;
    const unsigned cols_base = 4u * (unsigned)NW * 16u;
    const __amdgpu_buffer_rsrc_t mail =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.keys, 0, (int)(cols_base + 2u * (unsigned)NW * (unsigned)MP * 16u), 0x00020000);

    int npiv = 0;
    int nan_seen = 0;
    double max_error = 0.0;             // kept by the polling waves
    double error = __builtin_nan("");
    bool timed_out = false;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    double u[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) u[q] = 0.0;
    double prev_sq = __builtin_huge_val(); // nobody speculates on the first step
    // three launch constants of the step loop live in VECTOR registers: left in the kernel arguments they are re-read in every
    // step (the scalar registers do not hold them across the loop), each time with its load latency exposed — the tolerances
    // in the middle of the polling wave's stop test, i.e. on the critical path of the step
    double spec_frac = p.spec_frac, rel_tol_v = p.rel_tol, abs_tol_v = p.abs_tol;
    asm volatile("" : "+v"(spec_frac), "+v"(rel_tol_v), "+v"(abs_tol_v));
    constexpr unsigned XSPIN = 1u << 20; // bounded spins: a hand-off that does not arrive makes the launch give up
    int dpk = 0; // next diagonal element: row | column << 10 (worked out by the polling wave one step ahead)

    // maxima of the untouched matrix for the first arg-max
    double mq[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        mq[q] = -1.0;
        if (cpos[q] >= 0) {
            double m0 = -1.0, m1 = -1.0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (r & 1) m1 = vmax_abs(m1, a[q][r]);
                else m0 = vmax_abs(m0, a[q][r]);
            }
            mq[q] = vmax(m0, m1);
        }
    }

    for (int kn = 0; kn < max_steps; ++kn) {
        const int k = kn - 1; // rows / columns at positions > k form the trailing block searched for pivot kn
        const unsigned diagkey = ((unsigned)kn << 10) | (unsigned)kn;
        const int par = kn & 1;
        const unsigned tag = (p.salt << 16) | (unsigned)(kn + 1);
        int cps[CPT]; // column positions as scalars
#pragma unroll
        for (int q = 0; q < CPT; ++q) cps[q] = __builtin_amdgcn_readfirstlane(cpos[q]);
        // a NaN sitting on the next diagonal element wins outright (it is the reference's initial incumbent)
        {
            const int dr = dpk & 1023, dc = dpk >> 10;
#pragma unroll
            for (int q = 0; q < CPT; ++q)
                if (g + NW * q == dc) { // one wave of the chip
                    const double dv = a[q][dr >> 6];
                    if (__ballot((lane == (dr & 63)) & (dv != dv)) != 0ull) mq[q] = __builtin_huge_val();
                }
        }
        // ---- wave arg-max: (max score, smallest position among the maxima, value there), all wave-uniform ----
        double m = mq[0];
#pragma unroll
        for (int q = 1; q < CPT; ++q) m = vmax(m, mq[q]);
        const double wmax = wave_max_f64(m);
        const double sq = wmax * wmax; // the winning score v*v of this agent
        // ---- early key: the magnitude of the candidate goes out before its position is known.  In the normal case (one
        // agent holds the largest |v|, its square a normal number) the magnitudes alone decide the winner, so the pollers
        // gather them while every agent is still looking for the row slot and the position of its candidate; the full key
        // follows below and is only read for the winner.  (No candidate: 0, which sends the pick to the exact path.)
        const int kslot = (par * NW + g) * 16;
        {
            const double k1 = (wmax >= 0.0) ? wmax : 0.0;
            u32x4 kv;
            kv.x = lo32(k1);
            kv.y = hi32(k1);
            kv.z = 0u;
            kv.w = tag ^ kv.x ^ kv.y;
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(kv, mail, kslot, 0, 0);
        }
        u32x4 kg[4], kh[4];
        unsigned wpos = XKEY_NONE;     // position key of the candidate
        double cval = 0.0;             // its value
        int cirow = 0, qstar = 0;      // its row index and my column slot
        if (wmax >= 0.0) {
            bool done = false;
            // while v*v is a normal number, distinct |v| have distinct squares, so the equality sweep can compare |v| itself
            // (and the slots of rows that are already pivoted hold zeros, which cannot match)
            if ((sq >= 2.2250738585072014e-308) && (sq < __builtin_huge_val())) {
                unsigned long long bq[CPT];
                int nhit = 0;
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    bq[q] = __ballot(mq[q] == wmax); // (columns outside the trailing block keep mq = -1)
                    nhit += __builtin_popcountll(bq[q]);
                }
                if (nhit == 1) { // one lane of one column holds the maximum: the normal case
#pragma unroll
                    for (int q = 0; q < CPT; ++q)
                        if (bq[q] != 0ull) {
                            const int hl = (int)__builtin_ctzll(bq[q]);
                            // which row slot of that lane: bit RPT - 1 - r of `bits` says slot r holds the maximum
                            // (compare + add-with-carry per slot: bits = 2 bits + (|a| == wmax))
                            unsigned bits = 0u;
#pragma unroll
                            for (int r = 0; r < RPT; ++r)
                                asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"((double)a[q][r]), "s"(wmax) : "vcc");
                            const unsigned hb_ = (unsigned)__builtin_amdgcn_readlane((int)bits, hl);
                            if (__builtin_popcount(hb_) == 1) {
                                const int rstar = RPT - 1 - (int)__builtin_ctz(hb_);
                                cirow = hl + 64 * rstar;
                                const unsigned rp_ = (unsigned)__builtin_amdgcn_readfirstlane((int)rowpos[cirow]);
                                wpos = ROWMAJOR ? ((rp_ << 10) | (unsigned)cps[q]) : (((unsigned)cps[q] << 10) | rp_);
                                cval = readlane_f64(a[q][rstar], hl);
                                qstar = q;
                                done = true;
                            }
                        }
                }
            }
            if (!done) {
                // ties, zero / subnormal / infinite scores and the NaN incumbent: exact sweep on the squares
                unsigned mypos = XNOPOS;
                double myval = 0.0;
                int myrow = 0, myq = 0;
                const int lane_o = opaque_v(lane), M_o = opaque_s(M); // (nothing of this rare path is hoisted out of the step loop)
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const bool qhit = (mq[q] >= 0.0) & (mq[q] * mq[q] == sq);
                    if (__ballot(qhit) != 0ull) {
#pragma unroll
                        for (int r = 0; r < RPT; ++r) {
                            const int i = lane_o + 64 * r;
                            const unsigned rp_ = rowpos[i < M_o ? i : 0];
                            const unsigned key = ROWMAJOR ? ((rp_ << 10) | (unsigned)cps[q]) : (((unsigned)cps[q] << 10) | rp_);
                            const double av = a[q][r];
                            const double sc = av * av;
                            const bool hit = qhit & (i < M_o) & ((int)rp_ > k) &
                                             ((sc == sq) | ((key == diagkey) & (sc != sc) & (sq == __builtin_huge_val())));
                            if (hit && key < mypos) {
                                mypos = key;
                                myval = av;
                                myrow = i;
                                myq = q;
                            }
                        }
                    }
                }
                const unsigned wp = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(mypos));
                if (wp != XNOPOS) {
                    const unsigned long long sel = __ballot(mypos == wp);
                    const int hl = (int)__builtin_ctzll(sel);
                    wpos = wp;
                    cval = readlane_f64(myval, hl);
                    cirow = __builtin_amdgcn_readlane(myrow, hl);
                    qstar = __builtin_amdgcn_readlane(myq, hl);
                }
            }
        }
        XSTAMP(1);
        // ---- full key: value, position, row index, column slot (all fields are wave-uniform) ----
        {
            const unsigned meta = wpos | ((unsigned)cirow << 20) | ((unsigned)qstar << 30);
            u32x4 kv;
            kv.x = lo32(cval);
            kv.y = hi32(cval);
            kv.z = meta;
            kv.w = tag ^ kv.x ^ kv.y ^ meta;
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(kv, mail, (int)k2_base + kslot, 0, 0);
        }
        // the polling wave sweeps the early keys now — they left their agents a whole position search ago, so this first sweep
        // normally finds them all (a load only sees what had reached the L2 when it was served: issued earlier it comes back
        // stale and costs a second round trip).  Every lane fetches four keys; lanes beyond NW re-read the last key (a valid
        // duplicate), so neither the arrival check nor the maximum needs a mask or a count of live groups.  (The wave's own
        // stores are acknowledged long before it needs this data, so it publishes like everybody else.)
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kg[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (par * NW + min(lane + 64 * j, NW - 1)) * 16, 0, BUF_SC1);
        }
        // thresholded speculative publication of the candidate column: pivots shrink slowly, so the next winner is almost
        // always an agent whose candidate is close to the previous pivot; its column is then already in the L2 when the
        // keys have been gathered
        const bool early_pub = (wave != 0) && (wpos != XKEY_NONE) && (sq >= spec_frac * prev_sq); // (the polling wave never stores a column early: those stores would sit in front of its key loads)
        const int myslot = (int)cols_base + ((par * NW + g) * MP + lane) * 16; // byte offset of my row `lane` in the mailbox
        if (early_pub) {
            if (qstar == 0) xcd_publish_column<0, RPT>(a[0], mail, myslot, tag);
            if constexpr (CPT > 1) if (qstar == 1) xcd_publish_column<1, RPT>(a[1], mail, myslot, tag);
            if constexpr (CPT > 2) if (qstar == 2) xcd_publish_column<2, RPT>(a[2], mail, myslot, tag);
            if constexpr (CPT > 3) if (qstar == 3) xcd_publish_column<3, RPT>(a[3], mail, myslot, tag);
        }
        XSTAMP(2);

        // ---- wave 0 gathers the NW keys and decides (matrixlu.rs:480-519 across agents, stop rules :757-781) ----
        if (wave == 0) {
            // who sits at position kn and kn + 1 now (tables are stable between barrier (C) and the next (B))
            const int rk_ = posrow[kn], ck_ = poscol[kn];
            const int rn_ = (kn + 1 < M) ? (int)posrow[kn + 1] : 0, cn_ = (kn + 1 < N) ? (int)poscol[kn + 1] : 0;
            unsigned spins = 0;
            bool giveup = false;
            XSTAMP(6);
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) ok &= ((kg[j].x ^ kg[j].y ^ kg[j].z ^ kg[j].w) == tag);
                if (__all(ok)) break;
                xcd_poll_again();
                if (++spins > XSPIN) {
                    giveup = true;
                    break;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    kg[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (par * NW + min(lane + 64 * j, NW - 1)) * 16, 0, BUF_SC1);
            }
            if (stamp_on) lds_stamps[5] += spins;
            // the full keys: fetched now (every agent stored its own before it could have seen this step's early keys complete...
            // almost: a late one is fetched again below), in flight while the early ones are examined
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + min(lane + 64 * j, NW - 1)) * 16, 0, BUF_SC1);
            XSTAMP(8);
            double wv = 0.0;
            unsigned wm_ = 0u;
            int wa_ = 0;
            if (!giveup) {
                // winner over all agents.  Normal case: the largest candidate magnitude, its square a normal number (distinct
                // |v| <=> distinct scores), held by exactly one early key: one maximum reduction decides, and only the winner's
                // full key is needed.  (A duplicate of the last key can only push the count above one: then the exact path
                // decides.  A NaN incumbent travels as +inf, no candidate as 0: both end up on the exact path.)
                bool decided = false;
                {
                    double lm = -1.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lm = vmax(lm, mk_f64(kg[j].x, kg[j].y));
                    const double gm = wave_max_f64(lm);
                    const double gsq = gm * gm;
                    if ((gsq >= 2.2250738585072014e-308) && (gsq < __builtin_huge_val())) {
                        unsigned long long hb[4];
                        int nh = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hb[j] = __ballot(mk_f64(kg[j].x, kg[j].y) == gm);
                            nh += __builtin_popcountll(hb[j]);
                        }
                        if (nh == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (hb[j] != 0ull) {
                                    const int hl = (int)__builtin_ctzll(hb[j]);
                                    wa_ = hl + 64 * j;
                                    // the winner's full key: normally long there; otherwise fetched again until it is
                                    unsigned kx, ky, kz, kw;
                                    for (;;) {
                                        kx = (unsigned)__builtin_amdgcn_readlane((int)kh[j].x, hl);
                                        ky = (unsigned)__builtin_amdgcn_readlane((int)kh[j].y, hl);
                                        kz = (unsigned)__builtin_amdgcn_readlane((int)kh[j].z, hl);
                                        kw = (unsigned)__builtin_amdgcn_readlane((int)kh[j].w, hl);
                                        if ((kx ^ ky ^ kz ^ kw) == tag) break;
                                        xcd_poll_again();
                                        if (++spins > XSPIN) {
                                            giveup = true;
                                            break;
                                        }
                                        kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + min(lane + 64 * j, NW - 1)) * 16, 0, BUF_SC1);
                                    }
                                    wv = mk_f64(kx, ky);
                                    wm_ = kz;
                                }
                            decided = true;
                        }
                    }
                }
                XSTAMP(14);
                if (!decided) {
                    // ties between agents, zero / subnormal / infinite scores, the NaN incumbent: exact comparison of
                    // (v*v, position key) over the FULL keys.  The only NaN a key can carry is the incumbent on the diagonal,
                    // which wins outright; an agent without candidate carries value 0 and the largest position key.
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ok &= ((kh[j].x ^ kh[j].y ^ kh[j].z ^ kh[j].w) == tag);
                        if (__all(ok)) break;
                        xcd_poll_again();
                        if (++spins > XSPIN) {
                            giveup = true;
                            break;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + min(lane + 64 * j, NW - 1)) * 16, 0, BUF_SC1);
                    }
                    double csc = -1.0, cv = 0.0;
                    unsigned cpk = XNOPOS, cmeta = 0u;
                    int cag = 0;
                    const int lane_o = opaque_v(lane);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ag = lane_o + 64 * j;
                        const unsigned pk = (ag < NW) ? (kh[j].z & 0xFFFFFu) : XNOPOS;
                        const double v = mk_f64(kh[j].x, kh[j].y);
                        double sc = v * v;
                        sc = (sc != sc) ? __builtin_huge_val() : sc;
                        sc = (ag < NW) ? sc : -2.0;
                        const bool better = (sc > csc) | ((sc == csc) & (pk < cpk));
                        csc = better ? sc : csc;
                        cv = better ? v : cv;
                        cpk = better ? pk : cpk;
                        cmeta = better ? kh[j].z : cmeta;
                        cag = better ? ag : cag;
                    }
                    const double gmax = wave_max_f64(csc);
                    const unsigned gpos = wave_min_u32((csc == gmax) ? cpk : XNOPOS);
                    const unsigned long long sel = __ballot((csc == gmax) & (cpk == gpos));
                    const int wl = sel ? (int)__builtin_ctzll(sel) : 0;
                    wv = readlane_f64(cv, wl);
                    wm_ = (unsigned)__builtin_amdgcn_readlane((int)cmeta, wl);
                    wa_ = __builtin_amdgcn_readlane(cag, wl);
                }
            }
            if (giveup) {
                if (lane == 0) {
                    win_i[4] = 1;
                    atomicExch(reinterpret_cast<int*>(lds_ptrs[0]) + 1, 1);
                    if (lds_ptrs[1]) reinterpret_cast<volatile int*>(lds_ptrs[1])[5] = 1;
                }
            } else {
                // stop tests on the pivot magnitude sqrt(v*v), in the reference's order; while v*v is a normal number the
                // square root of the rounded square is |v| itself (the software square root stays on the cold path)
                const double wsq = wv * wv;
                double pivot_abs = __builtin_fabs(wv);
                if (!(wsq >= 2.2250738585072014e-308 && wsq < __builtin_huge_val())) pivot_abs = sqrt(mk_f64((unsigned)opaque_v((int)lo32(wsq)), hi32(wsq)));
                error = pivot_abs;
                int stop = 0;
                if (kn > 0 && (pivot_abs < rel_tol_v * max_error || pivot_abs < abs_tol_v)) stop = 1;
                else if (pivot_abs <= min_pivot_abs) stop = 1;
                else max_error = fmax(max_error, pivot_abs);
                // permutation bookkeeping for everybody: who sits at position kn now, and the next diagonal element
                const unsigned wk_ = wm_ & 0xFFFFFu;
                const int prp_ = (int)(ROWMAJOR ? (wk_ >> 10) : (wk_ & 1023u));
                const int pcp_ = (int)(ROWMAJOR ? (wk_ & 1023u) : (wk_ >> 10));
                XSTAMP(15);
                if (lane == 0) {
                    int4 rec;
                    rec.x = (int)lo32(wv);
                    rec.y = (int)hi32(wv);
                    rec.z = (int)wm_;
                    rec.w = wa_ | (rk_ << 8) | (ck_ << 18) | (stop << 28);
                    *reinterpret_cast<int4*>(win_i) = rec;
                    // next diagonal element: the row / column that moves from kn to the pivot's old position, or the untouched one
                    win_i[5] = ((prp_ == kn + 1) ? rk_ : rn_) | (((pcp_ == kn + 1) ? ck_ : cn_) << 10);
                }
                XSTAMP(9);
            }
        }
        __syncthreads(); // (B)
        XSTAMP(3);
        // the record and the flags in ONE LDS round trip (left to itself the compiler reads the abort flag, the stop bit and the
        // rest one after the other, each with its own wait)
        int4 rec;
        int2 rec2; // [0] abort [1] next diagonal element
        {
            int zero = 0;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(rec), "=&v"(rec2)
                         : "v"(zero), "n"(L::o_wi), "n"(L::o_wi + 16)
                         : "memory");
        }
        if (__builtin_amdgcn_readfirstlane(rec2.x)) {
            timed_out = true;
            break;
        }
        dpk = __builtin_amdgcn_readfirstlane(rec2.y);
        const int rpk = __builtin_amdgcn_readfirstlane(rec.w);
        if (rpk >> 28) break; // stop
        const double wval = mk_f64((unsigned)__builtin_amdgcn_readfirstlane(rec.x), (unsigned)__builtin_amdgcn_readfirstlane(rec.y));
        const unsigned wmeta = (unsigned)__builtin_amdgcn_readfirstlane(rec.z);
        const int wag = rpk & 255, rk = (rpk >> 8) & 1023, ck = (rpk >> 18) & 1023;
        const unsigned wkey = wmeta & 0xFFFFFu;
        const int prp = (int)(ROWMAJOR ? (wkey >> 10) : (wkey & 1023u));
        const int pcp = (int)(ROWMAJOR ? (wkey & 1023u) : (wkey >> 10));
        const int irow_p = (int)((wmeta >> 20) & 1023u);
        const int pc = wag + NW * (int)(wmeta >> 30); // original index of the pivot column

        // the winner did not speculate: its column goes out now
        if (g == wag && !early_pub) {
            if (qstar == 0) xcd_publish_column<0, RPT>(a[0], mail, myslot, tag);
            if constexpr (CPT > 1) if (qstar == 1) xcd_publish_column<1, RPT>(a[1], mail, myslot, tag);
            if constexpr (CPT > 2) if (qstar == 2) xcd_publish_column<2, RPT>(a[2], mail, myslot, tag);
            if constexpr (CPT > 3) if (qstar == 3) xcd_publish_column<3, RPT>(a[3], mail, myslot, tag);
        }
        // thread tid fetches slot rows tid + XT j = lane + 64 (wave + XWAVES j): valid while wave + XWAVES j < RPT (wave-uniform)
        constexpr int XR = (RPT + XWAVES - 1) / XWAVES;
        const int slot_off = (int)cols_base + ((par * NW + wag) * MP + tid) * 16;
        u32x4 cc[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (wave + XWAVES * j < RPT) cc[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, slot_off + j * XT * 16, 0, BUF_SC1);
        // (the shared reciprocal of the pivot does not depend on the column: it is formed while the column travels)
        const bool p_mid = exp_mid(wval);
        const double rp = refined_rcp(wval);
        XSTAMP(10);

        // ---- while the column travels: permutation tables, pivot row ----
        // (nobody reads the tables between barriers (B) and (C); one lane each of three waves that do not poll, so that no wave
        // reaches barrier (C) a whole table update later than the others)
        if (tid == 64) {
            posrow[prp] = (unsigned short)rk;
            posrow[kn] = (unsigned short)irow_p;
        }
        if (tid == 128) {
            rowpos[rk] = (unsigned short)prp;
            rowpos[irow_p] = (unsigned short)kn;
        }
        if (tid == 192) {
            poscol[pcp] = (unsigned short)ck;
            poscol[kn] = (unsigned short)pc;
            lds_pivots[kn] = wval;
        }
        prev_sq = wval * wval;
        // columns: ck (at kn) goes to pcp, the pivot column pc goes to kn
#pragma unroll
        for (int q = 0; q < CPT; ++q)
            if (g + NW * q == ck) cpos[q] = pcp;
#pragma unroll
        for (int q = 0; q < CPT; ++q)
            if (g + NW * q == pc) cpos[q] = kn;
        {
            // the pivot row: its entries in the columns of the trailing block (and the pivot column) are the finished row kn
            // of U.  They are broadcast as u, saved to the side buffer and replaced by zeros, so that from now on the row
            // takes part in the update as l = 0 / a = 0 without any row mask.
            const int ls = irow_p & 63, rs = irow_p >> 6;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const double av = a[q][rs];
                u[q] = readlane_f64(av, ls);
                if (__builtin_amdgcn_readfirstlane(cpos[q]) >= kn) {
                    if (u[q] != u[q]) nan_seen = 1;
                    if (p.urows && lane == ls) p.urows[(unsigned)(kn * N + (g + NW * q))] = u[q];
                    a[q][rs] = (lane == ls) ? 0.0 : av;
                }
            }
        }
        XSTAMP(11);

        // ---- pivot column -> l = column / pivot, parked in LDS for everybody ----
        {
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (wave + XWAVES * j < RPT) ok &= ((cc[j].x ^ cc[j].y ^ cc[j].z ^ cc[j].w) == tag);
                if (__all(ok)) break;
                xcd_poll_again();
                if (++spins > XSPIN) {
                    atomicExch(reinterpret_cast<int*>(lds_ptrs[0]) + 1, 1);
                    if (lds_ptrs[1]) reinterpret_cast<volatile int*>(lds_ptrs[1])[5] = 1;
                    win_i[4] = 1; // observed by everybody after the next barrier (B), or after the loop
                    break;
                }
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (wave + XWAVES * j < RPT) cc[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, slot_off + j * XT * 16, 0, BUF_SC1);
            }
            XSTAMP(12);
            // x / p through the shared refined reciprocal (bitwise the IEEE quotient, see refined_rcp); zeros keep the sign
            // rule through x * rp; anything unusual takes the full division
            double lq[XR];
            bool slow = false;
#pragma unroll
            for (int j = 0; j < XR; ++j) {
                lq[j] = 0.0;
                if (wave + XWAVES * j < RPT) {
                    const double x = mk_f64(cc[j].x, cc[j].y);
                    const double q0 = x * rp;
                    const double qf = __builtin_fma(__builtin_fma(-wval, q0, x), rp, q0);
                    lq[j] = (x == 0.0) ? q0 : qf;
                    slow |= !(p_mid & (exp_mid(x) | (x == 0.0)));
                }
            }
            if (__ballot(slow) != 0ull) {
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (wave + XWAVES * j < RPT) {
                        const double x = mk_f64(cc[j].x, cc[j].y);
                        if (!(p_mid & (exp_mid(x) | (x == 0.0)))) lq[j] = x / wval;
                    }
            }
            // (the pivot row itself leaves the trailing block: its l is 0 like that of every row pivoted before, whose
            // emptied slots already read 0 in the published column; slot rows beyond M are zeros divided by the pivot)
#pragma unroll
            for (int j = 0; j < XR; ++j)
                if (wave + XWAVES * j < RPT) lbuf[lane * LSTR + wave + XWAVES * j] = (tid + XT * j == irow_p) ? 0.0 : lq[j];
        }
        XSTAMP(4);
        __syncthreads(); // (C)
        XSTAMP(13);
        xvec<RPT> l;
#pragma unroll
        for (int r = 0; r < RPT; ++r) l[r] = lbuf[lane * LSTR + r];
        XSTAMP(7);
        // =====================================================================================
        // elimination step kn: the pivot column keeps l (scale_column_tail, matrixlu.rs:562-577; rows that are already
        // pivoted get the zeros of their emptied slots), the trailing block gets the rank-1 update
        // (update_trailing_submatrix, matrixlu.rs:593-612) fused with the per-column maxima for the next arg-max
        // =====================================================================================
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int cq = __builtin_amdgcn_readfirstlane(cpos[q]);
            mq[q] = -1.0;
            if (cq > kn) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) { // in place (see sub_in_place); un-fused, one rounding per operation like the reference
                    double t = a[q][r];
                    sub_in_place(t, l[r] * u[q]);
                    a[q][r] = t;
                }
                double m0 = -1.0, m1 = -1.0;
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    if (r & 1) m1 = vmax_abs(m1, a[q][r]);
                    else m0 = vmax_abs(m0, a[q][r]);
                }
                mq[q] = vmax(m0, m1);
            }
            // (the pivot column itself, cq == kn, is left as it is: its registers keep the un-scaled column, and L = column /
            // pivot is formed by the same division when the factored matrix is written out — the column is never read again)
        }
        npiv = kn + 1;
        XSTAMP(0);
    }

    // ---- results ----
    const unsigned long long t_done = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
    if (win_i[4]) timed_out = true; // (a column wait of the last step that gave up: its flag was set before barrier (C))
    if (npiv >= (M < N ? M : N)) error = 0.0; // matrixlu.rs:811-813
    if (rank == 0 && tid == 0) {
        p.iresult[0] = npiv;
        p.dresult[0] = error; // tid 0 belongs to the polling wave, which keeps the error
    }
    if (stamp_on) {
        for (int e = 0; e < 16; ++e) p.stamps[e] = lds_stamps[e];
        p.stamps[16] = t_elected - t_entry; // fixed part of a launch: election ...
        p.stamps[17] = t_loop - t_elected;  // ... matrix load, tables, first maxima ...
        p.stamps[18] = t_done - t_loop;     // (the pivot steps)
    }
    if (timed_out) return;
    // permutations: the tables are stable since the last barrier; device block and host mirror are written side by side
    if (rank == 0) {
        int* const h_rp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.row_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        int* const h_cp = p.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(p.h_block) + (reinterpret_cast<const char*>(p.col_perm) - reinterpret_cast<const char*>(p.dresult))) : nullptr;
        for (int i = tid; i < M; i += XT) {
            const int v = posrow[i];
            p.row_perm[i] = v;
            if (h_rp) h_rp[i] = v;
        }
        for (int j = tid; j < N; j += XT) {
            const int v = poscol[j];
            p.col_perm[j] = v;
            if (h_cp) h_cp[j] = v;
        }
        unsigned long long* const h_pv = p.h_block ? p.h_block + (reinterpret_cast<const char*>(p.pivot_vals) - reinterpret_cast<const char*>(p.dresult)) / 8 : nullptr;
        for (int e = tid; e < npiv; e += XT) {
            const double v = lds_pivots[e];
            p.pivot_vals[e] = v;
            if (h_pv) h_pv[e] = (unsigned long long)__double_as_longlong(v);
        }
    }
    // factored matrix in permuted coordinates: rows of U come from the side buffer (this wave wrote them itself), the rest
    // (L below the diagonal, the untouched trailing block) from the registers
#pragma unroll
    for (int q = 0; q < CPT; ++q)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = lane + 64 * r;
            if (cpos[q] >= 0 && i < M) {
                const int cp = cpos[q], rp = rowpos[i];
                const bool from_u = (rp < npiv) && (cp >= rp);
                double v = a[q][r];
                const bool in_l = (cp < npiv) && (rp > cp);
                if (in_l) { // scale_column_tail (matrixlu.rs:562-577), deferred: the same division the step itself used
                    const double pv = lds_pivots[cp];
                    v = xcd_div(v, pv, refined_rcp(pv), exp_mid(pv));
                    if (v != v) nan_seen = 1;
                }
                if (p.Aout) {
                    if (from_u)
                        v = __longlong_as_double((long long)__hip_atomic_load(
                            reinterpret_cast<const unsigned long long*>(p.urows) + ((size_t)rp * N + (g + NW * q)), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT));
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
    // host-visible header (the pivot values went to the mirror with the permutations, the two flag words belong to their setters)
    if (p.h_block && rank == 0 && tid == 0) {
        p.h_block[0] = (unsigned long long)__double_as_longlong(error);
        p.h_block[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.dresult) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((volatile int*)p.h_block)[4] = npiv;
        // completion token: the host accepts the result only if rank 0 ran to its end in THIS launch (a launch whose
        // workgroups never met the elected XCD would otherwise leave an all-zero block behind)
        ((volatile int*)p.h_block)[7] = (int)p.salt;
        if (p.ts_u64 > 0) {
            p.h_block[p.ts_u64] = ts_begin;
            p.h_block[p.ts_u64 + 1] = wall_clock64();
        }
        reinterpret_cast<unsigned long long*>(p.dresult)[1] = 0ull; // clean header for the next launch (every agent's atomicMax is long done)
        if (p.dims) { // bond chain: the device-side completion token for the next preparation kernel (max |a| stays in the mirror)
            __threadfence();
            p.iresult[3] = (int)p.salt;
        }
        if (stamp_on) p.stamps[19] = __builtin_amdgcn_s_memtime() - t_done; // ... write-out and host mirror
    }
    // bond chain without per-launch host mirror: the device block is complete as it is (error, max |a|, rank, flags, pivot
    // values, permutations) and is copied to the host once, behind the whole chain; it only lacks the time stamps and the token
    if (!p.h_block && p.dims && rank == 0 && tid == 0) {
        if (p.ts_u64 > 0) {
            unsigned long long* const blk = reinterpret_cast<unsigned long long*>(p.dresult);
            blk[p.ts_u64] = ts_begin;
            blk[p.ts_u64 + 1] = wall_clock64();
        }
        __threadfence();
        p.iresult[3] = (int)p.salt;
    }
}

#ifndef T4A_XCD_GROUP_TU
template <int RPT, int CPT, bool ROWMAJOR>
__global__ void __launch_bounds__(XT) __attribute__((amdgpu_waves_per_eu(XWAVES / 4, XWAVES / 4))) rrlu_xcd_kernel(RrluXcdArgs p)
{
    rrlu_xcd_body<RPT, CPT, ROWMAJOR>(p);
}
template <int RPT, int CPT, bool ROWMAJOR> void xcd_launch_tie(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once; // (launches come from several host threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_xcd_kernel<RPT, CPT, ROWMAJOR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_xcd_kernel<RPT, CPT, ROWMAJOR>), dim3(plan.grid), dim3(XT), plan.lds_bytes, stream, a);
}

template <int RPT, int CPT> void xcd_launch_rc(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    if (a.tie_row_major) xcd_launch_tie<RPT, CPT, true>(plan, a, stream);
    else xcd_launch_tie<RPT, CPT, false>(plan, a, stream);
}

template <int RPT> void xcd_launch_r(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    switch (plan.CPT) {
    case 1: xcd_launch_rc<RPT, 1>(plan, a, stream); break;
    case 2: xcd_launch_rc<RPT, 2>(plan, a, stream); break;
    case 3: if constexpr (RPT * 3 <= XCD_MAX_VALUES) xcd_launch_rc<RPT, 3>(plan, a, stream); break;
    case 4: if constexpr (RPT * 4 <= XCD_MAX_VALUES) xcd_launch_rc<RPT, 4>(plan, a, stream); break;
    case 5: if constexpr (RPT * 5 <= XCD_MAX_VALUES && XCD_MAX_CPT >= 5) xcd_launch_rc<RPT, 5>(plan, a, stream); break;
    default: if constexpr (RPT * 6 <= XCD_MAX_VALUES && XCD_MAX_CPT >= 6) xcd_launch_rc<RPT, 6>(plan, a, stream); break;
    }
}

// instantiated row counts per lane (a plan rounds RPT up to the next one)
#ifdef T4A_XCD_DEV
constexpr int kRpts[] = {2, 12};
#else
constexpr int kRpts[] = {1, 2, 3, 4, 6, 8, 12, 16};
#endif
int xcd_norm_rpt(int r)
{
    for (int v : kRpts)
        if (r <= v) return v;
    return -1;
}
int xcd_norm_cpt(int c)
{
    return c <= XCD_MAX_CPT ? (c < 1 ? 1 : c) : -1;
}

#else
// Group launch: eight factorisations, one per XCD.  Every workgroup takes the argument block of the XCD it landed on (read
// through the kernel-argument segment: scalar loads like those of the by-value argument, no private copy of the array); an
// empty slot carries xcc = -1, its workgroups return at once.
template <int RPT, int CPT, bool ROWMAJOR>
__global__ void __launch_bounds__(XT) __attribute__((amdgpu_waves_per_eu(XWAVES / 4, XWAVES / 4))) rrlu_xcd_group_kernel(RrluXcdGroupArgs g)
{
    (void)g;
    const unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)xcc_id()) & 7u;
    rrlu_xcd_body<RPT, CPT, ROWMAJOR>(*reinterpret_cast<const RrluXcdArgs*>(kernarg_base() + (size_t)x * sizeof(RrluXcdArgs)));
}

template <int RPT, int CPT, bool ROWMAJOR> void xcd_group_launch_tie(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_xcd_group_kernel<RPT, CPT, ROWMAJOR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_xcd_group_kernel<RPT, CPT, ROWMAJOR>), dim3(plan.grid), dim3(XT), plan.lds_bytes, stream, a);
}
template <int RPT, int CPT> void xcd_group_launch_rc(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool row_major, hipStream_t stream)
{
    if (row_major) xcd_group_launch_tie<RPT, CPT, true>(plan, a, stream);
    else xcd_group_launch_tie<RPT, CPT, false>(plan, a, stream);
}
template <int RPT> void xcd_group_launch_r(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool row_major, hipStream_t stream)
{
    switch (plan.CPT) {
    case 1: xcd_group_launch_rc<RPT, 1>(plan, a, row_major, stream); break;
    case 2: xcd_group_launch_rc<RPT, 2>(plan, a, row_major, stream); break;
    case 3: if constexpr (RPT * 3 <= XCD_MAX_VALUES) xcd_group_launch_rc<RPT, 3>(plan, a, row_major, stream); break;
    default: if constexpr (RPT * 4 <= XCD_MAX_VALUES) xcd_group_launch_rc<RPT, 4>(plan, a, row_major, stream); break;
    }
}

#endif

} // namespace

#ifndef T4A_XCD_GROUP_TU
// Plans beyond one XCD's 1024 x 1024 (round 5, kernels_rrlu_xcd2m.hip): up to 1 536 rows (24 row slots per lane) and the columns over
// the agents of K <= 3 neighbouring XCDs.  Only the instantiations that translation unit compiles: fewest columns per agent first (the
// update pass is what every agent pays per step), then fewest XCDs (every XCD more is four more key loads per polling lane).
static bool xcd_make_big_plan(int M, int N, RrluXcdPlan* out)
{
    static const bool off = std::getenv("T4A_NO_XCD_BIG") != nullptr;
    if (off || M < 1 || N < 1 || M > 1536 || N > 1536) return false;
    const int rpt = M <= 1024 ? 16 : 24;
    struct Cand { int rpt, cpt, k; };
    static const Cand cands[] = {{24, 1, 1}, {24, 2, 1}, {24, 2, 2}, {16, 2, 3}, {24, 2, 3}, {16, 3, 2}}; // (sorted by cpt, then k)
    static const int k_env = std::getenv("T4A_XCD_K") ? std::atoi(std::getenv("T4A_XCD_K")) : 0;
    for (const Cand& c : cands) {
        if (c.rpt != rpt) continue;
        if (k_env > 0 && c.k != k_env) continue;
        int w = (N + XWAVES * c.cpt * c.k - 1) / (XWAVES * c.cpt * c.k);
        const int wq = c.k == 3 ? 8 : c.k == 2 ? 4 : 1; // several XCDs: 8 k w agents must be a multiple of 64 (the polling lanes' key loads carry no clamp)
        w = (w + wq - 1) / wq * wq;
        if (w > 32) continue;
        RrluXcdPlan plan;
        plan.W = w;
        plan.RPT = c.rpt;
        plan.CPT = c.cpt;
        plan.K = c.k;
        plan.grid = 8 * w;
        plan.lds_bytes = xcd2_lds_total(c.rpt, c.k);
        *out = plan;
        return true;
    }
    return false;
}

bool rrlu_xcd_make_plan(int M, int N, RrluXcdPlan* out, bool any_size, int max_w, bool allow_big)
{
    if (max_w < 1 || max_w > 32) max_w = 32;
    if (M < 1 || N < 1) return false;
    if (M > 1024 || N > 1024) return allow_big && max_w == 32 && xcd_make_big_plan(M, N, out);
    static const int min_elems = diag_env("T4A_XCD_MIN") ? std::atoi(diag_env("T4A_XCD_MIN")) : 64 * 64;
    if (!any_size && (long long)M * N <= (long long)min_elems) return false; // tiny matrices: the single-workgroup plan of the chip-wide kernel
    const int rpt = xcd_norm_rpt((M + 63) / 64);
    if (rpt < 0) return false;
    // columns per agent: as few as the 32 compute units of an XCD allow.  A step costs an agent ~500 cycles per owned column
    // (update, its share of the search, pivot-row extraction) and the gather is the same four key loads per lane for any
    // number of agents up to 256 (measured per step: 2.0 - 2.2 us with one column per agent, 2.45 us with two, 2.65 us with
    // three; T4A_XCD_COST=old restores the round-2 model that traded columns against 64-agent key groups)
    static const bool old_cost = diag_env("T4A_XCD_COST") != nullptr;
    static const int w_env = diag_env("T4A_XCD_W") ? std::atoi(diag_env("T4A_XCD_W")) : 0;
    static const int cpt_env = diag_env("T4A_XCD_CPT") ? std::atoi(diag_env("T4A_XCD_CPT")) : 0;
    int best_cpt = -1, best_w = 0;
    long best_cost = 0;
    for (int c = 1; c <= XCD_MAX_CPT; ++c) {
        const int cpt = xcd_norm_cpt(c);
        if (cpt != c) continue;
        if (cpt_env > 0 && cpt != cpt_env) continue;
        const int w = (N + XWAVES * cpt - 1) / (XWAVES * cpt);
        if (w > max_w) continue;
        if (rpt * cpt > XCD_MAX_VALUES) continue;
        const long cost = old_cost ? 24L * rpt * cpt + 250L * ((w * XWAVES + 63) / 64) : (long)cpt;
        if (best_cpt < 0 || cost < best_cost) {
            best_cpt = cpt;
            best_w = w;
            best_cost = cost;
        }
    }
    if (best_cpt < 0) return false;
    if (w_env > 0) {
        best_w = w_env > max_w ? max_w : w_env;
        int c = (N + XWAVES * best_w - 1) / (XWAVES * best_w);
        best_cpt = xcd_norm_cpt(c);
        if (best_cpt < 0 || rpt * best_cpt > XCD_MAX_VALUES) return false;
    }
    RrluXcdPlan plan;
    plan.W = best_w;
    plan.RPT = rpt;
    plan.CPT = best_cpt;
    plan.grid = 8 * best_w;
    // (no padding of the LDS request: two workgroups of this kernel cannot share a compute unit anyway — 8 waves of ~200 VGPRs
    // each — and a padded request keeps other kernels' workgroups that only have to RETURN on this XCD, see lu_update_kernel,
    // from being placed at all)
    plan.lds_bytes = xcd_lds_total(rpt);
    static const bool pad_lds = diag_env("T4A_XCD_PAD_LDS") != nullptr;
    if (pad_lds && plan.lds_bytes < 84 * 1024) plan.lds_bytes = 84 * 1024;
    *out = plan;
    return true;
}

size_t rrlu_xcd_keys_bytes(const RrluXcdPlan& plan) { return (size_t)4 * plan.K * plan.W * XWAVES * 16; } // early keys + full keys, two step parities each
size_t rrlu_xcd_cols_bytes(const RrluXcdPlan& plan, int)
{
    // slots are padded to 64 * RPT rows; T4A_XCD_CSTRIDE (experiment, with a library built with -DT4A_X2_CSTRIDE): sparse slots
    static const size_t cstride = diag_env("T4A_XCD_CSTRIDE") ? (size_t)std::atol(diag_env("T4A_XCD_CSTRIDE")) : 256;
    return (size_t)2 * plan.K * plan.W * XWAVES * (size_t)(4 * plan.RPT) * (cstride < 256 ? 256 : cstride) + (plan.K > 1 ? 256 + (size_t)2 * plan.K * plan.W * XWAVES * 16 : 0); // (+ the finalist granules of the XCDs and the write-through copies of the full keys)
}

void rrlu_xcd_launch(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    switch (plan.RPT) {
#ifdef T4A_XCD_DEV
    case 2: xcd_launch_r<2>(plan, a, stream); break;
    default: xcd_launch_r<12>(plan, a, stream); break;
#else
    case 1: xcd_launch_r<1>(plan, a, stream); break;
    case 2: xcd_launch_r<2>(plan, a, stream); break;
    case 3: xcd_launch_r<3>(plan, a, stream); break;
    case 4: xcd_launch_r<4>(plan, a, stream); break;
    case 6: xcd_launch_r<6>(plan, a, stream); break;
    case 8: xcd_launch_r<8>(plan, a, stream); break;
    case 12: xcd_launch_r<12>(plan, a, stream); break;
    default: xcd_launch_r<16>(plan, a, stream); break;
#endif
    }
}
#else
// (this half of the file is compiled as its own translation unit: kernels_rrlu_xcd_group.hip)
void rrlu_xcd_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool tie_row_major, hipStream_t stream)
{
    switch (plan.RPT) {
    case 1: xcd_group_launch_r<1>(plan, a, tie_row_major, stream); break;
    case 2: xcd_group_launch_r<2>(plan, a, tie_row_major, stream); break;
    case 3: xcd_group_launch_r<3>(plan, a, tie_row_major, stream); break;
    case 4: xcd_group_launch_r<4>(plan, a, tie_row_major, stream); break;
    case 6: xcd_group_launch_r<6>(plan, a, tie_row_major, stream); break;
    case 8: xcd_group_launch_r<8>(plan, a, tie_row_major, stream); break;
    case 12: xcd_group_launch_r<12>(plan, a, tie_row_major, stream); break;
    default: xcd_group_launch_r<16>(plan, a, tie_row_major, stream); break;
    }
}
#endif

} // namespace t4a
