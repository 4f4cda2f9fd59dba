// kernels_rrlu_xcd2.hip — K2 fast path, round 4: second generation of the single-XCD register-resident full-pivot rrLU.
//
// Same contract as kernels_rrlu_xcd.hip (bit-identical to rrlu_mut, tensor4all-core/src/matrixlu.rs:735-819; arg-max semantics
// matrixlu.rs:480-519; a right-orthogonal factorisation runs as the left-orthogonal one of A^T with the row-major tie order), same
// placement scheme (8 W workgroups launched, the W that land on the elected XCD take part), same mailbox layout and the same
// data layout (wave = agent that owns whole columns, lane = rows lane + 64 r).  What changed is the PROTOCOL of a pivot step: the
// first generation spent ~6 200 cycles per step of which only ~590 are arithmetic — the rest was three serial instruction streams
// (agents: search; polling wave: pick, stop tests, record; everybody: record decode, tables, pivot row, division).  Here:
//
//   * the record that crosses barrier (B) is ONE word: the winning agent.  Every wave fetches the winner's full key
//     {value, position key, row index, column slot} from the mailbox itself, together with its rows of the winner's column — the
//     polling wave neither waits for that key nor decodes it for the others;
//   * the stop tests (matrixlu.rs:757-781) left the critical path: the polling wave evaluates them AFTER barrier (B), while the
//     other waves divide the column, and publishes the verdict before barrier (C); a stopped step is dropped there, before the
//     elimination.  The permutation tables are written by the same wave behind its stop test (nothing to undo on a stop);
//   * the pivot row is never zeroed by hand: l of the pivot row is pivot / pivot = 1.0 exactly, so the rank-1 update itself
//     leaves exact zeros in the trailing columns (x - 1.0 * x), which is all the self-masking of the first generation needed;
//   * column positions are not carried in registers: an `active` bit per owned column says whether it is still in the trailing
//     block, its position comes from an LDS table when a position key is built (normal path: one look-up per step);
//   * the polling wave owns columns like everybody else but takes no share of the division (seven waves divide);
//   * NON-FINITE values are not handled here at all: the launch gives up with code 2 (iresult[1]) and the caller runs the
//     first-generation kernel, which implements the NaN-incumbent rule.  This is exact, not heuristic: a NaN can only appear
//     in a trailing block after an infinity has (|l| <= 1 under full pivoting, so l * u and a - l * u overflow before anything
//     becomes NaN), an infinity in the trailing block is some agent's candidate magnitude, and the polling wave sees every
//     candidate magnitude of every step; NaN / infinity in the INPUT is found while the matrix is loaded and travels in the
//     first step's keys.  Ties, zeros and subnormal scores are resolved exactly as before (exact sweeps on (v*v, position)).
#include "kernels_rrlu_xcd_common.hpp"

namespace t4a {

namespace {

// LDS layout of one workgroup (compile-time offsets; the tables are sized for the largest matrix of the plan family)
template <int RPT, int KX = 1> struct Xcd2Lds {
    static constexpr int LSTR = xcd_lstr(RPT);
    static constexpr int TBL = xcd2_tbl(RPT, KX);            // entries of the position tables / pivot values (matrices beyond 1024 rows: round 5)
    static constexpr int o_l = 0;                        // double [64][LSTR]: l of row lane + 64 r at lane * LSTR + r
    static constexpr int o_wd = o_l + 64 * LSTR * 8;     // (16 bytes, unused)
    static constexpr int o_wi = o_wd + 16;               // int [16]: [0] stop verdict [1] give-up (any wave) [3] rank [4..7] record of the step: winning agent | give-up << 30, winner's value lo / hi, meta; [8..11], [12..15] (agents on several XCDs): what the polling waves of XCD 1 / 2 found
    static constexpr int o_pp = o_wi + 64;               // u64 [2]: iresult / h_block pointers for the give-up paths
    static constexpr int o_st = o_pp + 16;               // u64 [16] phase stamps (diagnostic builds)
    static constexpr int o_pv = o_st + 128;              // double [TBL] pivot values of this launch
    static constexpr int o_pr = o_pv + TBL * 8;          // u16 [TBL] position -> row index
    static constexpr int o_rp = o_pr + TBL * 2;          // u16 [TBL] row index -> position
    static constexpr int o_pc = o_rp + TBL * 2;          // u16 [TBL] position -> column index
    static constexpr int o_cp = o_pc + TBL * 2;          // u16 [TBL] column index -> position
    static constexpr int bytes = o_cp + TBL * 2;
};
static_assert(Xcd2Lds<12>::bytes == (int)xcd_lds_total(12), "plan.lds_bytes must cover the layout");
static_assert(Xcd2Lds<24>::bytes == (int)xcd2_lds_total(24), "plan.lds_bytes must cover the layout (plans beyond 1024 rows)");
static_assert(Xcd2Lds<16, 3>::bytes == (int)xcd2_lds_total(16, 3), "plan.lds_bytes must cover the layout (plans beyond 1024 columns)");

#ifndef T4A_XCD_STAMP_WAVE
#define T4A_XCD_STAMP_WAVE 0
#endif
// T4A_X2_POLLEARLY = 1 (experiment): the polling wave issues its sweep of the early keys in the middle of its own position search
// (behind the ballots, in front of the slot sweep) instead of behind its full-key store
// T4A_X2_CSTRIDE (experiment): byte distance between consecutive 256-byte chunks (16 rows) of a column slot in the mailbox; 256 =
// contiguous.  A larger stride spreads a column over more L2 channels if the channel interleave is coarser than 256 bytes (the
// host sizes the mailbox with T4A_XCD_CSTRIDE set to the same value).
#ifndef T4A_X2_CSTRIDE
#define T4A_X2_CSTRIDE 256
#endif
// T4A_X2_DUPLOAD = 1 (measurement only): every dividing wave fetches its rows of the winner's column TWICE — if the step gets slower by
// about the time the hand-off takes, the hand-off is bound by the L2 serving 29 workgroups the same lines, not by latency
// T4A_X2_NOBARB = 1 (experiment): barrier (B) is replaced by a step tag inside the record: the polling wave's
// lane 0 stores the record with the tag, every other wave spins on the record in the LDS.  (Safe: a record of step kn exists only after
// every agent's early key of step kn, i.e. after every wave of this workgroup has finished reading the previous step's l.)
#ifndef T4A_X2_NOBARB
#define T4A_X2_NOBARB 0
#endif
#ifndef T4A_X2_DUPLOAD
#define T4A_X2_DUPLOAD 0
#endif
#ifndef T4A_X2_POLLEARLY
#define T4A_X2_POLLEARLY 0
#endif
// T4A_X2_KHEARLY = 1: the polling wave requests the FULL keys together with every sweep of the early keys instead of behind the
// successful one — when the winner's position search has finished by then (its full key carries the right tag) the pick does not wait
// for a second L2 round trip; a stale one is fetched again as before
#ifndef T4A_X2_KHEARLY
#define T4A_X2_KHEARLY 0
#endif
// T4A_X2_FSTAGGER = 1 (agents on several XCDs): the remote finalists are requested THREE times, staggered — when the early keys are
// complete (as before), when the local winner is known, and behind the publication of the own finalist — and examined oldest first.
// A read across the fabric takes ~2 400 cycles and samples the memory side somewhere in the middle: a single read issued before the
// remote XCDs have picked comes back stale and the next one costs another full trip.
#ifndef T4A_X2_FSTAGGER
#define T4A_X2_FSTAGGER 1
#endif

// full-key meta word (second generation): bits 0..10 row index of the candidate (up to 1536 rows since round 5), 11..12 column slot
// of the publishing agent, bit 13 the agent has a candidate.  Positions are NOT carried: whoever needs one reads the LDS tables (the
// polling wave behind its stop test; the exact comparison of the rare paths).  Position keys of the exact paths: 11 + 11 bits.
constexpr unsigned X2_META_VALID = 1u << 13;
constexpr unsigned X2_ROW_MASK = 2047u;
constexpr int X2_QSHIFT = 11, X2_PSHIFT = 11;

constexpr int X2_DIVW = XWAVES - 1; // waves that divide the pivot column (all but the polling wave)

// KX: XCDs the agents live on (round 5).  1: everything inside one XCD's L2 (plain stores, sc1 loads).  > 1: the agents of KX
// neighbouring XCDs (p.xcc, p.xcc + 1, ... mod 8) — matrices beyond the registers of one XCD, BASELINE.json configs[3]: 1 450 x 1 450
// with the history extras — exchange through the same mailbox with write-through (sc1) stores; the L2s are kept coherent for such
// lines by the fabric, but a hop across it costs 3 - 4 times the intra-XCD one (measured: ~4 000 cycles from a remote agent's key
// store to a polling wave seeing it when all 8 KX W keys cross).  So the arg-max runs in TWO LEVELS: every workgroup's polling wave
// gathers only the early keys of ITS OWN XCD (an exchange inside one L2, exactly the single-XCD protocol) and names that XCD's
// finalist; ONE 16-byte finalist granule per XCD crosses the fabric (published by the workgroup of local rank 0), and every polling
// wave compares the KX finalists.  The winner's column crosses as before (write-through stores, speculative publication).
template <int RPT, int CPT, bool ROWMAJOR, int KX = 1>
__device__ __forceinline__ void rrlu_xcd2_body(const RrluXcdArgs& p, const RrluXcdArgs* pk)
{
    static_assert(XWAVES == 8, "the second-generation kernel is written for eight waves per workgroup");
    static_assert(KX >= 1 && KX <= 3, "instantiated for up to three XCDs");
    constexpr int ST_AUX = KX > 1 ? BUF_SC1 : 0;   // mailbox stores: write-through when other XCDs read them
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = Xcd2Lds<RPT, KX>;
    constexpr int LSTR = L::LSTR;
    constexpr int MP = 64 * RPT; // rows of a published column slot (rows beyond M carry zeros)
    double* const lbuf = reinterpret_cast<double*>(smem_raw + L::o_l);
    int* const ctl = reinterpret_cast<int*>(smem_raw + L::o_wi);
    unsigned long long* const lds_ptrs = reinterpret_cast<unsigned long long*>(smem_raw + L::o_pp);
    unsigned long long* const lds_stamps = reinterpret_cast<unsigned long long*>(smem_raw + L::o_st);
    double* const lds_pivots = reinterpret_cast<double*>(smem_raw + L::o_pv);
    unsigned short* const posrow = reinterpret_cast<unsigned short*>(smem_raw + L::o_pr);
    unsigned short* const rowpos = reinterpret_cast<unsigned short*>(smem_raw + L::o_rp);
    unsigned short* const poscol = reinterpret_cast<unsigned short*>(smem_raw + L::o_pc);
    unsigned short* const colpos = reinterpret_cast<unsigned short*>(smem_raw + L::o_cp);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t_entry = (kXcdStamps && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- election: only the workgroups that landed on the wanted XCD take part ----
    if (tid == 0) {
        int rank = -1;
        if constexpr (KX == 1) {
            if ((int)xcc_id() == p.xcc) {
                const unsigned t = atomicAdd(p.ticket, 1u) - p.ticket_base;
                if (t < (unsigned)p.W) rank = (int)t;
            }
        } else {
            // every workgroup of the launch takes a ticket from the counter of the XCD it landed on (8 counters, each advances by
            // grid / 8 per launch); those on the KX elected XCDs become ranks xi W + ticket
            const unsigned x = xcc_id() & 7u;
            const unsigned t = atomicAdd(p.ticket + x, 1u) - p.ticket_base;
            const unsigned xi = (x - (unsigned)p.xcc) & 7u;
            if (xi < (unsigned)KX && t < (unsigned)p.W) rank = (int)(xi * (unsigned)p.W + t);
        }
        ctl[3] = rank;
        ctl[0] = 0;
        ctl[1] = 0;
        ctl[2] = 0;
        for (int e = 8; e < 16; ++e) ctl[e] = 0;
        lds_ptrs[0] = (unsigned long long)p.iresult;
        lds_ptrs[1] = (unsigned long long)p.h_block;
        for (int e = 0; e < 16; ++e) lds_stamps[e] = 0ull;
    }
    __syncthreads();
    const int rank = __builtin_amdgcn_readfirstlane(ctl[3]);
    if (rank < 0) {
        // pass-through workgroup (another XCD).  Bond chain: it evaluates its share of the NEXT bond's candidate matrix first
        if (p.spec.out && p.dims) {
            const int m_spec = p.dims[2] != 0 ? 0 : (p.dims_swap ? p.dims[1] : p.dims[0]);
            if (m_spec > 0 && m_spec <= p.M)
                xcd_spec_work(reinterpret_cast<const XcdSpecArgs*>(kernarg_base() + offsetof(RrluXcdArgs, spec)), m_spec, ctl + 8);
        }
        return;
    }
    const unsigned long long ts_begin = p.ts_u64 > 0 ? wall_clock64() : 0ull;
    const unsigned long long t_elected = (kXcdStamps && p.stamps) ? __builtin_amdgcn_s_memtime() : 0ull;
    const int NWL = p.W * XWAVES;  // agents of one XCD: what one polling wave gathers
    const int NW = KX * NWL;
    const int g = rank * XWAVES + wave; // agent id
    const bool poller = wave == 0;     // the polling wave: gathers the early keys of ITS XCD's 8 W agents
    const int xi = KX > 1 ? rank / p.W : 0; // which of the KX XCDs this workgroup sits on
    const int pbase = xi * NWL;
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
    unsigned active = 0u; // bit q: column g + NW q exists and is still in the trailing block (wave-uniform)
#pragma unroll
    for (int q = 0; q < CPT; ++q)
        if (g + NW * q < N) active |= 1u << q;
    XSlab<RPT> a[CPT]; // ext vectors (two per column beyond 16 row slots): the run-time row-slot accesses become s_set_gpr_idx moves
    double local_sqmax = 0.0;
    bool bad = false; // a NaN or an infinity among my entries of the input
    // every load is issued before the first one is consumed (clamped addresses instead of branches): the whole matrix is
    // one round trip to memory per lane, not RPT * CPT dependent ones
    // (Row validity travels as a per-lane BIT MASK and is tested again for every column, behind a compiler barrier: as RPT x CPT
    // lane masks in scalar register pairs — which is what common-subexpression elimination makes of `i < M` — the validity of the
    // slab is what spilled the scalar register file of the wide instantiations.)
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
        const bool cok = g + NW * q < N; // (wave-uniform)
        const double* const colp = p.A + (cok ? (size_t)(g + NW * q) * lda : (size_t)0);
        const unsigned rm = (unsigned)opaque_v((int)rowmask);
#pragma unroll
        for (int r = 0; r < RPT; ++r) a[q].set(r, colp[(cok && ((rm >> r) & 1u)) ? srow[r] : 0]);
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const bool cok = g + NW * q < N;
        const unsigned rm = (unsigned)opaque_v((int)rowmask);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool ok = cok && ((rm >> r) & 1u);
            const double v = ok ? a[q].get(r) : 0.0;
            const double sqv = v * v; // max sqrt(v*v) == sqrt(max v*v): one square root per lane below
            if (sqv > local_sqmax) local_sqmax = sqv; // (NaN never enters, like the branchy form)
            bad |= !((v - v) == 0.0);                 // inf - inf and NaN - NaN are NaN
            a[q].set(r, v);
        }
    }
    for (int i = tid; i < M; i += XT) {
        posrow[i] = (unsigned short)i;
        rowpos[i] = (unsigned short)i;
    }
    for (int j = tid; j < N; j += XT) {
        poscol[j] = (unsigned short)j;
        colpos[j] = (unsigned short)j;
    }
    for (int e = tid; e < 64 * LSTR; e += XT) lbuf[e] = 0.0;
    {
        const double wm = wave_max_f64(sqrt(local_sqmax));
        if (lane == 0 && wm > 0.0)
            atomicMax((unsigned long long*)&p.dresult[1], (unsigned long long)__double_as_longlong(wm));
    }
    const unsigned wave_bad = __ballot(bad) != 0ull ? 1u : 0u; // travels in the z word of this agent's early keys
    __syncthreads();

    const bool stamp_on = kXcdStamps && (p.stamps != nullptr) && rank == 0 && tid == 64 * T4A_XCD_STAMP_WAVE;
    unsigned long long stamp_last = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_loop = stamp_last;

    // one mailbox: [2][NW] early keys (candidate magnitude only), [2][NW] full keys, then [2][NW][MP] column rows (one buffer
    // resource for every store and load of the exchange)
    const unsigned k2_base = 2u * (unsigned)NW * 16u;
    const unsigned cols_base = 4u * (unsigned)NW * 16u;
    const unsigned fin_base = cols_base + 2u * (unsigned)NW * (unsigned)(MP / 16) * (unsigned)T4A_X2_CSTRIDE; // [2][KX] finalist granules (agents on several XCDs)
    const unsigned k3_base = fin_base + 256u;  // [2][NW] write-through copies of the full keys (agents on several XCDs: the exact walk)
    const __amdgpu_buffer_rsrc_t mail =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.keys, 0, (int)(k3_base + (KX > 1 ? 2u * (unsigned)NW * 16u : 0u)), 0x00020000);

    int npiv = 0;
    double max_error = 0.0;             // kept by the polling waves
    double error = __builtin_nan("");
    bool timed_out = false;
    const double min_pivot_abs = (p.rel_tol == 0.0 && p.abs_tol == 0.0) ? 0.0 : 2.220446049250313e-16;
    double u[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) u[q] = 0.0;
    double prev_sq = __builtin_huge_val(); // nobody speculates on the first step
    // launch constants of the step loop in VECTOR registers (left in the kernel arguments they are re-read in every step)
    double spec_frac = p.spec_frac, rel_tol_v = p.rel_tol, abs_tol_v = p.abs_tol;
    asm volatile("" : "+v"(spec_frac), "+v"(rel_tol_v), "+v"(abs_tol_v));
    constexpr unsigned XSPIN = 1u << 20; // bounded spins: a hand-off that does not arrive makes the launch give up

    // maxima of the untouched matrix for the first arg-max
    double mq[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        mq[q] = -1.0;
        if (active & (1u << q)) {
            double m0 = -1.0, m1 = -1.0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (r & 1) m1 = vmax_abs(m1, a[q].get(r));
                else m0 = vmax_abs(m0, a[q].get(r));
            }
            mq[q] = vmax(m0, m1);
        }
    }
    // the division of the pivot column is shared by waves 1 .. 7: wave w takes slot rows (w - 1) + 7 j
    constexpr int XR = (RPT + X2_DIVW - 1) / X2_DIVW;
    const int sr0 = wave - 1; // first slot row of this wave (-1: the polling wave divides nothing)

    for (int kn = 0; kn < max_steps; ++kn) {
        const int k = kn - 1; // rows / columns at positions > k form the trailing block searched for pivot kn
        const int par = kn & 1;
        const unsigned tag = (p.salt << 16) | (unsigned)(kn + 1);
        // ---- wave arg-max: (max score, smallest position among the maxima, value there), all wave-uniform ----
        double m = mq[0];
#pragma unroll
        for (int q = 1; q < CPT; ++q) m = vmax(m, mq[q]);
        // (m >= 0, or -1 without a column in the trailing block: the high words order like the values)
        const int whi = wave_max_i32((int)hi32(m));
        const unsigned long long whb = __ballot((int)hi32(m) == whi);
        const bool hi_single = __builtin_popcountll(whb) == 1; // one lane holds the largest high word: it holds the maximum
        const double wmax = hi_single ? readlane_f64(m, (int)__builtin_ctzll(whb)) : wave_max_f64(m);
        const double sq = wmax * wmax; // the winning score v*v of this agent
        // ---- early key: the magnitude of the candidate goes out before its position is known.  In the normal case (one
        // agent holds the largest |v|, its square a normal number) the magnitudes alone decide the winner.  (No candidate: 0,
        // which sends the pick to the exact path.)  z: this agent met a non-finite input entry (read in the first step only).
        const int kslot = (par * NW + g) * 16;
        {
            const double k1 = (wmax >= 0.0) ? wmax : 0.0;
            u32x4 kv;
            kv.x = lo32(k1);
            kv.y = hi32(k1);
            kv.z = wave_bad;
            kv.w = tag ^ kv.x ^ kv.y ^ kv.z;
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(kv, mail, kslot, 0, 0); // (read inside this XCD only: a plain store, also when the agents span several XCDs)
        }
        u32x4 kg[4], kh[4];
        bool kg_issued = false;
        bool has_cand = false;         // this agent has a candidate
        double cval = 0.0;             // its value
        int cirow = 0, qstar = 0;      // its row index and my column slot
        if (wmax >= 0.0) {
            bool done = false;
            // while v*v is a normal number, distinct |v| have distinct squares, so the equality sweep can compare |v| itself
            // (and the rows that are already pivoted hold exact zeros in every active column, which cannot match)
            if (hi_mid(whi)) {
                unsigned long long bq[CPT];
                int nhit = 0;
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    bq[q] = __ballot(mq[q] == wmax); // (columns outside the trailing block keep mq = -1)
                    nhit += __builtin_popcountll(bq[q]);
                }
#if T4A_X2_POLLEARLY
                if (poller) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        kg[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
                    kg_issued = true;
                }
#endif
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
                                asm("v_cmp_eq_f64 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"((double)a[q].get(r)), "s"(wmax) : "vcc");
                            const unsigned hb_ = (unsigned)__builtin_amdgcn_readlane((int)bits, hl);
                            if (__builtin_popcount(hb_) == 1) {
                                const int rstar = RPT - 1 - (int)__builtin_ctz(hb_);
                                cirow = hl + 64 * rstar;
                                has_cand = true;
                                cval = readlane_f64(a[q].dyn(rstar), hl);
                                qstar = q;
                                done = true;
                            }
                        }
                }
            }
            if (!done) {
                // ties, zero / subnormal scores: exact sweep on the squares (an infinite score ends the launch at the pick)
                unsigned mypos = XNOPOS;
                double myval = 0.0;
                int myrow = 0, myq = 0;
                const int lane_o = opaque_v(lane), M_o = opaque_s(M); // (nothing of this rare path is hoisted out of the step loop)
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const bool qhit = (mq[q] >= 0.0) & (mq[q] * mq[q] == sq);
                    if (__ballot(qhit) != 0ull) {
                        const unsigned cp_ = colpos[opaque_s(g + NW * q)];
                        auto sweep = [&](int r, double av) {
                            const int i = lane_o + 64 * r;
                            const unsigned rp_ = rowpos[i < M_o ? i : 0];
                            const unsigned key = ROWMAJOR ? ((rp_ << X2_PSHIFT) | cp_) : ((cp_ << X2_PSHIFT) | rp_);
                            const double sc = av * av;
                            const bool hit = qhit & (i < M_o) & ((int)rp_ > k) & (sc == sq);
                            if (hit && key < mypos) {
                                mypos = key;
                                myval = av;
                                myrow = i;
                                myq = q;
                            }
                        };
                        if constexpr (RPT <= 16) {
#pragma unroll
                            for (int r = 0; r < RPT; ++r) sweep(r, a[q].get(r));
                        } else {
                            // (24 slots unrolled keep two dozen table reads in flight beside a 96-register slab: this rare path then
                            // decides the register allocation of the whole step loop — a rolled loop with run-time slot access instead)
#pragma unroll 1
                            for (int r = 0; r < RPT; ++r) sweep(r, a[q].dyn(opaque_s(r)));
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
                }
            }
        }
        XSTAMP(1);
        // ---- full key: value, position, row index, column slot (all fields are wave-uniform) ----
        {
            const unsigned meta = (unsigned)cirow | ((unsigned)qstar << X2_QSHIFT) | (has_cand ? X2_META_VALID : 0u);
            u32x4 kv;
            kv.x = lo32(cval);
            kv.y = hi32(cval);
            kv.z = meta;
            kv.w = tag ^ kv.x ^ kv.y ^ meta;
            if (lane == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(kv, mail, (int)k2_base + kslot, 0, 0);
                if constexpr (KX > 1) __builtin_amdgcn_raw_buffer_store_b128(kv, mail, (int)k3_base + kslot, 0, ST_AUX); // (a write-through copy for the exact walk over all XCDs' full keys)
            }
        }
        // the polling wave sweeps the early keys now — they left their agents a whole position search ago, so this first sweep
        // normally finds them all.  Every lane fetches four keys; lanes beyond NW re-read the last key (a valid duplicate), so
        // neither the arrival check nor the maximum needs a mask or a count of live groups.
        if (poller && !kg_issued) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kg[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
#if T4A_X2_KHEARLY
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
#endif
        }
        // thresholded speculative publication of the candidate column: pivots shrink slowly, so the next winner is almost
        // always an agent whose candidate is close to the previous pivot; its column is then already in the L2 when the
        // keys have been gathered
        const bool early_pub = !poller && has_cand && (sq >= spec_frac * prev_sq); // (a polling wave never stores a column early: those stores would sit in front of its key loads)
        constexpr int CS = T4A_X2_CSTRIDE, SLOTB = (MP / 16) * CS; // chunk stride / bytes of a column slot
        const int myslot = (int)cols_base + (par * NW + g) * SLOTB + (lane >> 4) * CS + (lane & 15) * 16; // byte offset of my row `lane` in the mailbox
        if (early_pub) {
            if (qstar == 0) xcd_publish_column<0, RPT, ST_AUX>(a[0], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 1) if (qstar == 1) xcd_publish_column<1, RPT, ST_AUX>(a[1], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 2) if (qstar == 2) xcd_publish_column<2, RPT, ST_AUX>(a[2], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 3) if (qstar == 3) xcd_publish_column<3, RPT, ST_AUX>(a[3], mail, myslot, tag, 4 * CS);
        }
        XSTAMP(2);

        // ---- the polling wave(s) gather the early keys and name the winner (matrixlu.rs:480-519 across agents) ----
        if (poller) {
            unsigned spins = 0;
            int giveup = 0; // 1: a hand-off did not arrive  2: non-finite values (the caller runs the first-generation kernel)
            XSTAMP(6);
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) ok &= ((kg[j].x ^ kg[j].y ^ kg[j].z ^ kg[j].w) == tag);
                if (__all(ok)) break;
                xcd_poll_again();
                if (++spins > XSPIN) {
                    giveup = 1;
                    break;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    kg[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
#if T4A_X2_KHEARLY
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
#endif
            }
            if (stamp_on) lds_stamps[5] += spins;
            // agents on several XCDs: the first read of the REMOTE finalists goes out now — a read across the fabric takes ~2 400 cycles,
            // and the remote XCDs publish theirs about a pick (~800 cycles) from now: it reaches the memory side after they have
            u32x4 fr;
            int xr = 0, fslot = 0;
            if constexpr (KX > 1) {
                fslot = (int)fin_base + (par * KX) * 16;
                xr = (lane < KX && lane != xi) ? lane : (xi == 0 ? 1 : 0); // lane x reads XCD x's finalist (the others a remote duplicate)
                fr = __builtin_amdgcn_raw_buffer_load_b128(mail, fslot + xr * 16, 0, BUF_SC1);
            }
            // the full keys: fetched now, in flight while the early ones are examined (a late one is fetched again below)
#if !T4A_X2_KHEARLY
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
#endif
            XSTAMP(8);
            int wa_ = 0;
            unsigned wkx = 0u, wky = 0u, wkz = 0u; // the winner's full key
            if (!giveup) {
                if (kn == 0) { // non-finite entries in the input: met by their owners while the matrix was loaded
                    unsigned zf = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) zf |= kg[j].z;
                    if (__ballot(zf != 0u) != 0ull) giveup = 2;
                }
                // winner over all agents.  Normal case: the largest candidate magnitude, its square a normal number (distinct
                // |v| <=> distinct scores), held by exactly one early key: one maximum reduction decides.  (A duplicate of the
                // last key can only push the count above one: then the exact path decides.  No candidate travels as 0.)
                bool decided = false;
                // first on the high words alone (early keys are magnitudes: non-negative): one integer reduction; the 64-bit
                // comparison only when several keys share the largest high word
                int lh = (int)kg[0].y;
#pragma unroll
                for (int j = 1; j < 4; ++j) lh = (int)kg[j].y > lh ? (int)kg[j].y : lh;
                const int ghi = wave_max_i32(lh);
                unsigned long long hb[4];
                int nh = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hb[j] = __ballot((int)kg[j].y == ghi);
                    nh += __builtin_popcountll(hb[j]);
                }
                double gm = 0.0, gsq = 0.0;
                bool in_range = hi_mid(ghi);
                if (!(in_range && nh == 1)) {
                    double lm = -1.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lm = vmax(lm, mk_f64(kg[j].x, kg[j].y));
                    gm = wave_max_f64(lm);
                    gsq = gm * gm;
                    in_range = (gsq >= 2.2250738585072014e-308) && (gsq < __builtin_huge_val());
                    nh = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hb[j] = __ballot(mk_f64(kg[j].x, kg[j].y) == gm);
                        nh += __builtin_popcountll(hb[j]);
                    }
                }
                if (in_range) {
                    if (nh == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (hb[j] != 0ull) {
                                const int hl = (int)__builtin_ctzll(hb[j]);
                                wa_ = hl + 64 * j;
                                // the winner's full key: normally long there; otherwise fetched again until it is
                                for (;;) {
                                    wkx = (unsigned)__builtin_amdgcn_readlane((int)kh[j].x, hl);
                                    wky = (unsigned)__builtin_amdgcn_readlane((int)kh[j].y, hl);
                                    wkz = (unsigned)__builtin_amdgcn_readlane((int)kh[j].z, hl);
                                    const unsigned kw = (unsigned)__builtin_amdgcn_readlane((int)kh[j].w, hl);
                                    if ((wkx ^ wky ^ wkz ^ kw) == tag) break;
                                    xcd_poll_again();
                                    if (++spins > XSPIN) {
                                        giveup = 1;
                                        break;
                                    }
                                    kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
                                }
                            }
                        decided = true;
                    }
                } else if (!(gsq < __builtin_huge_val())) {
                    giveup = 2; // an infinite score: overflow in the trailing block (or an infinite input)
                }
                (void)gm;
                XSTAMP(14);
                if constexpr (KX > 1) {
                    if (!decided && !giveup) {
                        wkz = 0x80000000u; // (wave 0 walks all full keys: ties across XCDs need every agent, not only this XCD's)
                        decided = true;
                    }
                }
                if (!decided && !giveup) {
                    // ties between agents, zero / subnormal scores: exact comparison of (v*v, position key) over the FULL keys;
                    // an agent without candidate carries value 0 and the largest position key
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ok &= ((kh[j].x ^ kh[j].y ^ kh[j].z ^ kh[j].w) == tag);
                        if (__all(ok)) break;
                        xcd_poll_again();
                        if (++spins > XSPIN) {
                            giveup = 1;
                            break;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            kh[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k2_base + (par * NW + pbase + min(lane + 64 * j, NWL - 1)) * 16, 0, BUF_SC1);
                    }
                    double csc = -1.0;
                    unsigned cpk = XNOPOS;
                    int cag = 0;
                    u32x4 ckey = kh[0];
                    const int lane_o = opaque_v(lane);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ag = lane_o + 64 * j;
                        // position key of the candidate from the tables of this workgroup (every workgroup keeps the same ones)
                        const unsigned rp_ = rowpos[kh[j].z & X2_ROW_MASK], cp_ = colpos[min(pbase + ag + NW * (int)((kh[j].z >> X2_QSHIFT) & 3u), N - 1)];
                        const unsigned pkey = ROWMAJOR ? ((rp_ << X2_PSHIFT) | cp_) : ((cp_ << X2_PSHIFT) | rp_);
                        const unsigned pk = ((ag < NWL) && (kh[j].z & X2_META_VALID)) ? pkey : XNOPOS;
                        const double v = mk_f64(kh[j].x, kh[j].y);
                        double sc = v * v;
                        sc = (ag < NWL) ? sc : -2.0;
                        const bool better = (sc > csc) | ((sc == csc) & (pk < cpk));
                        csc = better ? sc : csc;
                        cpk = better ? pk : cpk;
                        cag = better ? ag : cag;
                        ckey.x = better ? kh[j].x : ckey.x;
                        ckey.y = better ? kh[j].y : ckey.y;
                        ckey.z = better ? kh[j].z : ckey.z;
                    }
                    const double gmax = wave_max_f64(csc);
                    const unsigned gpos = wave_min_u32((csc == gmax) ? cpk : XNOPOS);
                    const unsigned long long sel = __ballot((csc == gmax) & (cpk == gpos));
                    const int wl = sel ? (int)__builtin_ctzll(sel) : 0;
                    wa_ = __builtin_amdgcn_readlane(cag, wl);
                    wkx = (unsigned)__builtin_amdgcn_readlane((int)ckey.x, wl);
                    wky = (unsigned)__builtin_amdgcn_readlane((int)ckey.y, wl);
                    wkz = (unsigned)__builtin_amdgcn_readlane((int)ckey.z, wl);
                }
            }
            wa_ += pbase; // (agent numbers are global from here on)
            if constexpr (KX > 1) {
                {
                    // ---- the finalists of the KX XCDs.  Every workgroup of an XCD has just named the SAME local winner from that XCD's
                    // 8 W early keys (an exchange inside one L2); only that one key per XCD crosses the fabric: the workgroup of local
                    // rank 0 publishes it (write-through), every polling wave reads the KX - 1 remote ones.  (Round 5, first version: every
                    // polling wave gathered all 8 KX W keys itself — 4 700 cycles per step waiting for remote keys, 1.2 MB of polling
                    // reads per round through the fabric; profiles/r05_xcd2m_phase_stamps.txt.)
                    // finalist granule: {value lo, value hi, meta | agent << 14 | give-up << 26 | undecided << 31, tag ^ fold}
                    bool need_exact = (wkz >> 31) != 0u; // (this XCD could not decide on its early keys alone)
#if T4A_X2_FSTAGGER
                    u32x4 fr1 = __builtin_amdgcn_raw_buffer_load_b128(mail, fslot + xr * 16, 0, BUF_SC1);
#endif
                    if (rank == xi * p.W && lane == 0) {
                        u32x4 fv;
                        fv.x = wkx;
                        fv.y = wky;
                        fv.z = (wkz & 0x80003FFFu) | ((unsigned)wa_ << 14) | ((unsigned)giveup << 26);
                        fv.w = tag ^ fv.x ^ fv.y ^ fv.z;
                        // (two 64-bit atomic exchanges at agent scope instead of a write-through store: an atomic is performed at the
                        // memory side at once, a store may sit in the write path for a while; a reader that sees one half old fails the
                        // tag check and polls again)
                        unsigned long long* const fp = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.keys) + fslot + xi * 16);
                        (void)__hip_atomic_exchange(fp, ((unsigned long long)fv.y << 32) | fv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        (void)__hip_atomic_exchange(fp + 1, ((unsigned long long)fv.w << 32) | fv.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    double best = __builtin_fabs(mk_f64(wkx, wky));
                    {
                        unsigned sp2 = 0;
#if T4A_X2_FSTAGGER
                        u32x4 fr2 = __builtin_amdgcn_raw_buffer_load_b128(mail, fslot + xr * 16, 0, BUF_SC1);
                        if (!__all((fr.x ^ fr.y ^ fr.z ^ fr.w) == tag)) {
                            fr = fr1;
                            if (!__all((fr.x ^ fr.y ^ fr.z ^ fr.w) == tag)) fr = fr2;
                        }
#endif
                        for (;;) {
                            if (__all((fr.x ^ fr.y ^ fr.z ^ fr.w) == tag)) break;
                            xcd_poll_again();
                            if (++sp2 > XSPIN) {
                                giveup = giveup ? giveup : 1;
                                break;
                            }
                            fr = __builtin_amdgcn_raw_buffer_load_b128(mail, fslot + xr * 16, 0, BUF_SC1);
                        }
                        if (stamp_on) lds_stamps[15] += sp2;
#pragma unroll
                        for (int x = 0; x < KX; ++x) {
                            if (x == xi) continue; // (uniform)
                            const unsigned fx = (unsigned)__builtin_amdgcn_readlane((int)fr.x, x), fy = (unsigned)__builtin_amdgcn_readlane((int)fr.y, x),
                                           fz = (unsigned)__builtin_amdgcn_readlane((int)fr.z, x);
                            const int fg = (int)((fz >> 26) & 3u);
                            if (fg) giveup = (giveup == 2 || fg == 2) ? 2 : 1;
                            need_exact |= (fz >> 31) != 0u;
                            const double fval = __builtin_fabs(mk_f64(fx, fy));
                            need_exact |= fval == best;
                            if (fval > best) {
                                best = fval;
                                wa_ = (int)((fz >> 14) & 0xFFFu);
                                wkx = fx;
                                wky = fy;
                                wkz = fz;
                            }
                        }
                    }
                    if (need_exact && !giveup) {
                        // exact comparison of (v*v, position key) over ALL full keys, four per lane at a time (slow, rare, exact): an
                        // agent without candidate carries value 0 and the largest position key; an infinite score ends the launch
                        double csc = -1.0;
                        unsigned cpk = XNOPOS;
                        int cag = 0;
                        unsigned cx = 0u, cy = 0u, cz = 0u;
                        const int lane_o = opaque_v(lane);
#pragma unroll 1
                        for (int j0 = 0; j0 < 4 * KX && !giveup; j0 += 4) {
                            u32x4 kq[4];
                            for (;;) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    kq[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, (int)k3_base + (par * NW + min(lane_o + 64 * (j0 + j), NW - 1)) * 16, 0, BUF_SC1);
                                bool ok = true;
#pragma unroll
                                for (int j = 0; j < 4; ++j) ok &= ((kq[j].x ^ kq[j].y ^ kq[j].z ^ kq[j].w) == tag);
                                if (__all(ok)) break;
                                xcd_poll_again();
                                if (++spins > XSPIN) {
                                    giveup = 1;
                                    break;
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int ag = lane_o + 64 * (j0 + j);
                                const unsigned rp_ = rowpos[kq[j].z & X2_ROW_MASK], cp_ = colpos[min(ag + NW * (int)((kq[j].z >> X2_QSHIFT) & 3u), N - 1)];
                                const unsigned pkey = ROWMAJOR ? ((rp_ << X2_PSHIFT) | cp_) : ((cp_ << X2_PSHIFT) | rp_);
                                const unsigned pk = ((ag < NW) && (kq[j].z & X2_META_VALID)) ? pkey : XNOPOS;
                                const double v = mk_f64(kq[j].x, kq[j].y);
                                double sc = v * v;
                                sc = (ag < NW) ? sc : -2.0;
                                const bool better = (sc > csc) | ((sc == csc) & (pk < cpk));
                                csc = better ? sc : csc;
                                cpk = better ? pk : cpk;
                                cag = better ? ag : cag;
                                cx = better ? kq[j].x : cx;
                                cy = better ? kq[j].y : cy;
                                cz = better ? kq[j].z : cz;
                            }
                        }
                        const double gmax = wave_max_f64(csc);
                        if (!(gmax < __builtin_huge_val()) && !giveup) giveup = 2;
                        const unsigned gpos = wave_min_u32((csc == gmax) ? cpk : XNOPOS);
                        const unsigned long long sel = __ballot((csc == gmax) & (cpk == gpos));
                        const int wl = sel ? (int)__builtin_ctzll(sel) : 0;
                        wa_ = __builtin_amdgcn_readlane(cag, wl);
                        wkx = (unsigned)__builtin_amdgcn_readlane((int)cx, wl);
                        wky = (unsigned)__builtin_amdgcn_readlane((int)cy, wl);
                        wkz = (unsigned)__builtin_amdgcn_readlane((int)cz, wl);
                    }
                    wkz &= 0x3FFFu; // (row index, column slot, valid bit)
                }
            }
            if (wave == 0 && lane == 0) {
                if (giveup) {
                    ctl[1] = giveup;
                    atomicExch(reinterpret_cast<int*>(lds_ptrs[0]) + 1, giveup);
                    if (lds_ptrs[1]) reinterpret_cast<volatile int*>(lds_ptrs[1])[5] = giveup;
                }
                int4 rec;
                rec.x = wa_ | (giveup << 30) | (T4A_X2_NOBARB ? ((((kn & 0x7F) + 1)) << 16) : 0);
                rec.y = (int)wkx;
                rec.z = (int)wky;
                rec.w = (int)wkz;
                *reinterpret_cast<int4*>(ctl + 4) = rec;
            }
            XSTAMP(9);
        }
#if !T4A_X2_NOBARB
        __syncthreads(); // (B)
#endif
        XSTAMP(3);
        int4 rec; // the record in ONE LDS round trip
        {
            int zero = 0;
#if T4A_X2_NOBARB
            for (;;) {
                asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rec) : "v"(zero), "n"(L::o_wi + 16) : "memory");
                if (((__builtin_amdgcn_readfirstlane(rec.x) >> 16) & 0xFF) == ((kn & 0x7F) + 1)) break;
#if T4A_X2_NOBARB > 1
                __builtin_amdgcn_s_sleep(T4A_X2_NOBARB - 1); // (spinning waves take issue slots from the polling wave on their SIMD)
#endif
            }
#else
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rec) : "v"(zero), "n"(L::o_wi + 16) : "memory");
#endif
        }
        const unsigned recw = (unsigned)__builtin_amdgcn_readfirstlane(rec.x) & (T4A_X2_NOBARB ? 0xC000FFFFu : 0xFFFFFFFFu);
        if (recw >> 30) {
            timed_out = true;
            break;
        }
        const int wag = (int)recw;
        // the winner did not speculate: its column goes out now
        if (g == wag && !early_pub) {
            if (qstar == 0) xcd_publish_column<0, RPT, ST_AUX>(a[0], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 1) if (qstar == 1) xcd_publish_column<1, RPT, ST_AUX>(a[1], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 2) if (qstar == 2) xcd_publish_column<2, RPT, ST_AUX>(a[2], mail, myslot, tag, 4 * CS);
            if constexpr (CPT > 3) if (qstar == 3) xcd_publish_column<3, RPT, ST_AUX>(a[3], mail, myslot, tag, 4 * CS);
        }
        // everybody fetches the winner's full key (one granule, the same for all lanes) and — waves 1 .. 7 — its rows of the
        // winner's column: lane + 64 (sr0 + 7 j)
        const int slot_off = (int)cols_base + (par * NW + wag) * SLOTB + ((lane >> 4) + 4 * sr0) * CS + (lane & 15) * 16;
        u32x4 cc[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (sr0 >= 0 && sr0 + X2_DIVW * j < RPT) cc[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, slot_off + j * X2_DIVW * 4 * CS, 0, BUF_SC1);
#if T4A_X2_DUPLOAD
        u32x4 cd[XR];
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (sr0 >= 0 && sr0 + X2_DIVW * j < RPT) cd[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, slot_off + j * X2_DIVW * 4 * CS, 0, BUF_SC1);
#endif
        // who sits at position kn now (the polling wave moves them behind its stop test)
        int rk_ = 0, ck_ = 0;
        int prp_ = 0, pcp_ = 0; // ... and where the pivot's row and column sit
        if (wave == 0) {
            rk_ = posrow[kn];
            ck_ = poscol[kn];
            prp_ = rowpos[rec.w & (int)X2_ROW_MASK];
            pcp_ = colpos[min(wag + NW * ((rec.w >> X2_QSHIFT) & 3), N - 1)];
        }
        const double wval = mk_f64((unsigned)__builtin_amdgcn_readfirstlane(rec.y), (unsigned)__builtin_amdgcn_readfirstlane(rec.z));
        const unsigned wmeta = (unsigned)__builtin_amdgcn_readfirstlane(rec.w);
        const int irow_p = (int)(wmeta & X2_ROW_MASK);
        const int qslot = (int)((wmeta >> X2_QSHIFT) & 3u);
        XSTAMP(10);
        // (the shared reciprocal of the pivot does not depend on the column: it is formed while the column travels)
        const bool p_mid = exp_mid(wval);
        double rp = refined_rcp(wval);
        asm volatile("" : "+v"(rp)); // (formed HERE, while the column travels: left to itself the compiler sinks the five dependent operations behind the column wait)
        prev_sq = wval * wval;
        {
            // the pivot row: its entries in the columns of the trailing block (and the pivot column) are the finished row kn
            // of U.  They are broadcast as u and saved to the side buffer.  The row itself is NOT touched: its l is
            // pivot / pivot = 1.0, so the rank-1 update below leaves exact zeros (x - 1.0 x) in every column that stays in the
            // trailing block — from then on the row takes part as l = 0 / a = 0 without any row mask.
            const int ls = irow_p & 63, rs = irow_p >> 6;
            // (all owned columns back to back — one index-mode region; a column that left the trailing block yields a u that
            // is never used)
            double ue[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) ue[q] = a[q].dyn(rs);
#pragma unroll
            for (int q = 0; q < CPT; ++q) u[q] = readlane_f64(ue[q], ls);
            if (p.urows) {
#pragma unroll
                for (int q = 0; q < CPT; ++q)
                    if ((active & (1u << q)) && lane == ls) p.urows[(unsigned)(kn * N + (g + NW * q))] = u[q];
            }
        }
        if (g == wag) active &= ~(1u << qslot); // the pivot column leaves the trailing block (its registers keep the un-scaled column)
        if (wave == 0) {
            // stop tests on the pivot magnitude sqrt(v*v), in the reference's order (matrixlu.rs:757-781); while v*v is a normal
            // number the square root of the rounded square is |v| itself (the software square root stays on the cold path)
            const double wsq = wval * wval;
            double pivot_abs = __builtin_fabs(wval);
            if (!(wsq >= 2.2250738585072014e-308 && wsq < __builtin_huge_val())) pivot_abs = sqrt(mk_f64((unsigned)opaque_v((int)lo32(wsq)), hi32(wsq)));
            error = pivot_abs;
            int stop = 0;
            if (kn > 0 && (pivot_abs < rel_tol_v * max_error || pivot_abs < abs_tol_v)) stop = 1;
            else if (pivot_abs <= min_pivot_abs) stop = 1;
            else max_error = fmax(max_error, pivot_abs);
            if (lane == 0) {
                if (stop) {
                    ctl[0] = 1;
                } else {
                    // permutation bookkeeping (swap_rows / swap_cols of the reference as index tables): the row / column that
                    // sat at position kn moves to the pivot's old position, the pivot's to kn
                    const int prp = prp_, pcp = pcp_;
                    const int pc = wag + NW * qslot; // original index of the pivot column
                    posrow[prp] = (unsigned short)rk_;
                    posrow[kn] = (unsigned short)irow_p;
                    rowpos[rk_] = (unsigned short)prp;
                    rowpos[irow_p] = (unsigned short)kn;
                    poscol[pcp] = (unsigned short)ck_;
                    poscol[kn] = (unsigned short)pc;
                    colpos[ck_] = (unsigned short)pcp;
                    colpos[pc] = (unsigned short)kn;
                    lds_pivots[kn] = wval;
                }
            }
        }
        XSTAMP(11);

        // ---- pivot column -> l = column / pivot, parked in LDS for everybody ----
        if (sr0 >= 0) {
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (sr0 + X2_DIVW * j < RPT) ok &= ((cc[j].x ^ cc[j].y ^ cc[j].z ^ cc[j].w) == tag);
#if T4A_X2_DUPLOAD
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (sr0 + X2_DIVW * j < RPT) asm volatile("" ::"v"(cd[j].x), "v"(cd[j].w)); // (the second copy must have arrived too)
#endif
                if (__all(ok)) break;
                xcd_poll_again();
                if (++spins > XSPIN) {
                    atomicExch(reinterpret_cast<int*>(lds_ptrs[0]) + 1, 1);
                    if (lds_ptrs[1]) reinterpret_cast<volatile int*>(lds_ptrs[1])[5] = 1;
                    ctl[1] = 1; // observed by everybody after barrier (C)
                    break;
                }
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (sr0 + X2_DIVW * j < RPT) cc[j] = __builtin_amdgcn_raw_buffer_load_b128(mail, slot_off + j * X2_DIVW * 4 * CS, 0, BUF_SC1);
            }
            XSTAMP(12);
            // x / p through the shared refined reciprocal (bitwise the IEEE quotient, see refined_rcp); zeros keep the sign
            // rule through x * rp; anything unusual takes the full division
            double lq[XR];
            bool slow = false;
#pragma unroll
            for (int j = 0; j < XR; ++j) {
                lq[j] = 0.0;
                if (sr0 + X2_DIVW * j < RPT) {
                    const double x = mk_f64(cc[j].x, cc[j].y);
                    const double q0 = x * rp;
                    const double qf = __builtin_fma(__builtin_fma(-wval, q0, x), rp, q0);
                    lq[j] = (x == 0.0) ? q0 : qf;
                    // |x| <= |pivot| for every entry of the pivot column (full pivoting): with a mid-range pivot only a tiny
                    // non-zero x can leave the range the fast quotient is proven for
                    slow |= (__builtin_fabs(x) < 4.909093465297727e-91) & (x != 0.0); // 2^-300
                }
            }
            slow |= !p_mid;
            if (__ballot(slow) != 0ull) {
#pragma unroll
                for (int j = 0; j < XR; ++j)
                    if (sr0 + X2_DIVW * j < RPT) {
                        const double x = mk_f64(cc[j].x, cc[j].y);
                        if (!(p_mid & (exp_mid(x) | (x == 0.0)))) lq[j] = x / wval;
                    }
            }
            // (rows pivoted before hold exact zeros in the published column: l = 0; the pivot row gets pivot / pivot = 1;
            // slot rows beyond M are zeros divided by the pivot)
#pragma unroll
            for (int j = 0; j < XR; ++j)
                if (sr0 + X2_DIVW * j < RPT) lbuf[lane * LSTR + sr0 + X2_DIVW * j] = lq[j];
        }
        XSTAMP(4);
        __syncthreads(); // (C)
        XSTAMP(13);
        // the verdicts first (one LDS read in front of the l reads: the wait for it leaves those in flight)
        int2 fl; // [0] stop verdict [1] give-up
        {
            static_assert(L::o_wi % 8 == 0, "the verdict pair is read with one 8-byte LDS load");
            int zero = 0;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(fl) : "v"(zero), "n"(L::o_wi) : "memory");
        }
        // (plans with more than 16 row slots read and apply l in two halves: 2 x 24 l registers beside a 96-register slab spilled)
        constexpr int NH = RPT > 16 ? 2 : 1, HR = RPT / NH;
        static_assert(RPT % NH == 0, "row slots split evenly into the halves of the l read");
        xvec<HR> l;
#pragma unroll
        for (int r = 0; r < HR; ++r) l[r] = lbuf[lane * LSTR + r];
        {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fl) : "n"((HR + 1) / 2) : "memory");
            if (__builtin_amdgcn_readfirstlane(fl.y)) {
                timed_out = true;
                break;
            }
            if (__builtin_amdgcn_readfirstlane(fl.x)) break; // stopped: pivot kn is not applied
        }
        XSTAMP(7);
        // =====================================================================================
        // elimination step kn: the trailing block gets the rank-1 update (update_trailing_submatrix, matrixlu.rs:593-612)
        // fused with the per-column maxima for the next arg-max; the pivot column keeps its un-scaled entries (L = column /
        // pivot is formed when the factored matrix is written out — the column is never read again)
        // =====================================================================================
        if constexpr (NH == 1) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                mq[q] = -1.0;
                if (active & (1u << q)) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) { // in place (see sub_in_place); un-fused, one rounding per operation like the reference
                        double t = a[q].get(r);
                        sub_in_place(t, l[r] * u[q]);
                        a[q].set(r, t);
                    }
                    double m0 = -1.0, m1 = -1.0;
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        if (r & 1) m1 = vmax_abs(m1, a[q].get(r));
                        else m0 = vmax_abs(m0, a[q].get(r));
                    }
                    mq[q] = vmax(m0, m1);
                }
            }
        } else {
            double m0[CPT], m1[CPT];
#pragma unroll
            for (int q = 0; q < CPT; ++q) m0[q] = m1[q] = -1.0;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                if (h > 0) {
                    asm volatile("" ::: "memory"); // (the second half of l is read only now: its registers are the first half's)
#pragma unroll
                    for (int r = 0; r < HR; ++r) l[r] = lbuf[lane * LSTR + h * HR + r];
                }
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    if (active & (1u << q)) {
#pragma unroll
                        for (int r = 0; r < HR; ++r) {
                            double t = a[q].get(h * HR + r);
                            sub_in_place(t, l[r] * u[q]);
                            a[q].set(h * HR + r, t);
                        }
#pragma unroll
                        for (int r = 0; r < HR; ++r) {
                            if (r & 1) m1[q] = vmax_abs(m1[q], a[q].get(h * HR + r));
                            else m0[q] = vmax_abs(m0[q], a[q].get(h * HR + r));
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < CPT; ++q) mq[q] = (active & (1u << q)) ? vmax(m0[q], m1[q]) : -1.0;
        }
        npiv = kn + 1;
        XSTAMP(0);
    }

    // ---- results ----
    // (the launch arguments of the epilogue are read again from the kernel-argument segment: kept from the prologue they would
    // occupy — i.e. spill — a dozen scalar register pairs across the whole step loop)
    const RrluXcdArgs* pe_ptr = pk;
    asm volatile("" : "+s"(pe_ptr));
    const RrluXcdArgs& pe = *pe_ptr;
    // (and the shape: the row / column masks of the matrix load would otherwise be kept, as scalar register pairs, for the write-out)
    const int Me = opaque_s(M), Ne = opaque_s(N);
    const unsigned long long t_done = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
    __syncthreads(); // (a give-up of the last step, the tables of the last applied step)
    if (ctl[1]) timed_out = true;
    if (npiv >= (Me < Ne ? Me : Ne)) error = 0.0; // matrixlu.rs:811-813
    if (rank == 0 && tid == 0) {
        pe.iresult[0] = npiv;
        pe.dresult[0] = error; // tid 0 belongs to the polling wave, which keeps the error
    }
    if (stamp_on) {
        for (int e = 0; e < 16; ++e) pe.stamps[e] = lds_stamps[e];
        pe.stamps[16] = t_elected - t_entry; // fixed part of a launch: election ...
        pe.stamps[17] = t_loop - t_elected;  // ... matrix load, tables, first maxima ...
        pe.stamps[18] = t_done - t_loop;     // (the pivot steps)
    }
    if (timed_out) return;
    // permutations: the tables are stable since the last barrier; device block and host mirror are written side by side
    if (rank == 0) {
        int* const h_rp = pe.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(pe.h_block) + (reinterpret_cast<const char*>(pe.row_perm) - reinterpret_cast<const char*>(pe.dresult))) : nullptr;
        int* const h_cp = pe.h_block ? reinterpret_cast<int*>(reinterpret_cast<char*>(pe.h_block) + (reinterpret_cast<const char*>(pe.col_perm) - reinterpret_cast<const char*>(pe.dresult))) : nullptr;
        for (int i = tid; i < Me; i += XT) {
            const int v = posrow[i];
            pe.row_perm[i] = v;
            if (h_rp) h_rp[i] = v;
        }
        for (int j = tid; j < Ne; j += XT) {
            const int v = poscol[j];
            pe.col_perm[j] = v;
            if (h_cp) h_cp[j] = v;
        }
        unsigned long long* const h_pv = pe.h_block ? pe.h_block + (reinterpret_cast<const char*>(pe.pivot_vals) - reinterpret_cast<const char*>(pe.dresult)) / 8 : nullptr;
        for (int e = tid; e < npiv; e += XT) {
            const double v = lds_pivots[e];
            pe.pivot_vals[e] = v;
            if (h_pv) h_pv[e] = (unsigned long long)__double_as_longlong(v);
        }
    }
    // factored matrix in permuted coordinates: rows of U come from the side buffer (this wave wrote them itself), the rest
    // (L below the diagonal, the untouched trailing block) from the registers.  (Rows pivoted before hold zeros in the columns
    // that stayed in the trailing block: those entries are rows of U and come from the side buffer.)
    int nan_seen = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = lane + 64 * r;
            if (g + NW * q < Ne && i < Me) {
                const int cp = colpos[g + NW * q], rp = rowpos[i];
                const bool from_u = (rp < npiv) && (cp >= rp);
                double v = a[q].get(r);
                const bool in_l = (cp < npiv) && (rp > cp);
                if (in_l) { // scale_column_tail (matrixlu.rs:562-577), deferred: the same division the step itself used
                    const double pv = lds_pivots[cp];
                    v = xcd_div(v, pv, refined_rcp(pv), exp_mid(pv));
                    if (v != v) nan_seen = 1;
                }
                if (pe.Aout) {
                    if (from_u)
                        v = __longlong_as_double((long long)__hip_atomic_load(
                            reinterpret_cast<const unsigned long long*>(pe.urows) + ((size_t)rp * Ne + (g + NW * q)), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT));
                    if (pe.out_transposed)
                        pe.Aout[(size_t)rp * Ne + cp] = v;
                    else
                        pe.Aout[(size_t)cp * Me + rp] = v;
                }
            }
        }
    if (nan_seen) {
        atomicExch(&pe.iresult[2], 1);
        if (pe.h_block) ((volatile int*)pe.h_block)[6] = 1;
    }
    // host-visible header (the pivot values went to the mirror with the permutations, the two flag words belong to their setters)
    if (pe.h_block && rank == 0 && tid == 0) {
        pe.h_block[0] = (unsigned long long)__double_as_longlong(error);
        pe.h_block[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pe.dresult) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((volatile int*)pe.h_block)[4] = npiv;
        // completion token: the host accepts the result only if rank 0 ran to its end in THIS launch (a launch whose
        // workgroups never met the elected XCD would otherwise leave an all-zero block behind)
        ((volatile int*)pe.h_block)[7] = (int)pe.salt;
        if (pe.ts_u64 > 0) {
            pe.h_block[pe.ts_u64] = ts_begin;
            pe.h_block[pe.ts_u64 + 1] = wall_clock64();
        }
        reinterpret_cast<unsigned long long*>(pe.dresult)[1] = 0ull; // clean header for the next launch (every agent's atomicMax is long done)
        if (pe.dims) { // bond chain: the device-side completion token for the next preparation kernel (max |a| stays in the mirror)
            __threadfence();
            pe.iresult[3] = (int)pe.salt;
        }
        if (stamp_on) pe.stamps[19] = __builtin_amdgcn_s_memtime() - t_done; // ... write-out and host mirror
    }
    // bond chain without per-launch host mirror: the device block is complete as it is (error, max |a|, rank, flags, pivot
    // values, permutations) and is copied to the host once, behind the whole chain; it only lacks the time stamps and the token
    if (!pe.h_block && pe.dims && rank == 0 && tid == 0) {
        if (pe.ts_u64 > 0) {
            unsigned long long* const blk = reinterpret_cast<unsigned long long*>(pe.dresult);
            blk[pe.ts_u64] = ts_begin;
            blk[pe.ts_u64 + 1] = wall_clock64();
        }
        __threadfence();
        pe.iresult[3] = (int)pe.salt;
    }
}

#if defined(T4A_XCD2_MULTI_TU)
template <int RPT, int CPT, bool ROWMAJOR, int KX>
__global__ void __launch_bounds__(XT) __attribute__((amdgpu_waves_per_eu(XWAVES / 4, XWAVES / 4))) rrlu_xcd2m_kernel(RrluXcdArgs p)
{
    rrlu_xcd2_body<RPT, CPT, ROWMAJOR, KX>(p, reinterpret_cast<const RrluXcdArgs*>(kernarg_base()));
}
template <int RPT, int CPT, bool ROWMAJOR, int KX> void xcd2m_launch_tie(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once; // (launches come from several host threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_xcd2m_kernel<RPT, CPT, ROWMAJOR, KX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_xcd2m_kernel<RPT, CPT, ROWMAJOR, KX>), dim3(plan.grid), dim3(XT), plan.lds_bytes, stream, a);
}
template <int RPT, int CPT, int KX> bool xcd2m_launch_rc(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    if (plan.RPT != RPT || plan.CPT != CPT || plan.K != KX) return false;
    if (a.tie_row_major) xcd2m_launch_tie<RPT, CPT, true, KX>(plan, a, stream);
    else xcd2m_launch_tie<RPT, CPT, false, KX>(plan, a, stream);
    return true;
}
#elif !defined(T4A_XCD_GROUP_TU)
template <int RPT, int CPT, bool ROWMAJOR>
__global__ void __launch_bounds__(XT) __attribute__((amdgpu_waves_per_eu(XWAVES / 4, XWAVES / 4))) rrlu_xcd2_kernel(RrluXcdArgs p)
{
    rrlu_xcd2_body<RPT, CPT, ROWMAJOR>(p, reinterpret_cast<const RrluXcdArgs*>(kernarg_base()));
}
template <int RPT, int CPT, bool ROWMAJOR> void xcd2_launch_tie(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once; // (launches come from several host threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_xcd2_kernel<RPT, CPT, ROWMAJOR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_xcd2_kernel<RPT, CPT, ROWMAJOR>), dim3(plan.grid), dim3(XT), plan.lds_bytes, stream, a);
}

template <int RPT, int CPT> void xcd2_launch_rc(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    if (a.tie_row_major) xcd2_launch_tie<RPT, CPT, true>(plan, a, stream);
    else xcd2_launch_tie<RPT, CPT, false>(plan, a, stream);
}

template <int RPT> void xcd2_launch_r(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    switch (plan.CPT) {
    case 1: xcd2_launch_rc<RPT, 1>(plan, a, stream); break;
    case 2: xcd2_launch_rc<RPT, 2>(plan, a, stream); break;
    case 3: if constexpr (RPT * 3 <= XCD_MAX_VALUES) xcd2_launch_rc<RPT, 3>(plan, a, stream); break;
    default: if constexpr (RPT * 4 <= XCD_MAX_VALUES) xcd2_launch_rc<RPT, 4>(plan, a, stream); break;
    }
}

#else
// Group launch: eight factorisations, one per XCD (see rrlu_xcd_group_kernel in kernels_rrlu_xcd.hip)
template <int RPT, int CPT, bool ROWMAJOR>
__global__ void __launch_bounds__(XT) __attribute__((amdgpu_waves_per_eu(XWAVES / 4, XWAVES / 4))) rrlu_xcd2_group_kernel(RrluXcdGroupArgs g)
{
    (void)g;
    const unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)xcc_id()) & 7u;
    const RrluXcdArgs* const slot = reinterpret_cast<const RrluXcdArgs*>(kernarg_base() + (size_t)x * sizeof(RrluXcdArgs));
    rrlu_xcd2_body<RPT, CPT, ROWMAJOR>(*slot, slot);
}

template <int RPT, int CPT, bool ROWMAJOR> void xcd2_group_launch_tie(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, hipStream_t stream)
{
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_xcd2_group_kernel<RPT, CPT, ROWMAJOR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL((rrlu_xcd2_group_kernel<RPT, CPT, ROWMAJOR>), dim3(plan.grid), dim3(XT), plan.lds_bytes, stream, a);
}
template <int RPT, int CPT> void xcd2_group_launch_rc(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool row_major, hipStream_t stream)
{
    if (row_major) xcd2_group_launch_tie<RPT, CPT, true>(plan, a, stream);
    else xcd2_group_launch_tie<RPT, CPT, false>(plan, a, stream);
}
template <int RPT> void xcd2_group_launch_r(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool row_major, hipStream_t stream)
{
    switch (plan.CPT) {
    case 1: xcd2_group_launch_rc<RPT, 1>(plan, a, row_major, stream); break;
    case 2: xcd2_group_launch_rc<RPT, 2>(plan, a, row_major, stream); break;
    case 3: if constexpr (RPT * 3 <= XCD_MAX_VALUES) xcd2_group_launch_rc<RPT, 3>(plan, a, row_major, stream); break;
    default: if constexpr (RPT * 4 <= XCD_MAX_VALUES) xcd2_group_launch_rc<RPT, 4>(plan, a, row_major, stream); break;
    }
}

#endif

} // namespace

#if defined(T4A_XCD2_MULTI_TU)
// the plans rrlu_xcd_make_plan(..., allow_big) hands out (kernels_rrlu_xcd.hip): one XCD with 24 row slots, or three XCDs
void rrlu_xcd2m_launch(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    const bool ok = xcd2m_launch_rc<24, 1, 1>(plan, a, stream) || xcd2m_launch_rc<24, 2, 1>(plan, a, stream) ||
                    xcd2m_launch_rc<16, 2, 3>(plan, a, stream) || xcd2m_launch_rc<24, 2, 3>(plan, a, stream) ||
                    xcd2m_launch_rc<16, 3, 2>(plan, a, stream) || xcd2m_launch_rc<24, 2, 2>(plan, a, stream);
    if (!ok) throw std::runtime_error("no kernel instantiation for this multi-XCD rrLU plan");
}
#elif !defined(T4A_XCD_GROUP_TU)
void rrlu_xcd2_launch(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    switch (plan.RPT) {
#ifdef T4A_XCD_DEV
    case 2: xcd2_launch_r<2>(plan, a, stream); break;
    default: xcd2_launch_r<12>(plan, a, stream); break;
#else
    case 1: xcd2_launch_r<1>(plan, a, stream); break;
    case 2: xcd2_launch_r<2>(plan, a, stream); break;
    case 3: xcd2_launch_r<3>(plan, a, stream); break;
    case 4: xcd2_launch_r<4>(plan, a, stream); break;
    case 6: xcd2_launch_r<6>(plan, a, stream); break;
    case 8: xcd2_launch_r<8>(plan, a, stream); break;
    case 12: xcd2_launch_r<12>(plan, a, stream); break;
    default: xcd2_launch_r<16>(plan, a, stream); break;
#endif
    }
}
#else
// (this half of the file is compiled as its own translation unit: kernels_rrlu_xcd2_group.hip)
void rrlu_xcd2_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool tie_row_major, hipStream_t stream)
{
    switch (plan.RPT) {
    case 1: xcd2_group_launch_r<1>(plan, a, tie_row_major, stream); break;
    case 2: xcd2_group_launch_r<2>(plan, a, tie_row_major, stream); break;
    case 3: xcd2_group_launch_r<3>(plan, a, tie_row_major, stream); break;
    case 4: xcd2_group_launch_r<4>(plan, a, tie_row_major, stream); break;
    case 6: xcd2_group_launch_r<6>(plan, a, tie_row_major, stream); break;
    case 8: xcd2_group_launch_r<8>(plan, a, tie_row_major, stream); break;
    case 12: xcd2_group_launch_r<12>(plan, a, tie_row_major, stream); break;
    default: xcd2_group_launch_r<16>(plan, a, tie_row_major, stream); break;
    }
}
#endif

} // namespace t4a
