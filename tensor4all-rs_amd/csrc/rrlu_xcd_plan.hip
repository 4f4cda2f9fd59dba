// rrlu_xcd_plan.hip — launch plans of the single-XCD / multi-XCD register-resident rrLU kernels (kernels_rrlu_xcd2.hip,
// kernels_rrlu_xcd2m.hip): which instantiation (row slots per lane, columns per agent, XCDs) takes a given shape, and the size of
// its mailbox.  Host code only.  (Round 6: moved out of kernels_rrlu_xcd.hip when the first-generation kernel was retired — non-finite
// matrices, the one case it was kept for, go to the chip-wide kernels, which implement the NaN-incumbent rule of matrixlu.rs:480-519.)
#include "kernels_rrlu_xcd_common.hpp"

namespace t4a {

namespace {

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


} // namespace

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


} // namespace t4a
