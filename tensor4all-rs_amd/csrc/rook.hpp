// rook.hpp — lazy block-rook LUCI (PivotSearchStrategy::Rook) on the gfx950 engine.
// Mirrors crates/tensor4all-core/src/matrixluci/block_rook.rs (rook_pivot :71, factorize_lazy :120),
// factors.rs (CrossFactors :43-113) and matrix_luci.rs (lazy_matrix_luci_factors_from_blocks :302-326).
#pragma once

#include <functional>

#include "engine.hpp"

namespace t4a {

// Candidate-matrix source (matrixluci/source.rs:15-24): the rook search only ever asks for full columns and full
// rows, so that is the granularity here.  `column(c, d_out)` enqueues (or completes) the evaluation of the M
// entries A[:, c] into device memory, `row(r, d_out)` the N entries A[r, :].
struct RookSource {
    int M = 0, N = 0;
    std::function<void(int, double*)> column;
    std::function<void(int, double*)> row;
    // optional: the WHOLE matrix into device memory (column-major, ld = M) — sources whose entries cost nothing to evaluate on the device
    // (built-in functors, dense device matrices).  The search then runs as ONE persistent launch on the materialised matrix
    // (rook_dense_kernel, round 5) instead of one host round trip per visited row / column; what the lazy evaluator WOULD have
    // evaluated (full rows / columns visited) is still what sampled_max and n_evaluated report.
    std::function<void(double*)> full;
};

struct RookWork { // grow-only scratch, reusable across calls
    DevBuf<double> A, At, P, X, vec, res;
    DevBuf<int> I, J, rowsel, colsel, piv, info;
    DevBuf<unsigned long long> maxbits;
    DevBuf<double> dres;   // device-resident search: [0] last error [1] sampled max [2] evaluated entries [3..] accepted pivot errors
    DevBuf<int> ires;      // [0] rank [1] LU info (singular pivot block) [2] host syncs saved (visits) [4..] selected rows, then columns
    DevBuf<int> seen;      // visited flags: rows, then columns
    DevBuf<double> packed; // device-resident search: everything the host reads as ONE block (rook_dense_kernel) ...
    PinBuf<double> hpacked; // ... and its pinned landing place
    PinBuf<char> hdesc;     // pinned staging of the factor build's LU / triangular-solve descriptors
    PinBuf<unsigned long long> hfin; // pinned landing place of the final {sampled maximum, LU status}
    size_t n_device_searches = 0, n_device_visits = 0, n_host_searches = 0, n_host_syncs = 0; // statistics
    DevBuf<LuProblem> lup;
    DevBuf<TrsmProblem> trp;
};

// Runs the lazy rook factorisation; on return eng.left() is M x rank and eng.right() rank x N
// (factors_to_public, matrix_luci.rs:109-135).  row_perm / col_perm hold the selected rows / columns first.
// *sampled_max (in/out) follows LazyPiEvaluator::sampled_max (tensorci2.rs:2035-2142); *n_evaluated counts the
// entries that were evaluated (full rows / columns visited).
LuciResult rook_luci(Engine& eng, RookWork& w, const RookSource& src, const RrLUOptions& opts, double* sampled_max,
                     double* n_evaluated);

} // namespace t4a
