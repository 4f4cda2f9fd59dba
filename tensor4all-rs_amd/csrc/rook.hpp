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
};

struct RookWork { // grow-only scratch, reusable across calls
    DevBuf<double> A, At, P, X, vec, res;
    DevBuf<int> I, J, rowsel, colsel, piv, info;
    DevBuf<unsigned long long> maxbits;
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
