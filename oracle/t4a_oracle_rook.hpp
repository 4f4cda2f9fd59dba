// oracle/t4a_oracle_rook.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp, which includes this file right
// after its dense section; do not include it on its own).
// CPU restatement of the lazy block-rook pivot kernel behind PivotSearchStrategy::Rook:
//   crates/tensor4all-core/src/matrixluci/block_rook.rs   (residual_block :21, argmax_abs :46, rook_pivot :71,
//                                                          factorize_lazy :120)
//   crates/tensor4all-core/src/matrixluci/factors.rs      (load_block :14, CrossFactors :43-113)
//   crates/tensor4all-core/src/matrix_luci.rs             (factors_to_public :109-135,
//                                                          lazy_matrix_luci_factors_from_blocks :302-326)
// Parity: the residuals go through solve_matrix / mat_mul (tenferro-rs in the reference) => "parity unpinned" at
// the bit level; pinned by the reference's own rook tests (block_rook/tests.rs: rook == dense on the
// diagonally dominant 4x4, abs_tol stop, never requests the full matrix).
#pragma once

namespace t4a_oracle {

// fill_block(rows, cols, out): out[i + rows.size()*j] = A[rows[i], cols[j]]   (source.rs:15-24)
using BlockFn = std::function<void(const std::vector<size_t>&, const std::vector<size_t>&, double*)>;

struct PivotSelectionCore { // matrixluci/types.rs
    std::vector<size_t> row_indices, col_indices;
    std::vector<double> pivot_errors;
    size_t rank = 0;
};

namespace rook {

inline Matrix load_block(const BlockFn& src, const std::vector<size_t>& rows, const std::vector<size_t>& cols) // factors.rs:14
{
    Matrix m(rows.size(), cols.size());
    if (!m.a.empty()) src(rows, cols, m.a.data());
    return m;
}

// block_rook.rs:21-44: A[r,c] - A[r,J] * (A[I,J]^{-1} * A[I,c])
inline Matrix residual_block(const BlockFn& src, const std::vector<size_t>& rows, const std::vector<size_t>& cols,
                             const std::vector<size_t>& sel_rows, const std::vector<size_t>& sel_cols)
{
    Matrix residual = load_block(src, rows, cols);
    if (sel_rows.empty()) return residual;
    Matrix pivot = load_block(src, sel_rows, sel_cols);
    Matrix a_rj = load_block(src, rows, sel_cols);
    Matrix a_ic = load_block(src, sel_rows, cols);
    Matrix solved = solve(pivot, a_ic);
    Matrix approx = mat_mul(a_rj, solved);
    for (size_t j = 0; j < residual.nc; ++j)
        for (size_t i = 0; i < residual.nr; ++i) residual(i, j) = residual(i, j) - approx(i, j);
    return residual;
}

struct ArgMax {
    size_t row = 0, col = 0;
    double abs = 0.0;
};
inline ArgMax argmax_abs(const Matrix& m) // block_rook.rs:46-61
{
    ArgMax r;
    double best = -1.0;
    for (size_t c = 0; c < m.nc; ++c)
        for (size_t i = 0; i < m.nr; ++i) {
            const double v = std::fabs(m(i, c));
            if (v > best) {
                r.row = i;
                r.col = c;
                best = v;
            }
        }
    r.abs = std::fmax(best, 0.0);
    return r;
}

inline std::vector<size_t> remaining_indices(size_t total, const std::vector<size_t>& selected) // :63-69
{
    std::vector<char> used(total, 0);
    for (size_t i : selected) used[i] = 1;
    std::vector<size_t> r;
    for (size_t i = 0; i < total; ++i)
        if (!used[i]) r.push_back(i);
    return r;
}

struct Pivot {
    size_t row, col;
    double abs;
};
inline Pivot rook_pivot(const BlockFn& src, const std::vector<size_t>& rem_rows, const std::vector<size_t>& rem_cols,
                        const std::vector<size_t>& sel_rows, const std::vector<size_t>& sel_cols) // :71-118
{
    size_t cur_col = rem_cols[0];
    size_t cur_row = rem_rows[0];
    const size_t max_steps = rem_rows.size() + rem_cols.size() + 1;
    for (size_t it = 0; it < max_steps; ++it) {
        Matrix cr = residual_block(src, rem_rows, {cur_col}, sel_rows, sel_cols);
        cur_row = rem_rows[argmax_abs(cr).row];
        Matrix rr = residual_block(src, {cur_row}, rem_cols, sel_rows, sel_cols);
        const ArgMax am = argmax_abs(rr);
        const size_t next_col = rem_cols[am.col];
        if (next_col == cur_col) return {cur_row, cur_col, am.abs};
        cur_col = next_col;
    }
    Matrix rr = residual_block(src, {cur_row}, rem_cols, sel_rows, sel_cols);
    const ArgMax am = argmax_abs(rr);
    return {cur_row, rem_cols[am.col], am.abs};
}

inline PivotSelectionCore factorize_lazy(size_t nrows, size_t ncols, const BlockFn& src, const RrLUOptions& opt) // :120-190
{
    PivotSelectionCore out;
    const size_t full_rank = std::min(nrows, ncols);
    if (full_rank == 0) {
        out.pivot_errors = {0.0};
        return out;
    }
    const size_t max_bond = std::min(opt.max_bond_dim, full_rank);
    std::vector<size_t> sel_rows, sel_cols;
    std::vector<double> accepted;
    double max_error = 0.0;
    double last_error = std::numeric_limits<double>::quiet_NaN();
    while (sel_rows.size() < max_bond) {
        const std::vector<size_t> rem_rows = remaining_indices(nrows, sel_rows);
        const std::vector<size_t> rem_cols = remaining_indices(ncols, sel_cols);
        if (rem_rows.empty() || rem_cols.empty()) break;
        const Pivot p = rook_pivot(src, rem_rows, rem_cols, sel_rows, sel_cols);
        last_error = p.abs;
        if (!sel_rows.empty() && (p.abs < opt.rel_tol * max_error || p.abs < opt.abs_tol)) break;
        if (p.abs < std::numeric_limits<double>::epsilon()) break;
        max_error = std::fmax(max_error, p.abs);
        sel_rows.push_back(p.row);
        sel_cols.push_back(p.col);
        accepted.push_back(p.abs);
    }
    const size_t rank = sel_rows.size();
    if (rank >= full_rank)
        last_error = 0.0;
    else if (rank == max_bond && rank > 0)
        last_error = accepted[rank - 1];
    accepted.push_back(last_error);
    out.row_indices = sel_rows;
    out.col_indices = sel_cols;
    out.pivot_errors = accepted;
    out.rank = rank;
    return out;
}

} // namespace rook

// matrix_luci.rs:302-326 (+ CrossFactors::from_source factors.rs:58-70, factors_to_public matrix_luci.rs:109-135)
inline MatrixLuciFactors lazy_matrix_luci_factors_from_blocks(size_t nrows, size_t ncols, const BlockFn& fill_block,
                                                              const RrLUOptions& opt)
{
    PivotSelectionCore sel = rook::factorize_lazy(nrows, ncols, fill_block, opt);
    std::vector<size_t> all_rows(nrows), all_cols(ncols);
    for (size_t i = 0; i < nrows; ++i) all_rows[i] = i;
    for (size_t j = 0; j < ncols; ++j) all_cols[j] = j;
    Matrix pivot = rook::load_block(fill_block, sel.row_indices, sel.col_indices);
    Matrix pivot_cols = rook::load_block(fill_block, all_rows, sel.col_indices);
    Matrix pivot_rows = rook::load_block(fill_block, sel.row_indices, all_cols);
    MatrixLuciFactors f;
    f.row_indices = sel.row_indices;
    f.col_indices = sel.col_indices;
    f.pivot_errors = sel.pivot_errors;
    f.rank = sel.rank;
    if (opt.left_orthogonal) {
        // cols_solve_pivot (factors.rs:72-82): (P^T \ C^T)^T
        f.left = sel.rank ? transpose(solve(transpose(pivot), transpose(pivot_cols))) : Matrix(nrows, 0);
        f.right = pivot_rows;
    } else {
        f.left = pivot_cols;
        f.right = sel.rank ? solve(pivot, pivot_rows) : Matrix(0, ncols);
    }
    return f;
}

} // namespace t4a_oracle
