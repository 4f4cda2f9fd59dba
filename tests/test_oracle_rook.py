"""Pins the oracle's lazy block-rook kernel (oracle/t4a_oracle_rook.hpp) against the reference's rook tests
(crates/tensor4all-core/src/matrixluci/block_rook/tests.rs) and exercises PivotSearchStrategy::Rook in the TCI2
driver restatement.  CPU only."""
import numpy as np
import pytest

import oracle_binding as ob
from t4a_amd import TCI2Options

# block_rook/tests.rs:10-17 — the data vector is column-major
UNIQUE = np.array([9.0, 0.2, 0.3, 0.4, 0.1, 8.0, 0.2, 0.3, 0.2, 0.1, 7.0, 0.2, 0.3, 0.2, 0.1, 6.0]).reshape((4, 4), order="F")
RNG = np.random.default_rng(99)


def test_rook_matches_dense_kernel_on_unique_pivot_matrix():
    lazy = ob.luci_rook(UNIQUE, rel_tol=0.0)          # PivotKernelOptions::no_truncation
    dense = ob.luci(UNIQUE, rel_tol=0.0)
    assert lazy["rank"] == dense["rank"] == 4
    assert np.array_equal(lazy["row_indices"], dense["rows"]) and np.array_equal(lazy["col_indices"], dense["cols"])
    assert np.allclose(lazy["pivot_errors"], dense["pivot_errors"], rtol=1e-14, atol=0)
    assert lazy["max_block"] < 16                      # never asks for the whole matrix (tests.rs:73-93)


def test_rook_abs_tol_stop_matches_dense():
    lazy = ob.luci_rook(UNIQUE, abs_tol=6.5)
    dense = ob.luci(UNIQUE, abs_tol=6.5)
    assert lazy["rank"] == dense["rank"]
    assert np.array_equal(lazy["row_indices"], dense["rows"]) and np.array_equal(lazy["col_indices"], dense["cols"])
    assert np.allclose(lazy["pivot_errors"], dense["pivot_errors"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("left", [True, False])
def test_rook_reconstructs_low_rank(left):
    x = RNG.standard_normal((30, 5)) @ RNG.standard_normal((5, 20))
    r = ob.luci_rook(x, rel_tol=1e-10, left_orthogonal=left)
    assert r["rank"] == 5
    assert np.abs(r["left"] @ r["right"] - x).max() < 1e-10
    # cross interpolation property: exact on the pivot rows and columns
    ii, jj = r["row_indices"], r["col_indices"]
    rec = r["left"] @ r["right"]
    assert np.abs(rec[ii, :] - x[ii, :]).max() < 1e-12 and np.abs(rec[:, jj] - x[:, jj]).max() < 1e-12


def test_rook_edge_cases():
    z = ob.luci_rook(np.zeros((3, 4)))
    assert z["rank"] == 0 and len(z["pivot_errors"]) == 1 and z["pivot_errors"][0] == 0.0
    one = ob.luci_rook(RNG.standard_normal((6, 6)), max_bond_dim=2, rel_tol=0.0)
    assert one["rank"] == 2 and one["pivot_errors"][2] == one["pivot_errors"][1]   # cap reached: last accepted


def test_tci2_rook_product_function_exact():
    f = lambda i: (i[0] + 1.0) * (i[1] + 1.0) * (i[2] + 1.0)
    o = ob.OracleTCI2([3, 3, 3])
    o.set_function(f)
    o.set_pivot_search(1)
    o.crossinterpolate2([[2, 2, 2]], TCI2Options(tolerance=1e-12, nsearch=0, max_nglobal_pivot=0))
    assert o.link_dims() == [1, 1]
    idx = [[a, b, c] for a in range(3) for b in range(3) for c in range(3)]
    assert np.abs(o.evaluate(idx) - np.array([f(i) for i in idx])).max() < 1e-10


def test_tci2_rook_samples_fewer_points_than_full():
    f = lambda i: 1.0 / (1.0 + sum((x + 1.0) ** 2 for x in i))
    res = {}
    for strat in (0, 1):
        o = ob.OracleTCI2([8] * 4)
        o.set_function(f)
        o.set_pivot_search(strat)
        o.crossinterpolate2([[0] * 4], TCI2Options(tolerance=1e-8, max_iter=6, nsearch=0, max_nglobal_pivot=0))
        idx = RNG.integers(0, 8, size=(200, 4))
        err = np.abs(o.evaluate(idx) - np.array([f(i) for i in idx])).max()
        res[strat] = (o.n_evals(), err)
    assert res[0][1] < 1e-6 and res[1][1] < 1e-6
    assert res[1][0] < res[0][0]
