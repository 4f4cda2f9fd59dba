"""GPU parity tests of PivotSearchStrategy::Rook through the C ABI:
t4a_gpu_luci_rook_f64, t4a_gpu_luci_blocks_f64 and the TCI2 driver with options.pivot_search = Rook.

The residuals of the rook search go through solve_matrix / mat_mul, which the reference takes from tenferro-rs
("parity unpinned" at the bit level).  The device follows the oracle's operation order, so pivot sequences are
compared exactly on generic inputs and values to 1e-10."""
import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_rook import UNIQUE

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(123)
PARITY = dict(nsearch=0, max_nglobal_pivot=0)


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def assert_matches_oracle(g, o, a, tol=1e-10):
    assert g.rank == o["rank"]
    assert np.array_equal(g.row_indices, o["row_indices"]) and np.array_equal(g.col_indices, o["col_indices"])
    scale = max(1.0, np.abs(a).max())
    assert np.abs(g.pivot_errors - o["pivot_errors"]).max() <= tol * scale
    assert np.abs(g.left - o["left"]).max() <= tol * max(1.0, np.abs(o["left"]).max())
    assert np.abs(g.right - o["right"]).max() <= tol * max(1.0, np.abs(o["right"]).max())


def test_rook_reference_cases(t4a):
    g = t4a.matrix_luci_factors_rook(UNIQUE, rel_tol=0.0)
    d = t4a.matrix_luci_factors_from_matrix(UNIQUE, rel_tol=0.0)
    assert g.rank == d.rank == 4
    assert np.array_equal(g.row_indices, d.row_indices) and np.array_equal(g.col_indices, d.col_indices)
    assert np.allclose(g.pivot_errors, d.pivot_errors, rtol=1e-14, atol=0)
    g = t4a.matrix_luci_factors_rook(UNIQUE, abs_tol=6.5)
    d = t4a.matrix_luci_factors_from_matrix(UNIQUE, abs_tol=6.5)
    assert g.rank == d.rank and np.array_equal(g.row_indices, d.row_indices)
    assert np.allclose(g.pivot_errors, d.pivot_errors, rtol=1e-14, atol=0)


def test_rook_blocks_never_requests_the_full_matrix(t4a):
    biggest = [0]

    def fill(rows, cols):
        biggest[0] = max(biggest[0], len(rows) * len(cols))
        return UNIQUE[np.ix_(rows, cols)]

    g = t4a.matrix_luci_factors_from_blocks(4, 4, fill, rel_tol=0.0)
    assert g.rank == 4 and biggest[0] < 16
    assert np.abs(g.left @ g.right - UNIQUE).max() < 1e-12


@pytest.mark.parametrize("shape,rank", [((30, 20), 5), ((17, 40), 7), ((64, 64), 12), ((200, 150), 20)])
@pytest.mark.parametrize("left", [True, False])
def test_rook_low_rank_matches_oracle(t4a, shape, rank, left):
    a = RNG.standard_normal((shape[0], rank)) @ RNG.standard_normal((rank, shape[1]))
    g = t4a.matrix_luci_factors_rook(a, rel_tol=1e-10, left_orthogonal=left)
    o = ob.luci_rook(a, rel_tol=1e-10, left_orthogonal=left)
    assert_matches_oracle(g, o, a, tol=1e-9)
    assert g.rank == rank and np.abs(g.left @ g.right - a).max() < 1e-9 * np.abs(a).max()


def test_rook_full_rank_cap_and_edge_cases(t4a):
    a = RNG.standard_normal((24, 24))
    g = t4a.matrix_luci_factors_rook(a, max_bond_dim=6, rel_tol=0.0)
    o = ob.luci_rook(a, max_bond_dim=6, rel_tol=0.0)
    assert_matches_oracle(g, o, a)
    assert g.pivot_errors[6] == g.pivot_errors[5]
    z = t4a.matrix_luci_factors_rook(np.zeros((3, 4)))
    assert z.rank == 0 and list(z.pivot_errors) == [0.0]
    g = t4a.matrix_luci_factors_rook(a[:5, :5], rel_tol=0.0)
    assert g.rank == 5 and g.pivot_errors[5] == 0.0 and np.abs(g.left @ g.right - a[:5, :5]).max() < 1e-10


def both(t4a, f, dims):
    g = t4a.TensorCI2(dims)
    g.set_function(f)
    o = ob.OracleTCI2(dims)
    o.set_function(f)
    o.set_pivot_search(1)
    return g, o


def assert_tci_match(g, o, n, tol=1e-10):
    for p in range(n):
        assert np.array_equal(g.i_set(p), o.i_set(p)), f"I set differs at site {p}"
        assert np.array_equal(g.j_set(p), o.j_set(p)), f"J set differs at site {p}"
    assert abs(g.max_sample_value() - o.max_sample_value()) <= 1e-15 * max(1.0, o.max_sample_value())
    for p in range(n):
        a, b = g.site_tensor(p), o.site_tensor(p)
        assert a.shape == b.shape
        if a.size:
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_tci2_rook_product_function(t4a):
    f = lambda i: (i[0] + 1.0) * (i[1] + 1.0) * (i[2] + 1.0)
    g, o = both(t4a, f, [3, 3, 3])
    opt = t4a.TCI2Options(tolerance=1e-12, pivot_search=t4a.TCI2Options.ROOK, **PARITY)
    g.crossinterpolate2([[2, 2, 2]], opt)
    o.crossinterpolate2([[2, 2, 2]], opt)
    assert g.link_dims() == [1, 1]
    assert_tci_match(g, o, 3)
    idx = [[a, b, c] for a in range(3) for b in range(3) for c in range(3)]
    assert np.abs(g.evaluate(idx) - np.array([f(i) for i in idx])).max() < 1e-10


def test_tci2_rook_generic_function_matches_oracle(t4a):
    f = lambda i: np.cos(0.7 * i[0] + 0.3 * i[1] * i[2] + 0.11 * i[3]) + 0.05 * i[1] + 0.013 * i[2] * i[3]
    g, o = both(t4a, f, [5, 4, 6, 5])
    opt = t4a.TCI2Options(tolerance=1e-9, max_iter=6, pivot_search=1, **PARITY)
    g.crossinterpolate2([[0, 0, 0, 0]], opt)
    o.crossinterpolate2([[0, 0, 0, 0]], opt)
    assert_tci_match(g, o, 4, tol=1e-8)
    rg, eg = g.history()
    ro, eo = o.history()
    assert list(rg) == list(ro) and np.allclose(eg, eo, rtol=1e-6, atol=1e-14)
    idx = RNG.integers(0, 4, size=(100, 4))
    want = np.array([f(i) for i in idx])
    assert np.abs(g.evaluate(idx) - want).max() < 1e-7


def test_tci2_rook_builtin_device_function(t4a):
    """Built-in device functor: rows / columns of Pi are evaluated on the GPU on demand (no host callback)."""
    spec = t4a.quantics_trig_exp(12)
    dims = [2] * 12
    g = t4a.TensorCI2(dims)
    g.set_function(spec)
    o = ob.OracleTCI2(dims)
    o.set_function(spec)
    o.set_pivot_search(1)
    opt = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=16, max_iter=4, pivot_search=1, **PARITY)
    g.crossinterpolate2([[0] * 12], opt)
    o.crossinterpolate2([[0] * 12], opt)
    assert_tci_match(g, o, 12, tol=1e-8)
    idx = RNG.integers(0, 2, size=(200, 12))
    exact = ob.fn_eval(spec, idx)
    assert np.abs(g.evaluate(idx) - exact).max() < 1e-6
