"""GPU parity tests of PivotSearchStrategy::Rook through the C ABI:
t4a_gpu_luci_rook_f64, t4a_gpu_luci_blocks_f64 and the TCI2 driver with options.pivot_search = Rook.

The residuals of the rook search go through solve_matrix / mat_mul, which the reference takes from tenferro-rs
("parity unpinned" at the bit level).  The device follows the oracle's operation order, so pivot sequences are
compared exactly on generic inputs and values to 1e-10."""
import os
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


def test_device_resident_search_and_host_driven_search_agree(t4a, tmp_path):
    """Round 5: sources that can materialise the matrix on the device (built-in functors, dense device matrices) run the whole pivot
    loop of the rook search as ONE persistent launch (rook_dense_kernel) — one host synchronisation per bond instead of two per visited
    row / column pair.  Same arithmetic, operation for operation, as the host-driven loop (T4A_ROOK_HOST=1, read once per process:
    a child process): selected rows / columns, pivot errors and the max_sample_value of the LAZY evaluator (visited rows / columns only)
    must be identical, and the device-resident run must have needed at least ten times fewer synchronisations."""
    import subprocess
    import sys
    code = r'''
import sys, os, json, numpy as np
sys.path.insert(0, "tensor4all-rs_amd/python")
import t4a_amd as t4a
spec = t4a.quantics_osc2d(16, k1=3, k2=5, k3=7, eps=0.3, k4=11, delta=0.2)
g = t4a.TensorCI2([2] * 16)
g.set_function(spec)
opt = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=24, max_iter=5, pivot_search=1, nsearch=0, max_nglobal_pivot=0)
g.crossinterpolate2([[0] * 16], opt)
rng = np.random.default_rng(5)
a = rng.standard_normal((90, 14)) @ rng.standard_normal((14, 70)) + 1e-9 * rng.standard_normal((90, 70))
r = t4a.matrix_luci_factors_rook(a, max_bond_dim=20, rel_tol=1e-12)
out = {"sets": [np.asarray(g.i_set(p)).ravel().tolist() + np.asarray(g.j_set(p)).ravel().tolist() for p in range(16)],
       "pivot_errors": [float(v) for v in g.pivot_errors()], "max_sample": g.max_sample_value(), "stats": g.rook_stats(),
       "rows": [int(v) for v in r.row_indices], "cols": [int(v) for v in r.col_indices], "perr": [float(v) for v in r.pivot_errors]}
print("RESULT" + json.dumps(out))
'''
    res = {}
    for name, env in (("device", {}), ("host", {"T4A_ROOK_HOST": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert line, r.stdout[-2000:] + r.stderr[-3000:]
        res[name] = __import__("json").loads(line[0][6:])
    d, h = res["device"], res["host"]
    assert d["sets"] == h["sets"] and d["rows"] == h["rows"] and d["cols"] == h["cols"]
    assert d["max_sample"] == h["max_sample"]
    assert np.allclose(d["pivot_errors"], h["pivot_errors"], rtol=1e-9, atol=0) and np.allclose(d["perr"], h["perr"], rtol=1e-9, atol=0)
    assert d["stats"]["device_searches"] > 0 and d["stats"]["host_searches"] == 0 and h["stats"]["device_searches"] == 0
    assert d["stats"]["host_syncs"] * 10 <= h["stats"]["host_syncs"], (d["stats"], h["stats"])
