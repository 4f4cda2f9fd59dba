"""GPU tests of the adaptive patching driver (BASELINE config 5) through the C ABI
(t4a_gpu_adaptive_interpolate_*, t4a_gpu_ptt_*): the reference's own test cases and parity with the CPU oracle
(same patch queue, same projectors in the same order, per-patch cores to 1e-9)."""
import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_patch import disjoint

pytestmark = pytest.mark.gpu
PARITY = dict(nsearch=0, max_nglobal_pivot=0)


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def assert_matches_oracle(g, o, tol=1e-9):
    assert len(g) == len(o)
    for k in range(len(g)):
        assert g.projector(k) == o.projector(k), f"projector of patch {k}"
        gc, oc = g.patch(k).site_tensors(), o.cores(k)
        for s, (a, b) in enumerate(zip(gc, oc)):
            assert a.shape == b.shape, f"patch {k} site {s}: {a.shape} vs {b.shape}"
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), f"patch {k} site {s}"


def test_reference_cases(t4a):
    f = lambda i: (i[0] + 1.0) * (i[1] + 2.0) * (i[2] + 3.0)
    r = t4a.adaptiveinterpolate(f, [2, 2, 2], [[1, 1, 1]], t4a.TCI2Options())
    assert len(r) == 1 and r.projector(0) == {}
    assert np.allclose(r.dense(), [6.0, 12.0, 9.0, 18.0, 8.0, 16.0, 12.0, 24.0], rtol=1e-12, atol=0)
    single = t4a.adaptiveinterpolate(lambda i: 10.0 if i[0] == 3 else float(i[0] * i[0] + 1), [4], [],
                                     t4a.TCI2Options(), n_initial_pivots=1)
    assert len(single) == 1 and list(single.dense()) == [1.0, 2.0, 5.0, 10.0]
    z = t4a.adaptiveinterpolate(lambda i: 0.0, [2, 2], [], t4a.TCI2Options())
    assert len(z) == 1 and list(z.dense()) == [0.0] * 4
    b = t4a.adaptiveinterpolate(lambda i: float(i[0] + i[1] + 1), [2, 2], [[1, 1]], t4a.TCI2Options())
    assert list(b.dense()) == pytest.approx([1.0, 2.0, 2.0, 3.0], abs=1e-12)


def test_rank_cap_forces_disjoint_exact_child_patches(t4a):
    f = lambda i: 2.0 if all(v == i[0] for v in i) else 0.5
    opt = t4a.TCI2Options(tolerance=1e-14, max_bond_dim=1, max_iter=4, ncheck_history=1, **PARITY)
    kw = dict(patch_order=[0, 1, 2], recycle_pivots=True)
    g = t4a.adaptiveinterpolate(f, [2, 2, 2], [[0, 0, 0], [1, 1, 1]], opt, **kw)
    o = ob.adaptiveinterpolate(f, [2, 2, 2], [[0, 0, 0], [1, 1, 1]], opt, **kw)
    assert len(g) >= 2 and disjoint(g.projectors())
    assert np.allclose(g.dense(), [2.0, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 2.0], rtol=1e-12, atol=0)
    assert_matches_oracle(g, o)


def test_invalid_inputs(t4a):
    f = lambda i: 1.0
    for kw in (dict(n_initial_pivots=0), dict(patch_order=[0, 0])):
        with pytest.raises(t4a.T4aError) as e:
            t4a.adaptiveinterpolate(f, [2, 2], [], t4a.TCI2Options(), **kw)
        assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError):
        t4a.adaptiveinterpolate(f, [2, 2], [[0, 2]], t4a.TCI2Options())


def test_generic_function_splits_like_the_oracle(t4a):
    """A function whose rank exceeds the cap: the queue splits along patch_order; every patch matches the oracle."""
    f = lambda i: np.cos(0.9 * i[0] + 0.37 * i[1] * i[2] + 0.21 * i[3] * i[4]) + 0.1 * i[2] * i[4] + 0.03 * i[1] * i[3]
    dims = [3, 4, 3, 4, 3]
    opt = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=3, max_iter=8, **PARITY)
    kw = dict(patch_order=[4, 0, 2, 1, 3], n_initial_pivots=2, recycle_pivots=True)
    piv = [[0, 0, 0, 0, 0], [2, 3, 2, 3, 2]]
    g = t4a.adaptiveinterpolate(f, dims, piv, opt, **kw)
    o = ob.adaptiveinterpolate(f, dims, piv, opt, **kw)
    assert len(g) > 1 and disjoint(g.projectors())
    assert_matches_oracle(g, o, tol=1e-8)
    grid = np.indices(dims[::-1]).reshape(5, -1)[::-1].T
    exact = np.array([f(p) for p in grid])
    assert np.abs(g.dense() - exact).max() < 1e-6
    assert np.abs(g.dense() - o.dense(dims)).max() < 1e-9


def test_cfg5_builtin_quantics_patches(t4a):
    """BASELINE config 5 in miniature: the oscillatory 2-variable quantics integrand with a rank cap far below its
    rank, so that the queue projects leading bits until every patch converges; the built-in device functor is
    restricted to the active sites of each patch by folding the projected digits into the weight tables."""
    n = 14
    spec = t4a.quantics_osc2d(n, k1=3, k2=5, k3=11, eps=0.5, k4=23, delta=0.5)
    opt = t4a.TCI2Options(tolerance=1e-7, max_bond_dim=12, max_iter=12, **PARITY)
    kw = dict(n_initial_pivots=3, recycle_pivots=True)
    piv = [[0] * n]
    g = t4a.adaptiveinterpolate(spec, [2] * n, piv, opt, **kw)
    o = ob.adaptiveinterpolate(spec, [2] * n, piv, opt, **kw)
    assert len(g) > 1 and disjoint(g.projectors())
    assert sum(2 ** (n - len(p)) for p in g.projectors()) == 2 ** n      # the patches tile the domain
    assert g.projectors() == [o.projector(k) for k in range(len(o))]
    pts = np.random.default_rng(5).integers(0, 2, size=(400, n))
    exact = ob.fn_eval(spec, pts)
    assert np.abs(g.evaluate(pts) - exact).max() < 1e-4
    assert np.abs(g.evaluate(pts) - o.evaluate(pts)).max() < 1e-8


def test_reference_fixtures_not_covered_above_run_through_the_device(t4a):
    """The remaining cases tests/test_oracle_patch.py pins to the reference (adaptive_interpolation/tests.rs), straight through the
    DEVICE path with the reference's own expectations — the host logic of csrc/patching.hip is not compared with its sibling in the
    oracle here, but with what the Rust tests assert (round-4 review, weak 1)."""
    # the batched callback is what gets called (tests.rs: batched evaluation), values exact
    calls = [0]

    class F:
        def __call__(self, i):
            return float(i[0] + i[1] + 1)

        def batched(self, pts):
            calls[0] += 1
            return [float(p[0] + p[1] + 1) for p in pts]

    r = t4a.adaptiveinterpolate(F(), [2, 2], [[1, 1]], t4a.TCI2Options())
    assert list(r.dense()) == pytest.approx([1.0, 2.0, 2.0, 3.0], abs=1e-12) and calls[0] > 0
    # issue #598 regression (tests.rs:434-491): near-zero child patches of a fused-quantics Gaussian mixture must be accepted; the
    # reference requires a valid, complete partition — disjoint projectors that tile the index space exactly once, finite values
    W, A, C, L, R = [1.3, 0.9, 0.9], [2.8, 5.4, 0.7], [(0.4, 0.1), (3.8, -0.8), (-5.5, -2.1)], 12.0, 7

    def batch(pts):
        pts = np.asarray(pts, dtype=np.int64).reshape(-1, R)
        sh = (R - 1 - np.arange(R))[None, :]
        ix = ((pts & 1) << sh).sum(axis=1)
        iy = (((pts >> 1) & 1) << sh).sum(axis=1)
        step = 2.0 * L / (1 << R)
        x, y = -L + ix * step, -L + iy * step
        return sum(W[i] * np.exp(-A[i] * ((x - C[i][0]) ** 2 + (y - C[i][1]) ** 2)) for i in range(3))

    class Mixture:
        def __call__(self, index):
            return float(batch([list(index)])[0])

        def batched(self, pts):
            return batch(pts)

    opt = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=12, max_iter=20, normalize_error=False, seed=1)
    g = t4a.adaptiveinterpolate(Mixture(), [4] * R, [], opt)
    assert len(g) > 1 and disjoint(g.projectors())
    assert sum(4 ** (R - len(g.projector(k))) for k in range(len(g))) == 4 ** R
    pts = np.random.default_rng(0).integers(0, 4, size=(300, R))
    vals = g.evaluate(pts)
    assert np.all(np.isfinite(vals))
