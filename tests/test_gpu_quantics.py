"""GPU tests of the quantics front end through the C ABI (t4a_gpu_quanticscrossinterpolate*, t4a_gpu_qtci_*):
the reference's own test cases (crates/tensor4all-quanticstci/src/quantics_tci/tests/mod.rs) and parity with the CPU oracle
(same pivot tables, cores to 1e-10, same evaluation cache)."""
import math

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def opts(t4a, **kw):
    return t4a.QtciOptions(**kw), ob.QtciOptions(**kw)


def assert_same(g, o, tol=1e-10):
    assert g.n_sites == o.n_sites and g.n_vars == o.n_vars and g.is_discretized() == o.is_discretized()
    assert g.history()[0] == o.history()[0]
    assert np.allclose(g.history()[1], o.history()[1], rtol=0, atol=1e-12)
    for k in range(1, g.n_sites):
        assert g.tree_pivots(range(k)).tolist() == o.tree_pivots(range(k)).tolist(), f"left pivots of bond {k}"
        assert g.tree_pivots(range(k, g.n_sites)).tolist() == o.tree_pivots(range(k, g.n_sites)).tolist()
    gc, oc = g.tensor_train().site_tensors(), o.cores()
    for s, (a, b) in enumerate(zip(gc, oc)):
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), f"core {s}"
    assert g.cachedata() == o.cachedata()


def test_reference_discrete_cases(t4a):
    # tests/mod.rs:41-114, 236-275, 357-396
    f = lambda idx: float(idx[0] + idx[1])
    go, oo = opts(t4a, tolerance=1e-10, n_random_init_pivot=3, unfolding_scheme=1, seed=1)
    g = t4a.quanticscrossinterpolate_discrete([4, 4], f, None, go)
    assert g.evaluate([[2, 3]])[0] == pytest.approx(5.0, abs=1e-8)
    assert g.evaluate([[0, 0]])[0] == pytest.approx(0.0, abs=1e-8)
    assert g.evaluate([[3, 3]])[0] == pytest.approx(6.0, abs=1e-8)
    assert 0 < g.rank() <= 3 and len(g.link_dims()) == 1 and not g.is_discretized()
    for quantics, val in g.cachedata().items():
        gi = g.quantics_to_grididx(list(quantics))
        assert abs(val - (gi[0] + gi[1])) < 1e-10
    assert_same(g, ob.quanticscrossinterpolate_discrete([4, 4], f, None, oo))
    one = t4a.quanticscrossinterpolate_discrete([4], lambda idx: 1.0, None, go)
    assert one.integral() == pytest.approx(one.sum(), abs=1e-10) and one.integral() == pytest.approx(4.0, abs=1e-8)
    with pytest.raises(t4a.T4aError) as e:  # tests/mod.rs:236-253: cachedata_origcoord on an inherent grid
        one.quantics_to_origcoord([0, 0])
    assert "original coordinates are only available for discretized grids" in str(e.value)
    prod = lambda idx: float(idx[0] * idx[1])
    g = t4a.quanticscrossinterpolate_discrete([4, 4], prod, [[0, 0], [1, 2]], go)
    assert g.evaluate([[3, 3]])[0] == pytest.approx(9.0, abs=1e-8)
    assert_same(g, ob.quanticscrossinterpolate_discrete([4, 4], prod, [[0, 0], [1, 2]], oo))


def test_reference_validation_errors(t4a):
    # tests/mod.rs:116-147, 166-190, 398-420, 453-503
    one = lambda x: 1.0
    bad = [lambda: t4a.quanticscrossinterpolate_discrete([5, 5], one),
           lambda: t4a.quanticscrossinterpolate_discrete([], one),
           lambda: t4a.quanticscrossinterpolate_discrete([4, 8], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([[], [0.0, 1.0]], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([[0.0, float("nan"), 1.0, 2.0]], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 1.0, 2.0]], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 2.0]], one),
           lambda: t4a.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 2.0, 3.0], list(map(float, range(8)))], one)]
    for call in bad:
        with pytest.raises(t4a.T4aError) as e:
            call()
        assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError) as e:
        t4a.quanticscrossinterpolate_discrete([4], one, [[4]], t4a.QtciOptions(n_random_init_pivot=0))
    assert "initial pivot [4] conversion failed" in str(e.value) and "Grid index 4" in str(e.value)
    with pytest.raises(t4a.T4aError) as e:
        t4a.quanticscrossinterpolate([3], one, [0.0], [1.0], include_endpoint=True, initial_pivots=[[8]],
                                     options=t4a.QtciOptions(n_random_init_pivot=0))
    assert "initial pivot [8] conversion failed" in str(e.value) and "Grid index 8" in str(e.value)
    with pytest.raises(t4a.T4aError):  # zero function: "initial pivots must not all evaluate to zero"
        t4a.quanticscrossinterpolate_discrete([4, 4], lambda idx: 0.0)


def test_reference_continuous_grid_cases(t4a):
    # tests/mod.rs:277-355, 422-451
    go, oo = opts(t4a, tolerance=1e-12, n_random_init_pivot=5, seed=4)
    sq = lambda x: x[0] * x[0]
    g = t4a.quanticscrossinterpolate([3], sq, [0.0], [1.0], include_endpoint=True, options=go)
    assert g.is_discretized() and g.rank() > 0
    for quantics, val in g.cachedata().items():
        x = g.quantics_to_origcoord(list(quantics))[0]
        assert abs(val - x * x) < 1e-10
    xs = np.arange(8) / 7.0
    assert np.allclose(g.evaluate(np.arange(8).reshape(-1, 1)), xs * xs, atol=1e-10)
    assert_same(g, ob.quanticscrossinterpolate([3], sq, [0.0], [1.0], include_endpoint=True, options=oo))
    c = t4a.quanticscrossinterpolate([4], lambda x: 1.0, [0.0], [1.0], options=go)
    assert c.integral() == pytest.approx(1.0, abs=1e-8) and c.grid_step() == [1.0 / 16.0]
    lin = t4a.quanticscrossinterpolate([3], lambda x: x[0], [0.0], [1.0], include_endpoint=True, initial_pivots=[[1], [4]],
                                       options=go)
    for quantics, val in lin.cachedata().items():
        assert abs(val - lin.quantics_to_origcoord(list(quantics))[0]) < 1e-10


def test_reference_from_arrays_cases(t4a):
    # tests/mod.rs:148-164, 505-541, 552-581
    go, oo = opts(t4a, tolerance=1e-10, n_random_init_pivot=2, unfolding_scheme=1, seed=2)
    g = t4a.quanticscrossinterpolate_from_arrays([[0.0, 0.5, 2.0, 5.0]], lambda x: x[0] + 1.0, None, go)
    assert g.evaluate([[1]])[0] == pytest.approx(1.5, abs=1e-8) and g.evaluate([[2]])[0] == pytest.approx(3.0, abs=1e-8)
    xv = [[0.0, 0.5, 2.0, 3.0], [0.0, 1.0, 2.0, 4.0]]
    add = lambda x: x[0] + x[1]
    g = t4a.quanticscrossinterpolate_from_arrays(xv, add, None, go)
    assert not g.is_discretized() and g.rank() > 0
    for quantics, val in g.cachedata().items():
        gi = g.quantics_to_grididx(list(quantics))
        assert abs(val - (xv[0][gi[0]] + xv[1][gi[1]])) < 1e-10
    assert g.evaluate([[0, 0]])[0] == pytest.approx(0.0, abs=1e-8) and g.evaluate([[3, 3]])[0] == pytest.approx(7.0, abs=1e-8)
    assert_same(g, ob.quanticscrossinterpolate_from_arrays(xv, add, None, oo))
    fs = lambda x: 0.1 * x * x - math.pi * x + 2.0
    n = 128
    grid = [-3.0 + 5.0 * i / (n - 1) for i in range(n)]
    go, oo = opts(t4a, tolerance=1e-8, seed=9)
    g = t4a.quanticscrossinterpolate_from_arrays([grid], lambda x: fs(x[0]), None, go)
    assert g.is_discretized() and g.history()[1][-1] < 1e-8
    assert np.abs(g.evaluate(np.arange(n).reshape(-1, 1)) - np.array([fs(x) for x in grid])).max() < 1e-6
    assert_same(g, ob.quanticscrossinterpolate_from_arrays([grid], lambda x: fs(x[0]), None, oo))


def test_unfolding_conventions_and_batched_callback(t4a):
    calls = []

    def f(idx):
        return 1.0 + idx[0] + 10.0 * idx[1]

    f.batched = lambda pts: (calls.append(len(pts)), 1.0 + pts[:, 0] + 10.0 * pts[:, 1])[1]
    g = t4a.quanticscrossinterpolate_discrete([8, 8], f, None, t4a.QtciOptions(tolerance=1e-10, n_random_init_pivot=0))
    assert g.local_dimensions() == [2] * 6
    assert g.grididx_to_quantics([5, 3]) == [1, 0, 0, 1, 1, 1] and g.quantics_to_grididx([1, 0, 0, 1, 1, 1]) == [5, 3]
    gf = t4a.quanticscrossinterpolate_discrete([8, 8], f, None,
                                               t4a.QtciOptions(tolerance=1e-10, n_random_init_pivot=0, unfolding_scheme=1))
    assert gf.local_dimensions() == [4] * 3
    assert gf.grididx_to_quantics([5, 3]) == [1, 2, 3] and gf.quantics_to_grididx([1, 2, 3]) == [5, 3]
    pts = np.array([[i, j] for i in range(8) for j in range(8)])
    exact = 1.0 + pts[:, 0] + 10.0 * pts[:, 1]
    assert np.allclose(g.evaluate(pts), exact, atol=1e-8) and np.allclose(gf.evaluate(pts), exact, atol=1e-8)
    # every distinct point reaches the user once; one user call per candidate matrix at most
    n_calls, n_points = gf.user_call_stats()
    assert n_points == len(gf.cachedata()) <= 64 and n_calls <= len(calls)


def test_two_variable_function_matches_oracle_at_scale(t4a):
    # 2 x 12 bits, interleaved: a smooth but non-trivial function; same pivots and cores as the oracle
    f = lambda x: math.exp(-3.0 * (x[0] - 0.3) ** 2 - 2.0 * (x[1] - 0.6) ** 2) * math.cos(9.0 * x[0] * x[1]) + 0.2 * x[0]
    f.batched = lambda p: np.exp(-3.0 * (p[:, 0] - 0.3) ** 2 - 2.0 * (p[:, 1] - 0.6) ** 2) * np.cos(9.0 * p[:, 0] * p[:, 1]) + 0.2 * p[:, 0]
    go, oo = opts(t4a, tolerance=1e-9, n_random_init_pivot=4, seed=12, max_iter=12)
    g = t4a.quanticscrossinterpolate([12, 12], f, [0.0, 0.0], [1.0, 1.0], options=go)
    fo = lambda x: float(f.batched(np.array([x]))[0])
    o = ob.quanticscrossinterpolate([12, 12], fo, [0.0, 0.0], [1.0, 1.0], options=oo)
    assert g.history()[0] == o.history()[0]
    for k in range(1, g.n_sites):
        assert g.tree_pivots(range(k)).tolist() == o.tree_pivots(range(k)).tolist()
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 4096, size=(500, 2))
    exact = f.batched(pts / 4096.0)
    got = g.evaluate(pts)
    assert np.abs(got - exact).max() < 1e-6
    assert np.abs(got - o.evaluate(pts)).max() < 1e-9
    assert g.integral() == pytest.approx(o.integral(), rel=1e-9)


# ------------------------------------------------------------------------------------------------ batched (vector valued)
def _bits(i, r):
    return [(i >> (r - 1 - b)) & 1 for b in range(r)]


def test_reference_batched_cases(t4a):
    # batched/tests/mod.rs:6-129, 158-211
    f2 = lambda x: [math.sin(2 * math.pi * x[0]) + 1.0, math.cos(2 * math.pi * x[0])]
    tt, ranks, errors, _ = t4a.quanticscrossinterpolate_batched([6], f2, [2], [0.0], [1.0],
                                                               options=t4a.QtciOptions(tolerance=1e-8, seed=1))
    assert len(tt) == 7 and len(ranks) == len(errors) > 0
    pts = np.array([_bits(i, 6) + [c] for i in range(64) for c in range(2)])
    exact = np.array([f2([i / 64.0])[c] for i in range(64) for c in range(2)])
    assert np.abs(tt.evaluate(pts) - exact).max() < 1e-6
    o = ob.quanticscrossinterpolate_batched([6], f2, [2], [0.0], [1.0], options=ob.QtciOptions(tolerance=1e-8, seed=1))
    assert ranks == o.history()[0]
    for a, b in zip(tt.site_tensors(), o.cores()):
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    f4 = lambda x: [(c + 1.0) * (x[0] + 1.0) for c in range(4)]
    tt, _, _, _ = t4a.quanticscrossinterpolate_batched([4], f4, [2, 2], [0.0], [1.0], options=t4a.QtciOptions(tolerance=1e-8, seed=2))
    assert len(tt) == 5 and tt.site_dims()[-1] == 4
    pts = np.array([_bits(i, 4) + [c] for i in range(16) for c in range(4)])
    exact = np.array([f4([i / 16.0])[c] for i in range(16) for c in range(4)])
    assert np.abs(tt.evaluate(pts) - exact).max() < 1e-8
    s = t4a.quanticscrossinterpolate([4], lambda x: x[0] * x[0], [0.0], [1.0], options=t4a.QtciOptions(tolerance=1e-8, seed=3))
    b, _, _, _ = t4a.quanticscrossinterpolate_batched([4], lambda x: [x[0] * x[0]], [1], [0.0], [1.0],
                                                     options=t4a.QtciOptions(tolerance=1e-8, seed=3))
    assert s.n_sites == 4 and len(b) == 5
    assert np.abs(s.evaluate(np.arange(16).reshape(-1, 1)) - b.evaluate(np.array([_bits(i, 4) + [0] for i in range(16)]))).max() < 1e-10


def test_reference_batched_errors_and_shared_cache(t4a):
    # batched/tests/mod.rs:131-156, 213-280
    for od in ([], [0]):
        with pytest.raises(t4a.T4aError) as e:
            t4a.quanticscrossinterpolate_batched([4], lambda x: [], od, [0.0], [1.0])
        assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError) as e:
        t4a.quanticscrossinterpolate_batched([2], lambda x: [1.0], [2], [0.0], [1.0])
    assert "expected at least 2" in str(e.value)
    calls = []

    def f(x):
        calls.append(x[0])
        return [x[0] + 1.0, x[0] * x[0] + 1.0]

    tt, _, _, user_points = t4a.quanticscrossinterpolate_batched([3], f, [2], [0.0], [1.0],
                                                                 options=t4a.QtciOptions(tolerance=1e-8, n_random_init_pivot=0))
    assert len(tt) == 4 and len(calls) == user_points <= 8 and len(set(calls)) == len(calls)


def test_three_variables_and_unequal_bits_match_oracle(t4a):
    # 3 variables fused (site dimension 8) and interleaved; unequal bits per variable through the grid builder arguments
    f3 = lambda x: math.cos(3.0 * x[0] + 2.0 * x[1] - x[2]) + 0.5 * x[0] * x[2] + 1.0
    for scheme in (0, 1):
        go, oo = opts(t4a, tolerance=1e-9, n_random_init_pivot=3, seed=21, max_iter=10)
        g = t4a.quanticscrossinterpolate([4, 4, 4], f3, [0.0] * 3, [1.0, 2.0, 0.5], grid_unfolding=scheme, options=go)
        o = ob.quanticscrossinterpolate([4, 4, 4], f3, [0.0] * 3, [1.0, 2.0, 0.5], grid_unfolding=scheme, options=oo)
        assert g.local_dimensions() == ([8] * 4 if scheme else [2] * 12)
        assert_same(g, o)
        rng = np.random.default_rng(scheme)
        pts = rng.integers(0, 16, size=(200, 3))
        exact = np.array([f3([p[0] / 16.0, p[1] / 8.0, p[2] / 32.0]) for p in pts])
        assert np.abs(g.evaluate(pts) - exact).max() < 1e-6
    f2 = lambda x: 1.0 / (1.0 + x[0] + 3.0 * x[1] * x[1])
    for scheme in (0, 1):
        go, oo = opts(t4a, tolerance=1e-10, n_random_init_pivot=2, seed=5, max_iter=10)
        g = t4a.quanticscrossinterpolate([5, 3], f2, [0.0, 0.0], [1.0, 1.0], include_endpoint=True, grid_unfolding=scheme, options=go)
        o = ob.quanticscrossinterpolate([5, 3], f2, [0.0, 0.0], [1.0, 1.0], include_endpoint=True, grid_unfolding=scheme, options=oo)
        assert g.local_dimensions() == o.local_dimensions() == ([4, 4, 4, 2, 2] if scheme else [2] * 8)
        assert_same(g, o)
        pts = np.array([[i, j] for i in range(32) for j in range(8)])
        exact = np.array([f2([p[0] / 31.0, p[1] / 7.0]) for p in pts])
        assert np.abs(g.evaluate(pts) - exact).max() < 1e-7
