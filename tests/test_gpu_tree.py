"""GPU tests of the tree TCI driver through the C ABI (t4a_gpu_treetci_*): the reference's own test
cases (crates/tensor4all-treetci/src/*/tests.rs) and parity with the CPU oracle — pivot tables bit-exact, errors and
site tensors to 1e-10."""
import itertools

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import OracleTreeTCI2, TreeOptions
from test_oracle_tree import SAMPLE_EDGES, _branched_fn, identity2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def gopts(t4a, o):
    return t4a.TreeTciOptions(tolerance=o.tolerance, max_iter=o.max_iter, max_bond_dim=o.max_bond_dim,
                              normalize_error=o.normalize_error, enable_global_pivots=o.enable_global_pivots,
                              nsearch=o.nsearch, max_nglobal_pivot=o.max_nglobal_pivot,
                              tol_margin_global_search=o.tol_margin_global_search, seed=o.seed)


def all_keys(tree, edges):
    keys = []
    for (u, v) in edges:
        l, r = tree.subregion_vertices(u, v)
        keys += [tuple(l), tuple(r)]
    return keys


def assert_same_state(g, o, edges, tol=1e-10):
    for key in all_keys(o, edges):
        assert g.pivots(key).tolist() == o.pivots(key).tolist(), f"pivot table of subtree {key}"
    assert g.max_sample_value() == o.max_sample_value()
    scale = max(1.0, o.max_sample_value())
    assert np.abs(g.bond_errors() - o.bond_errors()).max() <= tol * scale
    ge, oe = g.pivot_errors(), o.pivot_errors()
    assert len(ge) == len(oe) and (len(oe) == 0 or np.abs(ge - oe).max() <= tol * scale)


# ------------------------------------------------------------------------------------------------ reference fixtures
def test_reference_state_and_proposer_fixtures(t4a):
    # state/tests.rs:28-50, proposer/tests.rs:40-76, graph/tests.rs:21-40
    t = t4a.TreeTCI2([2] * 7, SAMPLE_EDGES)
    t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
    assert t.subregion_vertices(1, 3) == ([0, 1, 2], [3, 4, 5, 6])
    assert t.pivots([0, 1, 2]).tolist() == [[0, 0, 0], [1, 0, 1]]
    assert t.pivots([3, 4, 5, 6]).tolist() == [[0, 0, 0, 0], [0, 1, 0, 1]]
    assert t.pivots(range(7)).shape == (0, 7)
    iset, jset = t.candidates(1, 3)
    assert iset.tolist() == [[0, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1]]
    assert jset.tolist() == [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 1], [1, 1, 0, 1]]


def test_reference_update_edge_identity(t4a):
    # update/tests.rs:22-72
    t = t4a.TreeTCI2([2, 2], [(0, 1)])
    t.set_function(identity2)
    t.add_global_pivots([[0, 0]])
    t.flush_pivot_errors()
    sel = t.update_edge(0, 1, rel_tol=0.0, abs_tol=0.0)
    assert sel["rank"] == 2
    assert t.pivots([0]).tolist() == [[0], [1]] and t.pivots([1]).tolist() == [[0], [1]]
    assert abs(t.max_sample_value() - 1.0) < 1e-12 and abs(t.max_bond_error()) < 1e-12
    assert abs(t.pivot_errors()[-1]) < 1e-12


def test_reference_update_edge_rejects_bad_batch_length(t4a):
    # update/tests.rs:74-89
    t = t4a.TreeTCI2([2, 2], [(0, 1)])
    f = lambda idx: 1.0
    f.batched = lambda idx: np.ones(1)
    t.set_function(f)
    t.add_global_pivots([[0, 0]])
    with pytest.raises(t4a.T4aError) as e:
        t.update_edge(0, 1, rel_tol=0.0, abs_tol=0.0)
    assert e.value.code == t4a.CALLBACK_ERROR


def test_reference_candidate_matrix_order(t4a):
    # update/tests.rs:91-135: the callback sees a column-major (n_sites, n_points) batch, right candidates slowest
    seen = []

    def code(idx):
        return float(sum(int(v) * 10 ** (3 - k) for k, v in enumerate(idx)))

    code.batched = lambda idx: (seen.append(np.array(idx)), np.array([code(r) for r in idx]))[1]
    t = t4a.TreeTCI2([4, 4, 4, 4], [(0, 1), (1, 2), (2, 3)])
    t.set_function(code)
    t.add_global_pivots([[0, 0, 2, 3], [1, 1, 1, 0]])
    lc, rc = t.candidates(1, 2)
    t.update_edge(1, 2, rel_tol=0.0, abs_tol=0.0)
    batch = seen[-1]
    expect = np.array([list(l) + list(r) for r in rc for l in lc])
    assert batch.tolist() == expect.tolist()


def test_reference_optimize_options_and_early_stop(t4a):
    # optimize/tests.rs:10-222
    calls = []

    def counted(idx):
        calls.append(1)
        return identity2(idx)

    for tol in (-1.0, float("nan"), float("inf"), float("-inf")):
        t = t4a.TreeTCI2([2, 2], [(0, 1)])
        t.set_function(counted)
        t.add_global_pivots([[0, 0]])
        with pytest.raises(t4a.T4aError) as e:
            t.optimize(t4a.TreeTciOptions(tolerance=tol))
        assert e.value.code == t4a.INVALID_ARGUMENT and not calls
    for kw in ({"max_iter": 0}, {"tol_margin_global_search": -1.0}, {"tol_margin_global_search": float("nan")}):
        t = t4a.TreeTCI2([2, 2], [(0, 1)])
        t.set_function(identity2)
        t.add_global_pivots([[0, 0]])
        with pytest.raises(t4a.T4aError):
            t.optimize(t4a.TreeTciOptions(**kw))
    t = t4a.TreeTCI2([2, 2], [(0, 1)])
    t.set_function(identity2)
    t.add_global_pivots([[0, 0]])
    ranks, errors = t.optimize(t4a.TreeTciOptions(tolerance=1e-12, max_iter=4))
    assert ranks[-1] == 2 and t.max_bond_dim() == 2 and abs(errors[-1]) <= 1e-12
    assert len(ranks) < 4 and len(ranks) == len(errors)
    t.materialize(0)  # materialize/tests.rs:17-77
    assert np.allclose(t.evaluate([[0, 0], [0, 1], [1, 0], [1, 1]]), [1.0, 0.0, 0.0, 1.0], atol=1e-12)
    t = t4a.TreeTCI2([3, 3], [(0, 1)])
    t.set_function(identity2)
    t.add_global_pivots([[0, 0]])
    ranks, errors = t.optimize(t4a.TreeTciOptions(tolerance=1e-12, max_iter=10, max_bond_dim=1))
    assert len(ranks) < 10 and all(r <= 1 for r in ranks) and errors[-1] > 1e-12


def test_reference_zero_pivot_matrix_gives_zero_core(t4a):
    # materialize/tests.rs:160-176
    t = t4a.TreeTCI2([2, 2], [(0, 1)])
    t.set_function(lambda idx: 0.0)
    t.add_global_pivots([[0, 0]])
    t.materialize(0)
    assert np.all(t.site_tensor(1) == 0.0)
    assert np.all(t.evaluate([[0, 0], [0, 1], [1, 0], [1, 1]]) == 0.0)
    with pytest.raises(t4a.T4aError):  # api.rs:82-90
        t4a.tree_crossinterpolate2(lambda idx: 0.0, [2, 2], [(0, 1)], [[0, 0]], t4a.TreeTciOptions())


def test_graph_and_state_errors(t4a):
    for dims, edges in (([2] * 3, [(0, 1)]), ([2] * 3, [(0, 1), (1, 1)]), ([2] * 3, [(0, 1), (0, 1)]),
                        ([2] * 4, [(0, 1), (1, 2), (0, 2)]), ([2] * 6, SAMPLE_EDGES), ([2, 0], [(0, 1)])):
        with pytest.raises(t4a.T4aError) as e:
            t4a.TreeTCI2(dims, edges)
        assert e.value.code == t4a.INVALID_ARGUMENT
    t = t4a.TreeTCI2([2, 2, 2], [(0, 1), (1, 2)])
    with pytest.raises(t4a.T4aError):
        t.add_global_pivots([[0, 0, 2]])
    with pytest.raises(t4a.T4aError):
        t.update_edge(0, 1)  # no function
    t.set_function(lambda idx: 1.0)
    with pytest.raises(t4a.T4aError):
        t.update_edge(0, 2)  # not an edge
    with pytest.raises(t4a.T4aError):
        t.evaluate([[0, 0, 0]])  # nothing materialised
    with pytest.raises(t4a.T4aError):
        t.update_edge(0, 1)  # missing pivot tables


# ------------------------------------------------------------------------------------------------ parity with the oracle
def _pair(t4a, dims, edges, f):
    g = t4a.TreeTCI2(dims, edges)
    g.set_function(f)
    return g, OracleTreeTCI2(dims, edges, f)


@pytest.mark.parametrize("center", [0, 3, 6])
def test_branched_tree_callback_parity(t4a, center):
    dims = [3, 2, 3, 2, 2, 3, 2]
    g, o = _pair(t4a, dims, SAMPLE_EDGES, _branched_fn)
    opt = TreeOptions(tolerance=1e-10, max_iter=8, seed=3)
    og = o.crossinterpolate2([[0] * 7], opt)
    gg = g.crossinterpolate2([[0] * 7], gopts(t4a, opt))
    assert gg[0] == og[0] and np.allclose(gg[1], og[1], rtol=0, atol=1e-12)
    assert_same_state(g, o, SAMPLE_EDGES)
    g.materialize(center)
    o.materialize(center)
    for s in range(7):
        a, b = g.site_tensor(s), o.site_tensor(s)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), f"site {s}"
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    exact = np.array([_branched_fn(p) for p in pts])
    got = g.evaluate(pts)
    assert np.abs(got - o.evaluate(pts)).max() <= 1e-10
    assert np.abs(got - exact).max() < 1e-8


def test_builtin_functor_on_quantics_tree_parity(t4a):
    # the device functor (accumulators only travel, Pi is built inside the rrLU kernel) on a comb-like tree of 12 bits
    from t4a_amd.functions import quantics_trig_exp
    n = 12
    f = quantics_trig_exp(n)
    edges = [(i, i + 1) for i in range(0, 5)] + [(2, 6), (6, 7), (7, 8)] + [(4, 9), (9, 10), (10, 11)]
    g, o = _pair(t4a, [2] * n, edges, f)
    opt = TreeOptions(tolerance=1e-9, max_iter=6, enable_global_pivots=False)
    og = o.crossinterpolate2([[0] * n], opt)
    gg = g.crossinterpolate2([[0] * n], gopts(t4a, opt))
    assert gg[0] == og[0] and np.allclose(gg[1], og[1], rtol=0, atol=1e-12)
    assert_same_state(g, o, edges)
    g.materialize(2)
    o.materialize(2)
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 2, size=(400, n))
    exact = ob.fn_eval(f, pts) if hasattr(ob, "fn_eval") else None
    got = g.evaluate(pts)
    assert np.abs(got - o.evaluate(pts)).max() <= 1e-10
    if exact is not None:
        assert np.abs(got - exact).max() < 1e-6


def test_global_pivot_search_parity(t4a):
    # two separated spikes that the local updates miss: the global search has to find them (globalpivot/tests.rs)
    dims = [4] * 5
    edges = [(0, 1), (1, 2), (2, 3), (2, 4)]

    def f(idx):
        x = list(idx)
        v = 1.0 / (1.0 + sum(x))
        if x == [3, 3, 3, 3, 3]:
            v += 5.0
        if x == [3, 0, 2, 1, 3]:
            v -= 4.0
        return v

    g, o = _pair(t4a, dims, edges, f)
    for t in (g, o):
        t.add_global_pivots([[0] * 5])
        t.set_max_sample_value(1.0)
    o.optimize(TreeOptions(tolerance=1e-9, max_iter=2, enable_global_pivots=False))
    g.optimize(gopts(t4a, TreeOptions(tolerance=1e-9, max_iter=2, enable_global_pivots=False)))
    assert_same_state(g, o, edges)
    po = o.find_global_pivots(40, 6, 10.0, 1e-9, 11)
    pg = g.find_global_pivots(40, 6, 10.0, 1e-9, 11)
    assert pg.tolist() == po.tolist()
    opt = TreeOptions(tolerance=1e-9, max_iter=10, nsearch=40, max_nglobal_pivot=6, seed=5)
    og = o.optimize(opt)
    gg = g.optimize(gopts(t4a, opt))
    assert gg[0] == og[0] and np.allclose(gg[1], og[1], rtol=0, atol=1e-12)
    assert_same_state(g, o, edges)
    g.materialize(2)
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    assert np.abs(g.evaluate(pts) - np.array([f(p) for p in pts])).max() < 1e-7


def test_chain_graph_agrees_with_tensorci2_accuracy(t4a):
    # a linear chain is a tree: same function, chain TCI2 vs tree TCI reach the same accuracy (bond dims may differ)
    from t4a_amd.functions import quantics_trig_exp
    n = 10
    f = quantics_trig_exp(n)
    edges = [(i, i + 1) for i in range(n - 1)]
    t, ranks, errors = t4a.tree_crossinterpolate2(f, [2] * n, edges, [[0] * n],
                                                  t4a.TreeTciOptions(tolerance=1e-10, max_iter=8, seed=1), center_site=4)
    assert errors[-1] < 1e-10
    pts = np.array(list(itertools.product(*[range(2)] * n)))
    exact = ob.fn_eval(f, pts)
    assert np.abs(t.evaluate(pts) - exact).max() < 1e-8
    assert len(t.bond_errors()) == n - 1


def test_star_graph_with_many_incoming_bonds(t4a):
    # hub with 5 leaves: the hub tensor has 5 incoming bonds when a leaf is the root, 4 + parent otherwise
    dims = [3, 2, 2, 2, 2, 2]
    edges = [(0, k) for k in range(1, 6)]
    f = lambda idx: float(np.cos(0.7 * idx[0] + sum((k + 1) * 0.3 * idx[k] for k in range(1, 6))))
    g, o = _pair(t4a, dims, edges, f)
    opt = TreeOptions(tolerance=1e-10, max_iter=6, enable_global_pivots=False)
    og = o.crossinterpolate2([[0] * 6], opt)
    gg = g.crossinterpolate2([[0] * 6], gopts(t4a, opt))
    assert gg[0] == og[0]
    assert_same_state(g, o, edges)
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    exact = np.array([f(p) for p in pts])
    for center in (0, 3):
        g.materialize(center)
        o.materialize(center)
        assert g.site_tensor(0).ndim == (6 if center == 0 else 6)
        assert np.abs(g.evaluate(pts) - o.evaluate(pts)).max() <= 1e-10
        assert np.abs(g.evaluate(pts) - exact).max() < 1e-8


@pytest.mark.parametrize("kind", [1, 2])
def test_random_proposers_match_oracle(t4a, kind):
    # SimpleProposer / TruncatedDefaultProposer on the same splitmix64 stream as the oracle (proposer/tests.rs:117-166)
    g = t4a.TreeTCI2([2] * 7, SAMPLE_EDGES)
    o = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    for t in (g, o):
        t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
        t.set_proposer(kind, 7)
    for (u, v) in SAMPLE_EDGES:
        gi, gj = g.candidates(u, v)
        oi, oj = o.candidates(u, v)
        assert gi.tolist() == oi.tolist() and gj.tolist() == oj.tolist()
    dims = [3, 2, 3, 2, 2, 3, 2]
    g, o = _pair(t4a, dims, SAMPLE_EDGES, _branched_fn)
    for t in (g, o):
        t.set_proposer(kind, 5)
    opt = TreeOptions(tolerance=1e-10, max_iter=12, seed=3)
    og = o.crossinterpolate2([[0] * 7], opt)
    gg = g.crossinterpolate2([[0] * 7], gopts(t4a, opt))
    assert gg[0] == og[0] and np.allclose(gg[1], og[1], rtol=0, atol=1e-12)
    assert_same_state(g, o, SAMPLE_EDGES)
    with pytest.raises(t4a.T4aError):
        g.set_proposer(3)


def test_truncated_proposer_keeps_hub_matrices_small(t4a):
    # star with 4 arms of 3 binary sites: the hub's default candidate count is d * chi^3, the truncated one d * chi
    edges, n = [], 13
    for arm in range(4):
        first = 1 + 3 * arm
        edges += [(0, first), (first, first + 1), (first + 1, first + 2)]
    f = lambda idx: float(np.cos(0.4 * sum((k % 5 + 1) * v for k, v in enumerate(idx))) + 0.05 * idx[1] * idx[12])
    g = t4a.TreeTCI2([2] * n, edges)
    g.set_function(f)
    g.set_proposer(2, 1)
    ranks, errors = g.crossinterpolate2([[0] * n], t4a.TreeTciOptions(tolerance=1e-8, max_iter=10, seed=2))
    li, ri = g.candidates(0, 1)
    chi = len(g.pivots(g.subregion_vertices(0, 1)[0]))
    assert len(li) <= 2 * chi
    g.materialize(0)
    rng = np.random.default_rng(3)
    pts = rng.integers(0, 2, size=(300, n))
    assert np.abs(g.evaluate(pts) - np.array([f(p) for p in pts])).max() < 1e-5


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_trees_and_options_match_oracle(t4a, seed):
    """Random trees (Pruefer-free construction: every new site attaches to a random earlier one), random local dimensions,
    random options and proposers: pivot tables, histories and materialised tensors equal the oracle's."""
    rng = np.random.default_rng(900 + seed)
    for case in range(4):
        n = int(rng.integers(3, 9))
        dims = [int(rng.integers(2, 4)) for _ in range(n)]
        edges = [(int(rng.integers(0, k)), k) for k in range(1, n)]
        w = rng.standard_normal(n)
        c = rng.standard_normal((n, n)) * 0.15
        kind = int(rng.integers(0, 3))

        def f(idx, w=w, c=c, kind=kind):
            x = np.asarray(idx, dtype=np.float64)
            if kind == 0:
                return float(np.cos(w @ x) + 0.3)
            if kind == 1:
                return float(1.0 / (1.5 + np.abs(w) @ x + x @ np.abs(c) @ x))
            return float(np.exp(-0.2 * (w @ x) ** 2) + 0.1 * x[0] * x[-1])

        g, o = _pair(t4a, dims, edges, f)
        prop = int(rng.choice([0, 0, 1, 2]))
        for t in (g, o):
            t.set_proposer(prop, 11 + case)
        opt = TreeOptions(tolerance=float(10.0 ** rng.integers(-10, -3)), max_iter=int(rng.integers(2, 7)),
                          max_bond_dim=None if rng.random() < 0.5 else int(rng.integers(1, 6)),
                          normalize_error=bool(rng.integers(0, 2)), enable_global_pivots=bool(rng.integers(0, 2)),
                          nsearch=int(rng.integers(1, 6)), max_nglobal_pivot=int(rng.integers(1, 4)), seed=int(rng.integers(0, 100)))
        ctx = f"seed {seed} case {case}: dims {dims} edges {edges} kind {kind} proposer {prop} opts {vars(opt)}"
        first = [int(rng.integers(0, d)) for d in dims]
        if f(first) == 0.0:
            first = [0] * n
        og = o.crossinterpolate2([first], opt)
        gg = g.crossinterpolate2([first], gopts(t4a, opt))
        assert gg[0] == og[0], ctx
        assert np.allclose(gg[1], og[1], rtol=0, atol=1e-11), ctx
        assert_same_state(g, o, edges)
        center = int(rng.integers(0, n))
        try:
            o.materialize(center)
        except ob.OracleError:
            # e.g. a bond-dimension-saturated stop right after global pivots were injected: the two sides of an edge hold
            # different pivot counts and to_treetn refuses (materialize.rs:40-48) — on both sides
            with pytest.raises(t4a.T4aError):
                g.materialize(center)
            continue
        g.materialize(center)
        pts = np.array(list(itertools.product(*[range(d) for d in dims])))
        ge, oe = g.evaluate(pts), o.evaluate(pts)
        assert np.abs(ge - oe).max() <= 1e-9 * max(1.0, np.abs(oe).max()), ctx
