"""GPU tests of the ACI path (crates/tensor4all-aci) through the C ABI: the crate's fixtures, and parity with
oracle/t4a_oracle_aci.hpp — frames and pivot errors bit-identical (the candidate matrices are accumulated in the same order on
both sides), solution values to 1e-10."""
import itertools

import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_aci import constant_tt, dense, lcg_tt, separable_tt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def grid(site_dims):
    return np.array(list(itertools.product(*[range(d) for d in site_dims])), dtype=np.uint32)


def tt_values(tt, site_dims):
    return np.asarray(tt.evaluate(grid(site_dims)))


def test_reference_fixtures(t4a):
    # lib.rs doc test, tests.rs:190-240, :267-322
    r = t4a.elementwise_batched(t4a.ACI_PRODUCT, [constant_tt([2, 3], 2.0), constant_tt([2, 3], 4.0)])
    assert np.abs(tt_values(r.tensor_train, [2, 3]) - 8.0).max() < 1e-10 and r.tensor_train.site_dims() == [2, 3]
    assert len(r.ranks) == len(r.errors) == len(r.nglobal_pivots)
    r = t4a.elementwise(lambda v: v[0] * v[1], [constant_tt([2, 3, 2], 2.0), constant_tt([2, 3, 2], 4.0)])
    assert np.abs(tt_values(r.tensor_train, [2, 3, 2]) - 8.0).max() < 1e-12
    a = t4a.elementwise_batched(t4a.ACI_SUM, [constant_tt([2, 2], 2.0), constant_tt([2, 2], 5.0)])
    b = t4a.elementwise_batched(lambda v: v[0] + v[1], [constant_tt([2, 2], 2.0), constant_tt([2, 2], 5.0)])
    assert np.abs(tt_values(a.tensor_train, [2, 2]) - 7.0).max() < 1e-12
    assert np.array_equal(tt_values(a.tensor_train, [2, 2]), tt_values(b.tensor_train, [2, 2]))
    seen = {}

    def op(v):
        seen["shape"] = v.shape
        return v[0] * 10 + v[1]
    r = t4a.elementwise_batched(op, [[np.array([1.0, 2.0, 3.0]).reshape(1, 3, 1)], [np.array([4.0, 5.0, 6.0]).reshape(1, 3, 1)]])
    assert seen["shape"] == (2, 3) and r.ranks == [] and r.termination == t4a.ACI_CONVERGED
    assert np.array_equal(tt_values(r.tensor_train, [3]), [14.0, 25.0, 36.0])


def test_frames_and_one_bond_updates(t4a):
    # tests.rs:594-603, :729-745, :834-854, :870-913, :935-983
    p = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, [constant_tt([2, 2, 2], 1.0), constant_tt([2, 2, 2], 2.0)])
    for k in range(2):
        assert p.frame(False, k, 0).shape == (1, 1) and p.frame(True, k, 3).shape == (1, 1)
        assert p.frame(True, k, 1).shape == (1, 1) and p.frame(True, k, 2).shape == (1, 1) and p.frame(True, k, 0) is None
    inp = [np.array([1.0, 2.0, 10.0, 20.0]).reshape((1, 2, 2), order="F"), np.array([3.0, 30.0, 4.0, 40.0]).reshape((2, 2, 1), order="F")]
    guess = [np.ones((1, 2, 2)), np.array([2.0, 0.0, 0.0, 1.0]).reshape((2, 2, 1), order="F")]
    p = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, [inp], t4a.AciOptions(initial_guess=guess))
    assert np.array_equal(p.frame(True, 0, 1), [[3.0, 4.0], [30.0, 40.0]])
    for left in (True, False):
        p = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, [constant_tt([2, 2], 2.0), constant_tt([2, 2], 3.0)])
        p.local_update(0, left)
        assert np.abs(tt_values(p.solution(), [2, 2]) - 6.0).max() < 1e-12
        assert p.frame(not left, 0, 1).shape == (1, 1) and p.errors()[0][0] <= 1e-12
    p = t4a.ElementwiseProblem(lambda v: np.zeros(v.shape[1]), [constant_tt([2, 2, 2], 1.0), constant_tt([2, 2, 2], 2.0)])
    p.local_update(0, True)
    assert p.frame(False, 0, 1).shape == (1, 1) and np.abs(tt_values(p.solution(), [2, 2, 2])).max() == 0.0


def test_global_pivot_injection_fixtures(t4a):
    # tests.rs:652-727 and the same sequence on the oracle: identical frames afterwards
    site_dims = [2] * 5
    ins = [separable_tt(site_dims, 0.25), separable_tt(site_dims, 0.5)]
    p, o = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, ins), ob.OracleAciProblem(ob.ACI_PRODUCT, ins)
    for b in range(4):
        p.local_update(b, True)
        o.local_update(b, True)
    before = p.solution().link_dims()
    pivots = [[(k >> s) & 1 for s in range(5)] for k in range(8)]
    assert p.add_global_pivots(pivots) == o.add_global_pivots(pivots)
    after = p.solution().link_dims()
    assert after == o.solution().link_dims()
    assert all(d <= b for d, b in zip(after, [2, 4, 4, 2])) and any(a > b for a, b in zip(after, before))
    for bond in range(1, 5):
        for k in range(2):
            for right in (False, True):
                fo, fd = o.frame(right, k, bond), p.frame(right, k, bond)
                assert (fo is None) == (fd is None) and (fo is None or np.array_equal(fo, fd))
    site_dims = [2] * 6
    p = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, [separable_tt(site_dims, 0.25), separable_tt(site_dims, 0.5)])
    for b in range(5):
        p.local_update(b, True)
    assert p.add_global_pivots([[1, 0, 1, 0, 1, 0]]) == 1
    dims = p.solution().link_dims()
    assert p.add_global_pivots([[1, 0, 1, 0, 1, 0]]) == 0 and p.solution().link_dims() == dims


@pytest.mark.parametrize("op", ["product", "sum", "callback"])
def test_sweeps_match_the_oracle_step_by_step(t4a, op):
    site_dims, link = [2, 3, 2, 4, 2, 3, 2], [2, 4, 5, 5, 4, 2]
    ins = [lcg_tt(site_dims, link, 11), lcg_tt(site_dims, link, 23), lcg_tt(site_dims, link, 37)][: 2 if op != "sum" else 3]
    rng = np.random.default_rng(5)
    guess = [rng.standard_normal(c.shape) for c in lcg_tt(site_dims, [2, 6, 9, 9, 6, 2], 1)]
    fn = (lambda v: np.tanh(v[0]) + v[0] * v[1]) if op == "callback" else None
    dop = fn if fn else (t4a.ACI_PRODUCT if op == "product" else t4a.ACI_SUM)
    oop = fn if fn else (ob.ACI_PRODUCT if op == "product" else ob.ACI_SUM)
    p = t4a.ElementwiseProblem(dop, ins, t4a.AciOptions(initial_guess=guess, tolerance=1e-11))
    o = ob.OracleAciProblem(oop, ins, ob.AciOptions(initial_guess=guess, tolerance=1e-11))
    n = len(site_dims)

    def compare():
        assert p.solution().link_dims() == o.solution().link_dims()
        for k in range(len(ins)):
            for s in range(n + 1):
                for right in (False, True):
                    fo, fd = o.frame(right, k, s), p.frame(right, k, s)
                    assert (fo is None) == (fd is None)
                    if fo is not None:
                        assert np.array_equal(fo, fd), (k, s, right)
        (ed, sd), (eo, so) = p.errors(), o.errors()
        assert np.array_equal(ed, eo) and np.array_equal(sd, so)
    compare()  # initialize_right_frames
    for sweep in range(3):
        order = range(n - 1) if sweep % 2 == 0 else reversed(range(n - 1))
        for b in order:
            p.local_update(b, sweep % 2 == 0)
            o.local_update(b, sweep % 2 == 0)
        compare()
    vd, vo = tt_values(p.solution(), site_dims), np.asarray(o.solution().evaluate(grid(site_dims)))
    assert np.abs(vd - vo).max() < 1e-10 * np.abs(vo).max()
    dn = [dense(t) for t in ins]
    exact = fn(np.stack([d.ravel() for d in dn])) if fn else (dn[0] * dn[1] if op == "product" else dn[0] + dn[1] + dn[2]).ravel()
    assert np.abs(vd - exact).max() < 1e-8 * np.abs(exact).max()


@pytest.mark.parametrize("guard", [False, True])
def test_full_runs_match_the_oracle(t4a, guard):
    site_dims, link = [2, 3, 2, 3, 2, 2], [2, 3, 3, 3, 2]
    a, b = lcg_tt(site_dims, link, 7), lcg_tt(site_dims, link, 99)
    rng = np.random.default_rng(3)
    guess = [rng.standard_normal(c.shape) for c in lcg_tt(site_dims, [2, 4, 6, 4, 2], 1)]
    kw = dict(initial_guess=guess, enable_global_guard=guard, tolerance=1e-12)
    rd = t4a.elementwise_batched(t4a.ACI_PRODUCT, [a, b], t4a.AciOptions(**kw))
    ro = ob.aci_elementwise(ob.ACI_PRODUCT, [a, b], ob.AciOptions(**kw))
    assert rd.ranks == ro.ranks and rd.nglobal_pivots == ro.nglobal_pivots and rd.termination == ro.termination == 0
    assert np.array_equal(rd.errors, ro.errors)
    exact = (dense(a) * dense(b)).ravel()
    vd = tt_values(rd.tensor_train, site_dims)
    assert np.abs(vd - exact).max() < 1e-10 * np.abs(exact).max()
    assert np.abs(vd - np.asarray(ro.tensor_train.evaluate(grid(site_dims)))).max() < 1e-10 * np.abs(exact).max()
    # default random guess (same splitmix64 / Box-Muller stream on both sides)
    rd = t4a.elementwise_batched(t4a.ACI_PRODUCT, [a, b], t4a.AciOptions(rng_seed=5, enable_global_guard=guard))
    ro = ob.aci_elementwise(ob.ACI_PRODUCT, [a, b], ob.AciOptions(rng_seed=5, enable_global_guard=guard))
    assert rd.ranks == ro.ranks and rd.termination == ro.termination
    assert np.abs(tt_values(rd.tensor_train, site_dims) - exact).max() < 1e-9 * np.abs(exact).max()


def test_capped_run_and_larger_product(t4a):
    # tests.rs:452-499, then a chi = 16 x 16 product on 12 sites against the dense answer
    site_dims, link = [2] * 10, [2, 4, 4, 4, 4, 4, 4, 4, 2]
    ins = [lcg_tt(site_dims, link, 12345), lcg_tt(site_dims, link, 98765)]
    o = dict(max_iters=20, min_iters=2, max_bond_dim=4, tolerance=1e-10)
    r = t4a.elementwise_batched(t4a.ACI_PRODUCT, ins, t4a.AciOptions(**o))
    assert r.errors[-1] > 1e-10 and len(r.ranks) <= 4 and r.ranks[-2:] == [4, 4] and r.termination == t4a.ACI_RANK_LIMITED
    assert all(d <= 4 for d in r.tensor_train.link_dims())
    ro = ob.aci_elementwise(ob.ACI_PRODUCT, ins, ob.AciOptions(**o))
    assert r.ranks == ro.ranks and np.array_equal(r.errors, ro.errors)
    site_dims = [2] * 12
    link = [2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2]
    ins = [lcg_tt(site_dims, link, 5), lcg_tt(site_dims, link, 6)]
    r = t4a.elementwise_batched(t4a.ACI_PRODUCT, ins, t4a.AciOptions(tolerance=1e-12))
    exact = (dense(ins[0]) * dense(ins[1])).ravel()
    assert np.abs(tt_values(r.tensor_train, site_dims) - exact).max() < 1e-9 * np.abs(exact).max()
    assert max(r.tensor_train.link_dims()) == 16 and r.termination == t4a.ACI_CONVERGED


def test_errors(t4a):
    good = constant_tt([2, 2], 1.0)
    for kw in (dict(max_iters=0), dict(min_iters=0), dict(min_iters=5, max_iters=3), dict(tolerance=-1.0), dict(tolerance=float("nan")),
               dict(tol_margin_global_search=-1.0)):
        with pytest.raises(t4a.T4aError):
            t4a.elementwise_batched(t4a.ACI_PRODUCT, [good], t4a.AciOptions(**kw))
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(t4a.ACI_PRODUCT, [])
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(t4a.ACI_PRODUCT, [good, constant_tt([2, 2, 2], 1.0)])
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(t4a.ACI_PRODUCT, [good, constant_tt([2, 3], 1.0)])
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(t4a.ACI_PRODUCT, [good], t4a.AciOptions(initial_guess=constant_tt([2, 3], 1.0)))
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(t4a.ACI_PRODUCT, [good], t4a.AciOptions(initial_guess=[np.ones((1, 2, 2)), np.ones((2, 2, 1))], max_bond_dim=1))
    with pytest.raises(t4a.T4aError) as e:  # tests.rs:242-265: operator errors stop the sweep
        t4a.elementwise_batched(lambda v: 1 / 0, [good, good])
    assert e.value.code == t4a.CALLBACK_ERROR
    with pytest.raises(t4a.T4aError):
        t4a.elementwise_batched(7, [good])


def test_global_guard_search_and_injection_match_the_oracle(t4a):
    # a rank-capped first sweep leaves large errors: the guard's walks (same splitmix64 starts on both sides) must return the
    # same pivots, and sweeping on after the injection must keep the frames identical
    site_dims, link = [2, 3, 2, 3, 2, 3], [2, 4, 5, 4, 2]
    ins = [lcg_tt(site_dims, link, 3), lcg_tt(site_dims, link, 4)]
    guess = [np.ones((1, d, 1)) for d in site_dims]
    kw = dict(initial_guess=guess, tolerance=1e-12, nsearch_global_pivots=6, max_nglobal_pivot=4)
    p = t4a.ElementwiseProblem(t4a.ACI_PRODUCT, ins, t4a.AciOptions(**kw))
    o = ob.OracleAciProblem(ob.ACI_PRODUCT, ins, ob.AciOptions(**kw))
    n = len(site_dims)
    for b in range(n - 1):
        p.local_update(b, True)
        o.local_update(b, True)
    found = 0
    for seed in (1, 2, 3):
        pd, po = p.find_global_pivots(seed), o.find_global_pivots(seed, 4)
        assert pd == po
        found += len(pd)
        if pd:
            assert p.add_global_pivots(pd) == o.add_global_pivots(po)
        for sweep in (1, 2):
            order = reversed(range(n - 1)) if sweep % 2 else range(n - 1)
            for b in order:
                p.local_update(b, sweep % 2 == 0)
                o.local_update(b, sweep % 2 == 0)
        assert p.solution().link_dims() == o.solution().link_dims()
        for k in range(2):
            for s in range(n + 1):
                for right in (False, True):
                    fo, fd = o.frame(right, k, s), p.frame(right, k, s)
                    assert (fo is None) == (fd is None) and (fo is None or np.array_equal(fo, fd))
    assert found > 0
    exact = (dense(ins[0]) * dense(ins[1])).ravel()
    assert np.abs(tt_values(p.solution(), site_dims) - exact).max() < 1e-9 * np.abs(exact).max()


def test_random_shapes_match_the_oracle(t4a):
    rng = np.random.default_rng(2024)
    for case in range(10):
        n = int(rng.integers(2, 8))
        site_dims = [int(x) for x in rng.integers(2, 5, size=n)]
        link = [int(min(x, np.prod(site_dims[:b + 1]), np.prod(site_dims[b + 1:]))) for b, x in enumerate(rng.integers(1, 5, size=n - 1))]
        K = int(rng.integers(1, 4))
        ins = [lcg_tt(site_dims, link, 100 * case + k + 1) for k in range(K)]
        glink = [int(min(x, np.prod(site_dims[:b + 1]), np.prod(site_dims[b + 1:]))) for b, x in enumerate(rng.integers(1, 7, size=n - 1))]
        guess = [rng.standard_normal(c.shape) for c in lcg_tt(site_dims, glink, 1)]
        op_d, op_o = (t4a.ACI_PRODUCT, ob.ACI_PRODUCT) if case % 2 == 0 else (t4a.ACI_SUM, ob.ACI_SUM)
        kw = dict(initial_guess=guess, tolerance=1e-12, enable_global_guard=bool(case % 3 == 0), max_iters=12)
        rd = t4a.elementwise_batched(op_d, ins, t4a.AciOptions(**kw))
        ro = ob.aci_elementwise(op_o, ins, ob.AciOptions(**kw))
        assert rd.ranks == ro.ranks and rd.nglobal_pivots == ro.nglobal_pivots and rd.termination == ro.termination, case
        assert np.array_equal(rd.errors, ro.errors), case
        dn = [dense(t) for t in ins]
        exact = (np.prod(dn, axis=0) if case % 2 == 0 else np.sum(dn, axis=0)).ravel()
        vd = tt_values(rd.tensor_train, site_dims)
        assert np.abs(vd - np.asarray(ro.tensor_train.evaluate(grid(site_dims)))).max() <= 1e-10 * max(np.abs(exact).max(), 1e-300), case
        if rd.termination == 0:
            assert np.abs(vd - exact).max() < 1e-8 * np.abs(exact).max(), case
