"""oracle/t4a_oracle_aci.hpp against the fixtures of crates/tensor4all-aci/src/tests.rs and the crate's doc tests."""
import itertools

import numpy as np
import pytest

import oracle_binding as ob


def constant_tt(site_dims, value):  # SimpleTensorTrain::constant: first core carries the value, the others are ones
    cores = [np.ones((1, d, 1)) for d in site_dims]
    cores[0] = cores[0] * value
    return cores


def lcg_tt(site_dims, link_dims, seed):  # tests.rs:44-71 tensor_train_with_values
    state = [seed | 1]

    def nxt():
        state[0] = (state[0] * 6364136223846793005 + 1) % (1 << 64)
        return (state[0] >> 33) / float(1 << 31) - 0.5
    cores = []
    for s, d in enumerate(site_dims):
        l = 1 if s == 0 else link_dims[s - 1]
        r = link_dims[s] if s < len(link_dims) else 1
        cores.append(np.array([nxt() for _ in range(l * d * r)]).reshape((l, d, r), order="F"))
    return cores


def separable_tt(site_dims, seed):  # tests.rs:609-621
    return [np.array([1.0 + i + seed * s for i in range(d)]).reshape(1, d, 1) for s, d in enumerate(site_dims)]


def dense(cores):
    t = cores[0]
    for c in cores[1:]:
        t = np.tensordot(t, c, axes=([-1], [0]))
    return t.reshape(t.shape[1:-1])


def test_doc_examples_and_constant_products():
    # lib.rs doc test, tests.rs:190-210, :212-240
    r = ob.aci_elementwise(ob.ACI_PRODUCT, [constant_tt([2, 3], 2.0), constant_tt([2, 3], 4.0)])
    assert abs(r.tensor_train.evaluate([[1, 2]])[0] - 8.0) < 1e-10
    assert [int(d[1]) for d in r.tensor_train.dims()] == [2, 3] and len(r.ranks) == len(r.errors)
    r = ob.aci_elementwise(lambda v: v[0] * v[1], [constant_tt([2, 3, 2], 2.0), constant_tt([2, 3, 2], 4.0)])
    pts = list(itertools.product(range(2), range(3), range(2)))
    assert np.abs(r.tensor_train.evaluate(pts) - 8.0).max() < 1e-12
    a = ob.aci_elementwise(ob.ACI_SUM, [constant_tt([2, 2], 2.0), constant_tt([2, 2], 5.0)])
    b = ob.aci_elementwise(lambda v: v[0] + v[1], [constant_tt([2, 2], 2.0), constant_tt([2, 2], 5.0)])
    pts = list(itertools.product(range(2), range(2)))
    assert np.abs(a.tensor_train.evaluate(pts) - 7.0).max() < 1e-12
    assert np.array_equal(a.tensor_train.evaluate(pts), b.tensor_train.evaluate(pts))


def test_single_site_uses_column_major_batch_layout():
    # tests.rs:267-322: one site -> the operator is evaluated once over all site points, empty histories
    seen = {}

    def op(v):
        seen["shape"] = v.shape
        return v[0] * 10 + v[1]
    r = ob.aci_elementwise(op, [[np.array([1.0, 2.0, 3.0]).reshape(1, 3, 1)], [np.array([4.0, 5.0, 6.0]).reshape(1, 3, 1)]])
    assert seen["shape"] == (2, 3) and r.ranks == [] and r.errors == [] and r.termination == 0
    assert np.array_equal(r.tensor_train.evaluate([[0], [1], [2]]), [14.0, 25.0, 36.0])


def test_frames_of_the_reference_fixtures():
    # tests.rs:594-603 boundary frames, :729-745 rank-one right frames, :834-854 right frame values
    p = ob.OracleAciProblem(ob.ACI_PRODUCT, [constant_tt([2, 2, 2], 1.0), constant_tt([2, 2, 2], 2.0)])
    for k in range(2):
        assert p.frame(False, k, 0).shape == (1, 1) and p.frame(True, k, 3).shape == (1, 1)
        assert p.frame(True, k, 1).shape == (1, 1) and p.frame(True, k, 2).shape == (1, 1) and p.frame(True, k, 0) is None
    inp = [np.array([1.0, 2.0, 10.0, 20.0]).reshape((1, 2, 2), order="F"), np.array([3.0, 30.0, 4.0, 40.0]).reshape((2, 2, 1), order="F")]
    guess = [np.ones((1, 2, 2)), np.array([2.0, 0.0, 0.0, 1.0]).reshape((2, 2, 1), order="F")]  # LUCI picks columns 0, 1 in order
    p = ob.OracleAciProblem(ob.ACI_PRODUCT, [inp], ob.AciOptions(initial_guess=guess))
    assert np.array_equal(p.frame(True, 0, 1), [[3.0, 4.0], [30.0, 40.0]])


@pytest.mark.parametrize("left", [True, False])
def test_one_bond_local_update_matches_dense_product(left):
    # tests.rs:870-913
    p = ob.OracleAciProblem(ob.ACI_PRODUCT, [constant_tt([2, 2], 2.0), constant_tt([2, 2], 3.0)])
    p.local_update(0, left)
    assert np.abs(p.solution().evaluate(list(itertools.product(range(2), range(2)))) - 6.0).max() < 1e-12
    for k in range(2):
        assert p.frame(not left, k, 1).shape == (1, 1)
    e, _ = p.errors()
    assert len(e) == 1 and e[0] <= 1e-12


def test_zero_operator_keeps_nonzero_dimensional_frames():
    # tests.rs:935-983: a zero local block is replaced by a rank-one zero update
    p = ob.OracleAciProblem(lambda v: np.zeros(v.shape[1]), [constant_tt([2, 2, 2], 1.0), constant_tt([2, 2, 2], 2.0)])
    p.local_update(0, True)
    assert p.frame(False, 0, 1).shape == (1, 1)
    assert np.abs(p.solution().evaluate(list(itertools.product(range(2), repeat=3)))).max() == 0.0


def test_global_pivot_injection_fixtures():
    # tests.rs:652-727
    site_dims = [2] * 5
    o = ob.AciOptions()
    p = ob.OracleAciProblem(ob.ACI_PRODUCT, [separable_tt(site_dims, 0.25), separable_tt(site_dims, 0.5)], o)
    for b in range(4):
        p.local_update(b, True)
    before = p.solution().link_dims()
    cols_before = [p.frame(True, 0, b).shape[1] for b in range(2, 5)]
    pivots = [[(k >> s) & 1 for s in range(5)] for k in range(8)]
    p.add_global_pivots(pivots)
    after = p.solution().link_dims()
    assert all(d <= b for d, b in zip(after, [2, 4, 4, 2])) and any(a > b for a, b in zip(after, before))
    for bond in range(1, 5):
        growth = after[bond - 1] - before[bond - 1]
        if bond + 1 < 5:
            assert p.frame(False, 0, bond).shape[0] == after[bond - 1]
        if bond >= 2:
            assert p.frame(True, 0, bond).shape[1] == cols_before[bond - 2] + growth
    site_dims = [2] * 6
    p = ob.OracleAciProblem(ob.ACI_PRODUCT, [separable_tt(site_dims, 0.25), separable_tt(site_dims, 0.5)], o)
    for b in range(5):
        p.local_update(b, True)
    assert p.add_global_pivots([[1, 0, 1, 0, 1, 0]]) == 1
    dims = p.solution().link_dims()
    assert p.add_global_pivots([[1, 0, 1, 0, 1, 0]]) == 0 and p.solution().link_dims() == dims


def test_capped_run_stops_once_rank_saturates():
    # tests.rs:452-499
    site_dims, link = [2] * 10, [2, 4, 4, 4, 4, 4, 4, 4, 2]
    o = ob.AciOptions(max_iters=20, min_iters=2, max_bond_dim=4, tolerance=1e-10)
    r = ob.aci_elementwise(ob.ACI_PRODUCT, [lcg_tt(site_dims, link, 12345), lcg_tt(site_dims, link, 98765)], o)
    assert r.errors[-1] > o.tolerance and len(r.ranks) <= o.min_iters + 2 and r.ranks[-2:] == [4, 4]
    assert r.termination == 1 and all(d <= 4 for d in r.tensor_train.link_dims())


@pytest.mark.parametrize("guard", [False, True])
def test_product_of_random_trains_converges(guard):
    site_dims, link = [2, 3, 2, 3, 2, 2], [2, 3, 3, 3, 2]
    a, b = lcg_tt(site_dims, link, 7), lcg_tt(site_dims, link, 99)
    rng = np.random.default_rng(3)
    guess = [rng.standard_normal(c.shape) for c in lcg_tt(site_dims, [2, 4, 6, 4, 2], 1)]
    r = ob.aci_elementwise(ob.ACI_PRODUCT, [a, b], ob.AciOptions(initial_guess=guess, enable_global_guard=guard, tolerance=1e-12))
    exact = dense(a) * dense(b)
    pts = list(itertools.product(*[range(d) for d in site_dims]))
    got = r.tensor_train.evaluate(pts).reshape(exact.shape)
    assert np.abs(got - exact).max() < 1e-10 * np.abs(exact).max() and r.termination == 0
    # default (splitmix64 Box-Muller) initial guess: same answer, deterministic for a seed
    r1 = ob.aci_elementwise(ob.ACI_PRODUCT, [a, b], ob.AciOptions(rng_seed=5, enable_global_guard=guard))
    r2 = ob.aci_elementwise(ob.ACI_PRODUCT, [a, b], ob.AciOptions(rng_seed=5, enable_global_guard=guard))
    assert np.array_equal(r1.tensor_train.evaluate(pts), r2.tensor_train.evaluate(pts))
    assert np.abs(r1.tensor_train.evaluate(pts).reshape(exact.shape) - exact).max() < 1e-9 * np.abs(exact).max()


def test_validation_errors():
    # tests.rs:1322-1470, :1580-1636
    good = constant_tt([2, 2], 1.0)
    for kw in (dict(max_iters=0), dict(min_iters=0), dict(max_bond_dim=0), dict(min_iters=5, max_iters=3), dict(tolerance=-1.0),
               dict(tolerance=float("nan")), dict(tolerance=float("inf")), dict(tol_margin_global_search=-1.0)):
        with pytest.raises(ob.OracleError):
            o = ob.AciOptions(**{k: v for k, v in kw.items() if k != "max_bond_dim"})
            if "max_bond_dim" in kw:
                o.max_bond_dim, c = 0, None
                oc = o.to_c()
                oc.has_max_bond_dim = 1
                o.to_c = lambda oc=oc: oc
            ob.aci_elementwise(ob.ACI_PRODUCT, [good], o)
    with pytest.raises(ob.OracleError):
        ob.aci_elementwise(ob.ACI_PRODUCT, [])
    with pytest.raises(ob.OracleError):
        ob.aci_elementwise(ob.ACI_PRODUCT, [good, constant_tt([2, 2, 2], 1.0)])
    with pytest.raises(ob.OracleError):
        ob.aci_elementwise(ob.ACI_PRODUCT, [good, constant_tt([2, 3], 1.0)])
    with pytest.raises(ob.OracleError):  # explicit guess with other site dims / above the cap
        ob.aci_elementwise(ob.ACI_PRODUCT, [good], ob.AciOptions(initial_guess=constant_tt([2, 3], 1.0)))
    with pytest.raises(ob.OracleError):
        ob.aci_elementwise(ob.ACI_PRODUCT, [good], ob.AciOptions(initial_guess=[np.ones((1, 2, 2)), np.ones((2, 2, 1))], max_bond_dim=1))
