"""Randomised bitwise parity of the rrLU entry point against the oracle: mixed shapes (single- and multi-workgroup
plans back to back, which exercises the alternating key tables and the in-kernel clean-up), rank-deficient inputs,
ties, all stop rules, both orthogonalities."""
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


def random_case(rng):
    kind = rng.integers(0, 5)
    m = int(rng.integers(1, 420)) if kind != 4 else int(rng.integers(300, 900))
    n = int(rng.integers(1, 420)) if kind != 4 else int(rng.integers(300, 900))
    if kind == 0:
        a = rng.uniform(-1, 1, size=(m, n))
    elif kind == 1:  # low rank
        r = int(rng.integers(1, max(2, min(m, n))))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    elif kind == 2:  # many exact ties
        a = rng.integers(-2, 3, size=(m, n)).astype(float)
    elif kind == 3:  # wide dynamic range
        a = rng.standard_normal((m, n)) * 10.0 ** rng.integers(-12, 3, size=(m, 1))
    else:
        a = rng.uniform(-1, 1, size=(m, n))
    opts = {}
    if rng.random() < 0.5:
        opts["max_bond_dim"] = int(rng.integers(1, min(m, n) + 1))
    mode = rng.integers(0, 4)
    if mode == 0:
        opts.update(rel_tol=0.0, abs_tol=0.0)
    elif mode == 1:
        opts.update(rel_tol=float(10.0 ** rng.integers(-14, -2)), abs_tol=0.0)
    elif mode == 2:
        opts.update(rel_tol=0.0, abs_tol=float(10.0 ** rng.integers(-10, 0)))
    opts["left_orthogonal"] = bool(rng.integers(0, 2))
    if kind == 4:
        opts["max_bond_dim"] = int(rng.integers(8, 120))  # keeps the oracle fast on the large shapes
    return a, opts


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_rrlu_random_cases_bitwise(t4a, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(60):
        a, opts = random_case(rng)
        lu = t4a.rrlu(a, **opts)
        f, rp, cp, npiv, err = ob.rrlu(a, **opts)
        ctx = f"seed {seed} case {case} shape {a.shape} opts {opts}"
        assert lu.npivots() == npiv, ctx
        assert np.array_equal(lu.row_permutation, rp) and np.array_equal(lu.col_permutation, cp), ctx
        assert np.array_equal(lu.factored, f), ctx
        assert lu.error == err or (np.isnan(lu.error) and np.isnan(err)), ctx


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tci2_random_option_combinations_match_oracle(t4a, seed):
    """Random option combinations on built-in functions: index sets, error history and termination equal the oracle's."""
    from t4a_amd.functions import lorentz, quantics_osc2d, quantics_trig_exp
    rng = np.random.default_rng(500 + seed)
    for case in range(6):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            d = int(rng.integers(3, 6))
            dims = [int(rng.integers(2, 6)) for _ in range(d)]
            spec = lorentz(dims)
        elif kind == 1:
            d = 2 * int(rng.integers(3, 7))
            dims = [2] * d
            spec = quantics_osc2d(d, k1=int(rng.integers(1, 9)), k2=int(rng.integers(1, 9)), k3=int(rng.integers(1, 40)),
                                  eps=0.5, k4=int(rng.integers(1, 90)), delta=0.5)
        else:
            d = int(rng.integers(6, 14))
            dims = [2] * d
            spec = quantics_trig_exp(d)
        opt = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-11, -4)), max_iter=int(rng.integers(2, 7)),
                              max_bond_dim=(None if rng.random() < 0.3 else int(rng.integers(2, 20))),
                              normalize_error=bool(rng.integers(0, 2)), sweep_strategy=int(rng.integers(0, 3)),
                              strictly_nested=bool(rng.integers(0, 2)), ncheck_history=int(rng.integers(1, 4)),
                              nsearch=int(rng.integers(0, 4)), max_nglobal_pivot=int(rng.integers(0, 3)),
                              seed=int(rng.integers(0, 100)))
        g = t4a.TensorCI2(dims)
        g.set_function(spec)
        o = ob.OracleTCI2(dims)
        o.set_function(spec)
        piv = [[0] * d]
        g.crossinterpolate2(piv, opt)
        o.crossinterpolate2(piv, opt)
        ctx = f"seed {seed} case {case} kind {kind} dims {dims} opt {vars(opt)}"
        for p in range(d):
            assert np.array_equal(g.i_set(p), o.i_set(p)), ctx
            assert np.array_equal(g.j_set(p), o.j_set(p)), ctx
        rg, eg = g.history()
        ro, eo = o.history()
        assert list(rg) == list(ro), ctx
        assert np.allclose(eg, eo, rtol=1e-9, atol=1e-300), ctx
        assert g.termination() == o.termination(), ctx
        pts = np.stack([rng.integers(0, dd, size=50) for dd in dims], axis=1)
        gv, ov = g.evaluate(pts), o.evaluate(pts)
        assert np.abs(gv - ov).max() <= 1e-9 * max(1.0, np.abs(ov).max()), ctx
