"""GPU parity tests of the small-problem engine (kernels_small.hip / tci2_small.hip): optimize_with_finder of a small problem as ONE
launch (tensorci2.rs:1626-1802: iteration loop, update_pivots chain, fill_site_tensors, convergence_criterion, final 1-site sweep).

Three runs side by side on the same inputs: the engine, the general device path (set_chain(small_engine=False)) and the CPU oracle.
Index sets, bond errors, pivot errors, ranks, errors, termination and max_sample_value must be IDENTICAL (bit-exact pivot contract);
site tensors within 1e-10.  Cases cover a run the engine completes (configs[1]), runs it hands back to the general path when the
rank outgrows its tiles (configs[2] at reduced depth), both sweep directions, strictly nested sets, local dimensions other than two
and optimize() without the final sweep.
"""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu

PARITY = dict(nsearch=0, max_nglobal_pivot=0)


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def three(t4a, spec, dims, tile32=False):
    s = t4a.TensorCI2(dims)
    s.set_function(spec)
    if tile32:
        s.set_chain(True, small_tile32=True)
    g = t4a.TensorCI2(dims)
    g.set_function(spec)
    g.set_chain(True, small_engine=False)
    o = ob.OracleTCI2(dims)
    o.set_function(spec)
    return s, g, o


def assert_identical(s, g, o, n, core_tol=1e-10, cores=True):
    for p in range(n):
        assert np.array_equal(s.i_set(p), o.i_set(p)), f"I set differs from the oracle at site {p}"
        assert np.array_equal(s.j_set(p), o.j_set(p)), f"J set differs from the oracle at site {p}"
        assert np.array_equal(s.i_set(p), g.i_set(p)) and np.array_equal(s.j_set(p), g.j_set(p))
    sr, se = s.history()
    orr, oe = o.history()
    gr, ge = g.history()
    assert sr == orr == gr
    assert np.array_equal(se, oe) and np.array_equal(se, ge)
    assert s.termination() == o.termination() == g.termination()
    assert s.max_sample_value() == o.max_sample_value() == g.max_sample_value()
    assert np.array_equal(s.bond_errors(), o.bond_errors()) and np.array_equal(s.bond_errors(), g.bond_errors())
    assert np.array_equal(s.pivot_errors(), o.pivot_errors()) and np.array_equal(s.pivot_errors(), g.pivot_errors())
    assert np.array_equal(s.last_sweep_shapes(), o.last_sweep_shapes())
    if cores:
        for p in range(n):
            a, b = s.site_tensor(p), o.site_tensor(p)
            assert a.shape == b.shape, f"core shape differs at site {p}: {a.shape} vs {b.shape}"
            if a.size:
                assert np.abs(a - b).max() <= core_tol * max(1.0, np.abs(b).max()), f"core values differ at site {p}"


@pytest.mark.parametrize("nbits", [6, 12, 20])
def test_cfg2_runs_completely_inside_one_launch(t4a, nbits):
    """BASELINE configs[1] (d = 20) and shorter versions: rank 2, every candidate matrix at most 6 x 6."""
    from t4a_amd.functions import quantics_trig_exp
    spec = quantics_trig_exp(nbits)
    opts = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, seed=42, **PARITY)
    s, g, o = three(t4a, spec, [2] * nbits)
    for h in (s, g, o):
        h.crossinterpolate2([[0] * nbits], opts)
    assert_identical(s, g, o, nbits)
    st = s.small_stats()
    assert st["completed"] == 1 and st["handed_back"] == 0 and st["iterations"] == len(s.history()[0]), st
    assert g.small_stats()["completed"] == 0
    rng = np.random.default_rng(5)
    pts = rng.integers(0, 2, size=(300, nbits))
    assert np.abs(s.evaluate(pts) - o.evaluate(pts)).max() <= 1e-10
    assert np.abs(s.evaluate(pts) - ob.fn_eval(spec, pts)).max() <= 1e-6


@pytest.mark.parametrize("strategy", [0, 1, 2])
@pytest.mark.parametrize("nested", [False, True])
def test_sweep_strategies_and_strictly_nested_sets(t4a, strategy, nested):
    from t4a_amd.functions import quantics_trig_exp
    n = 10
    spec = quantics_trig_exp(n, a=7.0, b=0.5, cc=0.3, cs=1.0)
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=8, max_iter=9, sweep_strategy=strategy, strictly_nested=nested, seed=1, **PARITY)
    s, g, o = three(t4a, spec, [2] * n)
    for h in (s, g, o):
        h.crossinterpolate2([[0] * n, [1] * n], opts)
    assert_identical(s, g, o, n)
    assert s.small_stats()["iterations"] >= 1


@pytest.mark.parametrize("nsites,maxb,iters", [(10, 4, 6), (12, 16, 8), (14, 32, 7)])
def test_engine_hands_a_growing_problem_to_the_general_path(t4a, nsites, maxb, iters):
    """BASELINE configs[2] at reduced depth: the rank grows past the engine's tiles (16 entries per set, 16 x 16 matrices by default); the first
    iterations run in the launch, the rest on the bond chain — every observable equals the oracle's and the general path's."""
    from t4a_amd.functions import quantics_osc2d
    spec = quantics_osc2d(nsites, k1=3, k2=5, k3=7, eps=0.1, k4=11, delta=0.3)
    opts = t4a.TCI2Options(tolerance=1e-12, max_bond_dim=maxb, max_iter=iters, seed=42, **PARITY)
    s, g, o = three(t4a, spec, [2] * nsites)
    for h in (s, g, o):
        h.crossinterpolate2([[0] * nsites], opts)
    assert_identical(s, g, o, nsites)
    st = s.small_stats()
    assert st["iterations"] >= 1, st
    if maxb >= 16:
        assert st["handed_back"] == 1 and st["completed"] == 0, st
    rng = np.random.default_rng(6)
    pts = rng.integers(0, 2, size=(300, nsites))
    assert np.abs(s.evaluate(pts) - o.evaluate(pts)).max() <= 1e-10 * max(1.0, s.max_sample_value())


def test_local_dimensions_other_than_two(t4a):
    """Lorentzian on 10^5 (tensorci2/tests/mod.rs:945-1002) and a linear function on mixed dimensions (:397-440)."""
    from t4a_amd.functions import lorentz, linear_sum
    spec = lorentz([10] * 5)
    opts = t4a.TCI2Options(tolerance=1e-8, max_iter=20, **PARITY)
    s, g, o = three(t4a, spec, [10] * 5)
    for h in (s, g, o):
        h.crossinterpolate2([[1] * 5], opts)
    assert_identical(s, g, o, 5)  # (d = 10: the second iteration's matrices outgrow 32 rows, the engine hands over early)
    dims = [3, 4, 2, 5, 3, 2]
    spec = linear_sum(dims, scale=0.5, shift=1.0)
    opts = t4a.TCI2Options(tolerance=1e-10, max_iter=10, **PARITY)
    s, g, o = three(t4a, spec, dims)
    for h in (s, g, o):
        h.crossinterpolate2([[1] * len(dims)], opts)
    assert_identical(s, g, o, len(dims))
    assert s.small_stats()["completed"] == 1, s.small_stats()
    assert max(s.link_dims()) <= 2


def test_optimize_without_the_final_sweep_keeps_the_fill_tensors(t4a):
    from t4a_amd.functions import quantics_trig_exp
    n = 9
    spec = quantics_trig_exp(n, a=3.0, b=2.0)
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=6, max_iter=7, **PARITY)
    s, g, o = three(t4a, spec, [2] * n)
    for h in (s, g, o):
        h.add_global_pivots([[0] * n, [1, 0] * 4 + [1]])
        h.set_max_sample_value(1.0)
        h.optimize(opts, final_sweep1site=False)
    assert_identical(s, g, o, n)
    assert s.small_stats()["completed"] == 1
    # a second optimize call on the same handle has a history: the general path takes it, results still equal
    for h in (s, g, o):
        h.optimize(opts, final_sweep1site=True)
    assert_identical(s, g, o, n)
    assert s.small_stats()["completed"] == 1 and s.small_stats()["not_eligible"] >= 1


def test_engine_is_not_offered_callbacks_or_global_search(t4a):
    from t4a_amd.functions import quantics_trig_exp
    n = 8
    spec = quantics_trig_exp(n)
    s = t4a.TensorCI2([2] * n)
    s.set_function(spec)
    s.crossinterpolate2([[0] * n], t4a.TCI2Options(tolerance=1e-8, max_bond_dim=8, max_iter=6, nsearch=3, max_nglobal_pivot=2, seed=3))
    assert s.small_stats()["completed"] == 0 and s.small_stats()["iterations"] == 0
    c = t4a.TensorCI2([2] * n)
    c.set_function(lambda idx: float(np.cos(sum(idx) * 0.3)))
    c.crossinterpolate2([[0] * n], t4a.TCI2Options(tolerance=1e-8, max_bond_dim=8, max_iter=6, **PARITY))
    assert c.small_stats()["completed"] == 0 and c.small_stats()["iterations"] == 0


@pytest.mark.parametrize("nbits,maxb", [(8, 8), (12, 16), (20, 64)])
def test_rook_pivot_search_inside_the_launch(t4a, nbits, maxb):
    """PivotSearchStrategy::Rook (tensorci2.rs:1904-1929, matrixluci/block_rook.rs:71-190) on a small problem: the candidate matrix is
    materialised in the LDS and factorize_lazy runs on it inside the launch (a built-in functor costs nothing to evaluate) with the
    arithmetic of rook_dense_kernel; max_sample_value sees only the rows and columns the lazy evaluator would have visited.  Engine,
    general device path (device-resident search per bond) and oracle agree on every observable."""
    from t4a_amd.functions import quantics_trig_exp
    spec = quantics_trig_exp(nbits)
    opts = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=maxb, max_iter=8, pivot_search=1, seed=42, **PARITY)
    s, g, o = three(t4a, spec, [2] * nbits)
    o.set_pivot_search(1)
    for h in (s, g, o):
        h.crossinterpolate2([[0] * nbits], opts)
    assert_identical(s, g, o, nbits, core_tol=1e-8)
    st = s.small_stats()
    assert st["completed"] == 1 and st["handed_back"] == 0, st
    assert s.rook_stats()["device_searches"] >= (nbits - 1) * st["iterations"] and s.rook_stats()["host_searches"] == 0
    pts = np.random.default_rng(3).integers(0, 2, size=(200, nbits))
    assert np.abs(s.evaluate(pts) - ob.fn_eval(spec, pts)).max() <= 1e-6


def test_rook_growing_problem_is_handed_over(t4a):
    from t4a_amd.functions import quantics_osc2d
    n = 12
    spec = quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.3, k4=11, delta=0.2)
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=24, max_iter=5, pivot_search=1, **PARITY)
    s, g, o = three(t4a, spec, [2] * n)
    o.set_pivot_search(1)
    for h in (s, g, o):
        h.crossinterpolate2([[0] * n], opts)
    assert_identical(s, g, o, n, core_tol=1e-8)
    assert s.small_stats()["iterations"] >= 1


def _fuzz_case(t4a, seed):
    """One random small problem: function family, local dimensions, initial pivots and options drawn from `seed`."""
    from t4a_amd.functions import quantics_trig_exp, quantics_osc2d, lorentz, linear_sum
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        n = int(rng.integers(4, 15))
        dims = [2] * n
        spec = quantics_trig_exp(n, a=float(rng.uniform(1, 30)), b=float(rng.uniform(0, 3)), cc=float(rng.uniform(-1, 1)), cs=float(rng.uniform(-1, 1)))
    elif kind == 1:
        n = 2 * int(rng.integers(2, 7))
        dims = [2] * n
        spec = quantics_osc2d(n, k1=int(rng.integers(1, 9)), k2=int(rng.integers(1, 9)), k3=int(rng.integers(1, 9)), eps=float(rng.uniform(0, 0.5)),
                              k4=int(rng.integers(0, 5)), delta=float(rng.uniform(0, 0.5)))
    elif kind == 2:
        n = int(rng.integers(3, 8))
        dims = [int(v) for v in rng.integers(2, 5, size=n)]
        spec = lorentz(dims, coeff=float(rng.uniform(0.5, 2)))
    else:
        n = int(rng.integers(3, 9))
        dims = [int(v) for v in rng.integers(2, 6, size=n)]
        spec = linear_sum(dims, scale=float(rng.uniform(0.1, 2)), shift=float(rng.uniform(-1, 1)), site_weights=[int(v) for v in rng.integers(1, 4, size=n)])
    opts = dict(tolerance=float(rng.choice([1e-3, 1e-6, 1e-9, 1e-12])), max_iter=int(rng.integers(1, 9)),
                max_bond_dim=None if rng.random() < 0.4 else int(rng.integers(1, 9)), ncheck_history=int(rng.integers(1, 5)),
                sweep_strategy=int(rng.integers(0, 3)), strictly_nested=bool(rng.random() < 0.3), normalize_error=bool(rng.random() < 0.7),
                pivot_search=int(rng.random() < 0.25), nsearch=0 if rng.random() < 0.7 else 4, max_nglobal_pivot=0)
    n_piv = int(rng.integers(1, 4))
    pivots = [[int(rng.integers(0, d)) for d in dims] for _ in range(n_piv)]
    if kind == 3:
        pivots[0] = [d - 1 for d in dims]  # (a linear function may vanish at a random point: keep one pivot away from zero)
    return spec, dims, opts, pivots, bool(rng.random() < 0.75)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("T4A_FUZZ_N", "60"))))  # (T4A_FUZZ_N=1000: the soak of profiles/r06_small_engine_fuzz.txt)
def test_fuzz_engine_general_path_and_oracle_agree(t4a, seed):
    """Random small problems — four function families, local dimensions 2 .. 5, one to three initial pivots, every option that changes the
    control flow (tolerance, iteration cap, rank cap, history length of the convergence test, sweep direction, strictly nested sets, error
    normalisation, pivot search strategy, with and without the final 1-site sweep): the launch, the general device path and the oracle
    must agree on every observable, whether the launch completes the call or hands it back."""
    spec, dims, opts, pivots, final = _fuzz_case(t4a, seed)
    n = len(dims)
    o_opts = t4a.TCI2Options(**opts)
    s, g, o = three(t4a, spec, dims, tile32=bool(seed % 2))  # (odd seeds: the opt-in 32 x 32 tile)
    o.set_pivot_search(opts["pivot_search"])
    results = []
    for h in (s, g, o):
        try:
            h.add_global_pivots(pivots)
            h.set_max_sample_value(1.0)
            h.optimize(o_opts, final_sweep1site=final)
            results.append(None)
        except Exception as e:  # noqa: BLE001 - all three must fail alike
            results.append(type(e).__name__)
    assert results[0] == results[1], results
    if results[0] is not None or results[2] is not None:
        assert results[1] is not None and results[2] is not None, results
        return
    assert_identical(s, g, o, n, core_tol=1e-8 if opts["pivot_search"] else 1e-10)
    st = s.small_stats()
    assert st["completed"] + st["handed_back"] + st["not_eligible"] >= 1, st


def test_engine_on_several_host_threads_at_once(t4a):
    """Eight host threads, each with its own handles, run engine solves at the same time (one launch per solve, static LDS of a whole compute
    unit per launch, pinned result blocks per handle): every result equals the single-threaded one."""
    import threading
    from t4a_amd.functions import quantics_trig_exp
    n = 16
    specs = [quantics_trig_exp(n, a=3.0 + k, b=0.5 + 0.1 * k, cc=1.0, cs=0.2 * k) for k in range(8)]
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=32, max_iter=12, **PARITY)

    def solve(spec):
        g = t4a.TensorCI2([2] * n)
        g.set_function(spec)
        g.crossinterpolate2([[0] * n], opts)
        assert g.small_stats()["completed"] == 1
        return ([np.asarray(g.i_set(p)).tobytes() + np.asarray(g.j_set(p)).tobytes() for p in range(n)], np.asarray(g.history()[1]).tobytes(),
                [g.site_tensor(p).tobytes() for p in range(n)])

    ref = [solve(s) for s in specs]
    out, errors = {}, []

    def work(k):
        try:
            for _ in range(25):
                out[k] = solve(specs[k])
                if out[k] != ref[k]:
                    errors.append((k, "result differs"))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert all(out[k] == ref[k] for k in range(8))
