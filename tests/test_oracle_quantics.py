"""CPU oracle of the quantics front end (oracle/t4a_oracle_quantics.hpp) against the reference's own tests
(crates/tensor4all-quanticstci/src/quantics_tci/tests/mod.rs)."""
import math

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import FUSED, INTERLEAVED, QtciOptions


def check_cache_discrete(q, f):
    cache = q.cachedata()
    assert cache
    for quantics, val in cache.items():
        assert abs(val - f(q.quantics_to_grididx(list(quantics)))) < 1e-10


def test_discrete_simple_function_and_structure():
    # tests/mod.rs:41-114
    f = lambda idx: float(idx[0] + idx[1])
    q = ob.quanticscrossinterpolate_discrete([4, 4], f, None,
                                             QtciOptions(tolerance=1e-10, n_random_init_pivot=3, unfolding_scheme=FUSED, seed=1))
    assert q.evaluate([[2, 3]])[0] == pytest.approx(5.0, abs=1e-8)
    assert q.evaluate([[0, 0]])[0] == pytest.approx(0.0, abs=1e-8)
    assert q.evaluate([[3, 3]])[0] == pytest.approx(6.0, abs=1e-8)
    assert q.rank() <= 3 and q.rank() > 0
    assert len(q.link_dims()) == 1 and q.local_dimensions() == [4, 4]
    assert not q.is_discretized()
    check_cache_discrete(q, f)


def test_size_and_input_validation():
    # tests/mod.rs:116-147, 494-503, 166-190, 480-492
    one = lambda x: 1.0
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_discrete([5, 5], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_discrete([], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_discrete([4, 8], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([[], [0.0, 1.0]], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([[0.0, float("nan"), 1.0, 2.0]], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 1.0, 2.0]], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 2.0]], one)
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_from_arrays([[0.0, 1.0, 2.0, 3.0], list(map(float, range(8)))], one)


def test_from_arrays_uses_interior_coordinates():
    # tests/mod.rs:148-164: non-uniform coordinates -> inherent grid + lookup
    q = ob.quanticscrossinterpolate_from_arrays([[0.0, 0.5, 2.0, 5.0]], lambda x: x[0] + 1.0, None,
                                                QtciOptions(tolerance=1e-10, n_random_init_pivot=2, unfolding_scheme=FUSED, seed=2))
    assert q.evaluate([[1]])[0] == pytest.approx(1.5, abs=1e-8)
    assert q.evaluate([[2]])[0] == pytest.approx(3.0, abs=1e-8)
    assert not q.is_discretized()


def test_discrete_integral_returns_sum():
    # tests/mod.rs:255-275
    q = ob.quanticscrossinterpolate_discrete([4], lambda idx: 1.0, None,
                                             QtciOptions(tolerance=1e-10, n_random_init_pivot=3, unfolding_scheme=FUSED, seed=3))
    assert q.integral() == pytest.approx(q.sum(), abs=1e-10) and q.integral() == pytest.approx(4.0, abs=1e-8)
    with pytest.raises(ob.OracleError) as e:  # tests/mod.rs:236-253: cachedata_origcoord on an inherent grid
        q.quantics_to_origcoord([0, 0])
    assert "original coordinates are only available for discretized grids" in str(e.value)


def test_continuous_grid_interpolation_and_cache_coordinates():
    # tests/mod.rs:277-331: f(x) = x^2 on [0, 1] with 8 points including the end point
    q = ob.quanticscrossinterpolate([3], lambda x: x[0] * x[0], [0.0], [1.0], include_endpoint=True,
                                    options=QtciOptions(tolerance=1e-12, n_random_init_pivot=5, seed=4))
    assert q.is_discretized() and q.rank() > 0
    cache = q.cachedata()
    assert cache
    for quantics, val in cache.items():
        x = q.quantics_to_origcoord(list(quantics))[0]
        assert abs(val - x * x) < 1e-10
    assert q.grid_step() == [pytest.approx(1.0 / 7.0, rel=1e-15)]
    xs = np.arange(8) / 7.0
    assert np.allclose(q.evaluate(np.arange(8).reshape(-1, 1)), xs * xs, atol=1e-10)


def test_continuous_grid_integral():
    # tests/mod.rs:333-355: 16 points without the end point, step 1/16
    q = ob.quanticscrossinterpolate([4], lambda x: 1.0, [0.0], [1.0],
                                    options=QtciOptions(tolerance=1e-12, n_random_init_pivot=3, seed=5))
    assert q.integral() == pytest.approx(1.0, abs=1e-8)
    assert q.grid_step() == [1.0 / 16.0]


def test_initial_pivots_are_converted_and_validated():
    # tests/mod.rs:357-420 and :453-478
    f = lambda idx: float(idx[0] * idx[1])
    q = ob.quanticscrossinterpolate_discrete([4, 4], f, [[0, 0], [1, 2]],
                                             QtciOptions(tolerance=1e-10, n_random_init_pivot=3, unfolding_scheme=FUSED, seed=6))
    check_cache_discrete(q, f)
    assert q.evaluate([[0, 0]])[0] == pytest.approx(0.0, abs=1e-8)
    assert q.evaluate([[3, 3]])[0] == pytest.approx(9.0, abs=1e-8)
    with pytest.raises(ob.OracleError) as e:
        ob.quanticscrossinterpolate_discrete([4], lambda idx: 1.0, [[4]], QtciOptions(n_random_init_pivot=0))
    assert "Grid index 4" in str(e.value)
    with pytest.raises(ob.OracleError) as e:
        ob.quanticscrossinterpolate([3], lambda x: 1.0, [0.0], [1.0], include_endpoint=True, initial_pivots=[[8]],
                                    options=QtciOptions(n_random_init_pivot=0))
    assert "Grid index 8" in str(e.value)
    q = ob.quanticscrossinterpolate([3], lambda x: x[0], [0.0], [1.0], include_endpoint=True, initial_pivots=[[1], [4]],
                                    options=QtciOptions(tolerance=1e-12, n_random_init_pivot=3, seed=7))
    for quantics, val in q.cachedata().items():
        assert abs(val - q.quantics_to_origcoord(list(quantics))[0]) < 1e-10


def test_from_arrays_valid_nonuniform_2d():
    # tests/mod.rs:505-541
    xv = [[0.0, 0.5, 2.0, 3.0], [0.0, 1.0, 2.0, 4.0]]
    q = ob.quanticscrossinterpolate_from_arrays(xv, lambda x: x[0] + x[1], None,
                                                QtciOptions(tolerance=1e-10, n_random_init_pivot=3, unfolding_scheme=FUSED, seed=8))
    assert not q.is_discretized() and q.rank() > 0
    for quantics, val in q.cachedata().items():
        g = q.quantics_to_grididx(list(quantics))
        assert abs(val - (xv[0][g[0]] + xv[1][g[1]])) < 1e-10
    assert q.evaluate([[0, 0]])[0] == pytest.approx(0.0, abs=1e-8)
    assert q.evaluate([[3, 3]])[0] == pytest.approx(7.0, abs=1e-8)


def test_from_arrays_1d_polynomial():
    # tests/mod.rs:552-581: 128 uniform points from -3 to 2 -> DiscretizedGrid with the end point
    fs = lambda x: 0.1 * x * x - math.pi * x + 2.0
    n = 128
    xv = [-3.0 + 5.0 * i / (n - 1) for i in range(n)]
    q = ob.quanticscrossinterpolate_from_arrays([xv], lambda x: fs(x[0]), None, QtciOptions(tolerance=1e-8, seed=9))
    assert q.is_discretized()
    ranks, errors = q.history()
    assert errors[-1] < 1e-8
    got = q.evaluate(np.arange(n).reshape(-1, 1))
    assert np.abs(got - np.array([fs(x) for x in xv])).max() < 1e-6


def test_grid_unfolding_conventions():
    # interleaved: one binary site per (level, variable), most significant bits first; fused: first variable least significant
    q = ob.quanticscrossinterpolate_discrete([8, 8], lambda idx: 1.0 + idx[0] + 10.0 * idx[1], None,
                                             QtciOptions(tolerance=1e-10, n_random_init_pivot=0, unfolding_scheme=INTERLEAVED))
    assert q.local_dimensions() == [2] * 6
    assert q.grididx_to_quantics([5, 3]) == [1, 0, 0, 1, 1, 1]   # x = 101b, y = 011b interleaved x first
    assert q.quantics_to_grididx([1, 0, 0, 1, 1, 1]) == [5, 3]
    qf = ob.quanticscrossinterpolate_discrete([8, 8], lambda idx: 1.0 + idx[0] + 10.0 * idx[1], None,
                                              QtciOptions(tolerance=1e-10, n_random_init_pivot=0, unfolding_scheme=FUSED))
    assert qf.local_dimensions() == [4] * 3
    assert qf.grididx_to_quantics([5, 3]) == [1, 2, 3]           # levels: (x1,y0) (x0,y1) (x1,y1) -> x + 2 y
    assert qf.quantics_to_grididx([1, 2, 3]) == [5, 3]
    pts = np.array([[i, j] for i in range(8) for j in range(8)])
    exact = 1.0 + pts[:, 0] + 10.0 * pts[:, 1]
    assert np.allclose(q.evaluate(pts), exact, atol=1e-8) and np.allclose(qf.evaluate(pts), exact, atol=1e-8)


# ------------------------------------------------------------------------------------------------ batched (vector valued)
def _eval_components(tt, q, n_comp):
    return np.array([tt.evaluate([list(q) + [c]])[0] for c in range(n_comp)])


def test_batched_two_components_1d():
    # batched/tests/mod.rs:6-71
    f = lambda x: [math.sin(2 * math.pi * x[0]) + 1.0, math.cos(2 * math.pi * x[0])]
    r = ob.quanticscrossinterpolate_batched([6], f, [2], [0.0], [1.0], options=QtciOptions(tolerance=1e-8, seed=1))
    tt = r.tensor_train()
    assert len(tt) == 7
    for i in range(64):
        q = [(i >> (5 - b)) & 1 for b in range(6)]
        assert np.abs(_eval_components(tt, q, 2) - np.array(f([i / 64.0]))).max() < 1e-6


def test_batched_matrix_valued_and_scalar_equivalent():
    # batched/tests/mod.rs:73-129, 158-211
    f = lambda x: [(c + 1.0) * (x[0] + 1.0) for c in range(4)]
    r = ob.quanticscrossinterpolate_batched([4], f, [2, 2], [0.0], [1.0], options=QtciOptions(tolerance=1e-8, seed=2))
    tt = r.tensor_train()
    assert len(tt) == 5
    for i in range(16):
        q = [(i >> (3 - b)) & 1 for b in range(4)]
        assert np.abs(_eval_components(tt, q, 4) - np.array(f([i / 16.0]))).max() < 1e-8
    s = ob.quanticscrossinterpolate([4], lambda x: x[0] * x[0], [0.0], [1.0], options=QtciOptions(tolerance=1e-8, seed=3))
    b = ob.quanticscrossinterpolate_batched([4], lambda x: [x[0] * x[0]], [1], [0.0], [1.0], options=QtciOptions(tolerance=1e-8, seed=3))
    bt = b.tensor_train()
    assert s.n_sites == 4 and len(bt) == 5
    for i in range(16):
        q = [(i >> (3 - b_)) & 1 for b_ in range(4)]
        assert abs(s.evaluate([[i]])[0] - bt.evaluate([q + [0]])[0]) < 1e-10


def test_batched_errors_and_shared_cache():
    # batched/tests/mod.rs:131-156, 213-280
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_batched([4], lambda x: [], [], [0.0], [1.0])
    with pytest.raises(ob.OracleError):
        ob.quanticscrossinterpolate_batched([4], lambda x: [], [0], [0.0], [1.0])
    with pytest.raises(ob.OracleError) as e:
        ob.quanticscrossinterpolate_batched([2], lambda x: [1.0], [2], [0.0], [1.0])
    assert "expected at least 2" in str(e.value)
    r = ob.quanticscrossinterpolate_batched([3], lambda x: [x[0] + 1.0, x[0] * x[0] + 1.0], [2], [0.0], [1.0],
                                            options=QtciOptions(tolerance=1e-8, n_random_init_pivot=0))
    assert len(r.tensor_train()) == 4 and r.user_calls() <= 8
