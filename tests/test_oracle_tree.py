"""CPU oracle of the TreeTCI crate (oracle/t4a_oracle_tree.hpp) against the reference's own tests
(crates/tensor4all-treetci/src/{graph,state,proposer,update,optimize,materialize}/tests.rs)."""
import itertools

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import OracleTreeTCI2, TreeOptions

SAMPLE_EDGES = [(0, 1), (1, 2), (1, 3), (3, 4), (4, 5), (4, 6)]  # graph/tests.rs:5-18


def identity2(idx):
    return 1.0 if idx[0] == idx[1] else 0.0


def test_graph_utils_match_reference_tree():
    # graph/tests.rs:21-77
    t = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    assert t.subregion_vertices(1, 3) == ([0, 1, 2], [3, 4, 5, 6])
    assert set(t.candidate_edges(1, 3)) == {(0, 1), (1, 2), (3, 4)}
    assert t.distance_edges(1, 3) == {(1, 3): 0, (0, 1): 1, (1, 2): 1, (3, 4): 1, (4, 5): 2, (4, 6): 2}
    assert t.edges() == SAMPLE_EDGES  # proposer/tests.rs:24-38 (AllEdges visits the sorted edge list)


def test_graph_rejects_non_trees():
    with pytest.raises(ob.OracleError):
        OracleTreeTCI2([2] * 3, [(0, 1)])  # disconnected / too few edges
    with pytest.raises(ob.OracleError):
        OracleTreeTCI2([2] * 3, [(0, 1), (1, 1)])  # self loop
    with pytest.raises(ob.OracleError):
        OracleTreeTCI2([2] * 3, [(0, 1), (0, 1)])  # duplicate
    with pytest.raises(ob.OracleError):
        OracleTreeTCI2([2] * 4, [(0, 1), (1, 2), (0, 2)])  # cycle + isolated site
    with pytest.raises(ob.OracleError):
        OracleTreeTCI2([2] * 6, SAMPLE_EDGES)  # state/tests.rs:22-26


def test_linear_chain_edges():
    # graph/tests.rs:79-90
    t = OracleTreeTCI2([2] * 5, [(i, i + 1) for i in range(4)])
    assert t.edges() == [(0, 1), (1, 2), (2, 3), (3, 4)]


def test_add_global_pivots_projects_to_each_edge_bipartition():
    # state/tests.rs:28-50
    t = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
    assert t.pivots([0, 1, 2]).tolist() == [[0, 0, 0], [1, 0, 1]]
    assert t.pivots([3, 4, 5, 6]).tolist() == [[0, 0, 0, 0], [0, 1, 0, 1]]
    assert t.pivots(range(7)).shape == (0, 7)


def test_default_proposer_matches_neighbor_product_assembly():
    # proposer/tests.rs:40-76
    t = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
    iset, jset = t.candidates(1, 3)
    assert iset.tolist() == [[0, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1]]
    assert jset.tolist() == [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 1], [1, 1, 0, 1]]


def test_default_proposer_unions_history_candidates():
    # proposer/tests.rs:78-95
    t = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
    t.push_history([0, 1, 2, 3], [[1, 1, 1, 1]])
    iset, _ = t.candidates(3, 4)
    assert [1, 1, 1, 1] in iset.tolist()
    assert len({tuple(r) for r in iset.tolist()}) == len(iset)  # union keeps first occurrences only


def test_update_edge_selects_identity_pivots_on_two_site_tree():
    # update/tests.rs:22-72
    t = OracleTreeTCI2([2, 2], [(0, 1)], identity2)
    t.add_global_pivots([[0, 0]])
    t.flush_pivot_errors()
    sel = t.update_edge(0, 1, rel_tol=0.0, abs_tol=0.0)
    assert sel["rank"] == 2
    assert t.pivots([0]).tolist() == [[0], [1]]
    assert t.pivots([1]).tolist() == [[0], [1]]
    assert abs(t.max_sample_value() - 1.0) < 1e-12
    assert abs(t.max_bond_error()) < 1e-12
    assert abs(t.pivot_errors()[-1]) < 1e-12


def test_candidate_matrix_is_column_major_over_left_candidates():
    # update/tests.rs:91-135: the value encodes the point, right candidates vary slowest
    dims = [4, 4, 4, 4]

    def code(idx):
        v = 0.0
        for x in idx:
            v = v * 10.0 + float(x)
        return v

    t = OracleTreeTCI2(dims, [(0, 1), (1, 2), (2, 3)], code)
    t.add_global_pivots([[0, 0, 2, 3], [1, 1, 1, 0]])
    lc, rc = t.candidates(1, 2)
    sel = t.update_edge(1, 2, rel_tol=0.0, abs_tol=0.0)
    # the first pivot is the largest code: rows (0..1 sites) and cols (2..3 sites) of the maximum entry
    vals = np.array([[code(list(l) + list(r)) for r in rc] for l in lc])
    i, j = np.unravel_index(np.argmax(np.abs(vals)), vals.shape)
    assert (sel["row_indices"][0], sel["col_indices"][0]) == (i, j)
    assert t.max_sample_value() == vals.max()


def test_optimize_rejects_invalid_options():
    # optimize/tests.rs:10-75
    for tol in (-1.0, float("nan"), float("inf"), float("-inf")):
        t = OracleTreeTCI2([2, 2], [(0, 1)], identity2)
        t.add_global_pivots([[0, 0]])
        with pytest.raises(ob.OracleError):
            t.optimize(TreeOptions(tolerance=tol))
    for kw in ({"max_iter": 0}, {"tol_margin_global_search": -1.0}, {"tol_margin_global_search": float("nan")},
               {"tol_margin_global_search": float("inf")}):
        t = OracleTreeTCI2([2, 2], [(0, 1)], identity2)
        t.add_global_pivots([[0, 0]])
        with pytest.raises(ob.OracleError):
            t.optimize(TreeOptions(**kw))


def test_optimize_converges_and_stops_early_on_two_site_identity():
    # optimize/tests.rs:100-180
    t = OracleTreeTCI2([2, 2], [(0, 1)], identity2)
    t.add_global_pivots([[0, 0]])
    ranks, errors = t.optimize(TreeOptions(tolerance=1e-12, max_iter=4))
    assert ranks[-1] == 2 and t.max_bond_dim() == 2
    assert abs(errors[-1]) <= 1e-12
    assert len(ranks) < 4 and len(ranks) == len(errors)


def test_optimize_stops_early_when_bond_dim_saturated():
    # optimize/tests.rs:187-222
    t = OracleTreeTCI2([3, 3], [(0, 1)], identity2)
    t.add_global_pivots([[0, 0]])
    ranks, errors = t.optimize(TreeOptions(tolerance=1e-12, max_iter=10, max_bond_dim=1))
    assert len(ranks) < 10 and all(r <= 1 for r in ranks) and errors[-1] > 1e-12


def test_materialize_preserves_two_site_identity():
    # materialize/tests.rs:17-77
    t = OracleTreeTCI2([2, 2], [(0, 1)], identity2)
    t.add_global_pivots([[0, 0]])
    t.optimize(TreeOptions(tolerance=1e-12, max_iter=4))
    t.materialize(0)
    got = t.evaluate([[0, 0], [0, 1], [1, 0], [1, 1]])
    assert np.allclose(got, [1.0, 0.0, 0.0, 1.0], atol=1e-12)


def test_materialize_emits_zero_core_for_zero_pivot_matrix():
    # materialize/tests.rs:160-176
    t = OracleTreeTCI2([2, 2], [(0, 1)], lambda idx: 0.0)
    t.add_global_pivots([[0, 0]])
    t.materialize(0)
    assert np.all(t.site_tensor(1) == 0.0)
    assert np.all(t.evaluate([[0, 0], [0, 1], [1, 0], [1, 1]]) == 0.0)


def test_solve_right_full_piv_lu_recovers_target():
    # materialize/tests.rs:104-158 (real-valued analogue)
    target = np.array([[1.0, 5.0], [-3.0, -7.0]])
    pivot = np.array([[2.0, 4.0], [-1.0, 3.0]])
    assert np.allclose(ob.solve_right_full_piv_lu(target @ pivot, pivot), target, atol=1e-12)
    assert np.allclose(ob.solve_right_full_piv_lu(target, np.eye(2)), target, atol=1e-12)
    rng = np.random.default_rng(5)
    p = rng.standard_normal((9, 9))
    x = rng.standard_normal((14, 9))
    assert np.allclose(ob.solve_right_full_piv_lu(x @ p, p), x, atol=1e-10)
    with pytest.raises(ob.OracleError):
        ob.solve_right_full_piv_lu(x @ p, np.zeros((9, 9)))


def _branched_fn(idx):
    # separable-plus-coupling function on the 7-site sample tree
    x = np.asarray(idx, dtype=np.float64)
    return float(np.cos(0.3 * x.sum()) + 0.1 * x[0] * x[6] + 1.0 / (1.0 + x[2] + 2.0 * x[5]))


def test_crossinterpolate2_on_branched_tree_reproduces_function():
    dims = [3, 2, 3, 2, 2, 3, 2]
    t = OracleTreeTCI2(dims, SAMPLE_EDGES, _branched_fn)
    ranks, errors = t.crossinterpolate2([[0] * 7], TreeOptions(tolerance=1e-10, max_iter=8, seed=3))
    assert errors[-1] < 1e-10
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    exact = np.array([_branched_fn(p) for p in pts])
    for center in (0, 3, 6):
        t.materialize(center)
        assert np.max(np.abs(t.evaluate(pts) - exact)) < 1e-8
    # tensor shapes: [d, incoming bonds..., bond to the parent]
    t.materialize(0)
    assert t.site_tensor(0).ndim == 2 and t.site_tensor(1).ndim == 4 and t.site_tensor(6).ndim == 2


def test_chain_graph_matches_function_and_bond_bookkeeping():
    dims = [2] * 6
    f = lambda idx: 1.0 / (1.0 + sum((k + 1) * v for k, v in enumerate(idx)))
    t = OracleTreeTCI2(dims, [(i, i + 1) for i in range(5)], f)
    ranks, errors = t.crossinterpolate2([], TreeOptions(tolerance=1e-12, max_iter=6, enable_global_pivots=False))
    assert len(t.bond_errors()) == 5
    assert t.max_bond_dim() == ranks[-1]
    t.materialize(2)
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    assert np.max(np.abs(t.evaluate(pts) - np.array([f(p) for p in pts]))) < 1e-9


def test_crossinterpolate2_rejects_zero_initial_pivots():
    # api.rs:82-90
    t = OracleTreeTCI2([2, 2], [(0, 1)], lambda idx: 0.0)
    with pytest.raises(ob.OracleError):
        t.crossinterpolate2([[0, 0]], TreeOptions())


def test_simple_and_truncated_proposers():
    # proposer/tests.rs:117-166: deterministic for a fixed seed, d * chi candidates per side, ordered subset of the default
    t = OracleTreeTCI2([2] * 7, SAMPLE_EDGES)
    t.add_global_pivots([[0] * 7, [1, 0, 1, 0, 1, 0, 1]])
    default_i, default_j = t.candidates(1, 3)
    t.set_proposer(1, 7)
    first, second = t.candidates(1, 3), t.candidates(1, 3)
    assert first[0].tolist() == second[0].tolist() and first[1].tolist() == second[1].tolist()
    assert len(first[0]) > 0 and first[0].shape[1] == 3 and first[1].shape[1] == 4
    t.set_proposer(2, 7)
    ti, tj = t.candidates(1, 3)
    assert t.candidates(1, 3)[0].tolist() == ti.tolist()
    assert len(ti) == 4 and len(tj) == 4 and tj.tolist() == default_j.tolist()
    pos = [default_i.tolist().index(c) for c in ti.tolist()]
    assert pos == sorted(pos) and len(set(pos)) == 4
    t.set_proposer(2, 8)
    assert t.candidates(1, 3)[1].tolist() == default_j.tolist()


def test_truncated_proposer_interpolates_branched_tree():
    dims = [3, 2, 3, 2, 2, 3, 2]
    t = OracleTreeTCI2(dims, SAMPLE_EDGES, _branched_fn)
    t.set_proposer(2, 5)
    ranks, errors = t.crossinterpolate2([[0] * 7], TreeOptions(tolerance=1e-10, max_iter=12, seed=3))
    assert errors[-1] < 1e-8
    t.materialize(0)
    pts = np.array(list(itertools.product(*[range(d) for d in dims])))
    assert np.max(np.abs(t.evaluate(pts) - np.array([_branched_fn(p) for p in pts]))) < 1e-6
