"""The oracle's TreeACI local step (oracle/t4a_oracle_treeaci.hpp) against the vectors the reference's own tests hold
(crates/tensor4all-treeaci/src/local_update/tests/mod.rs).  CPU only."""
import numpy as np

import oracle_binding as ob


def two_node_frames(scale):
    """two_node_tree(scale) of the reference's tests (tests/mod.rs:12-32): left (s0, bond) = [s, 2s, 10s, 20s], right (bond, s1) =
    [3, 4, 30, 40], both column-major.  For the edge 0 -> 1 the row candidates are the two values of s0 with frames left[s0, :] and the
    column candidates the two values of s1 with frames right[:, s1] (frames.rs: a leaf's frame is its tensor row)."""
    left = np.array([scale, 2 * scale, 10 * scale, 20 * scale]).reshape((2, 2), order="F")   # [s0, bond]
    right = np.array([3.0, 4.0, 30.0, 40.0]).reshape((2, 2), order="F")                      # [bond, s1]
    return left.T.copy(), right.copy()  # (bond, row_count), (bond, col_count)


def test_local_entries_and_callback_layout_of_the_reference():
    # tests/mod.rs:34-82: two inputs (scale 1 and 2), product operator
    r1, c1 = two_node_frames(1.0)
    r2, c2 = two_node_frames(2.0)
    seen = {}

    def op(values):
        seen["shape"] = values.shape
        seen["flat"] = values.reshape(-1, order="F").copy()
        return values[0] * values[1]

    u = ob.treeaci_local_update([r1, r2], [c1, c2], op)
    assert seen["shape"] == (2, 4)
    assert np.array_equal(seen["flat"], [43.0, 86.0, 86.0, 172.0, 430.0, 860.0, 860.0, 1720.0])
    assert np.array_equal(u.batch, [43.0, 86.0, 86.0, 172.0, 430.0, 860.0, 860.0, 1720.0])
    assert np.array_equal(u.local_values, [3698.0, 14792.0, 369800.0, 1479200.0])
    assert u.sampled_scale == 1479200.0
    # the built-in product is the same operator
    v = ob.treeaci_local_update([r1, r2], [c1, c2], ob.ACI_PRODUCT)
    assert np.array_equal(v.local_values, u.local_values) and v.row_indices == u.row_indices and v.col_indices == u.col_indices


def test_rank_one_and_zero_targets_reconstruct():
    # tests/mod.rs:84-123
    r1, c1 = two_node_frames(1.0)
    for zero in (False, True):
        u = ob.treeaci_local_update([r1], [c1], (lambda v: np.zeros(v.shape[1])) if zero else (lambda v: v[0]))
        assert u.left.shape == (2, 1) and u.right.shape == (1, 2)
        rec = u.left @ u.right
        assert np.abs(rec.reshape(-1, order="F") - u.local_values).max() < 1e-10
        if zero:
            assert u.row_indices == [0] and u.col_indices == [0] and not u.left.any() and not u.right.any()


def test_tolerance_modes_and_rank_cap():
    rng = np.random.default_rng(3)
    rf = [rng.standard_normal((3, 12)), rng.standard_normal((2, 12))]
    cf = [rng.standard_normal((3, 9)), rng.standard_normal((2, 9))]
    full = ob.treeaci_local_update(rf, cf, ob.ACI_PRODUCT, tolerance=1e-13)
    assert full.rank == 6  # product of a rank-3 and a rank-2 matrix
    rec = full.left @ full.right
    assert np.abs(rec.reshape(-1, order="F") - full.local_values).max() < 1e-9 * full.sampled_scale
    cap = ob.treeaci_local_update(rf, cf, ob.ACI_PRODUCT, max_bond_dim=4)
    assert cap.rank == 4 and len(cap.pivot_errors) == 5
    absolute = ob.treeaci_local_update(rf, cf, ob.ACI_SUM, tolerance=1e-9, scale_tolerance=False, left_orthogonal=False)
    assert absolute.rank == 5  # sum of a rank-3 and a rank-2 matrix
