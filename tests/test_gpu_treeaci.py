"""TreeACI local step on the device (t4a_gpu_treeaci_local_update_f64: aci_pi_kernel + the engine's rrLU / factor kernels) against the
oracle and the reference's vectors (crates/tensor4all-treeaci/src/local_update/tests/mod.rs)."""
import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_treeaci import two_node_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.skip("no GPU")
    return t4a_amd


def test_reference_vectors_on_the_device(t4a):
    r1, c1 = two_node_frames(1.0)
    r2, c2 = two_node_frames(2.0)
    seen = {}

    def op(values):
        seen["flat"] = values.reshape(-1, order="F").copy()
        return values[0] * values[1]

    u = t4a.treeaci_local_update([r1, r2], [c1, c2], op)
    assert np.array_equal(seen["flat"], [43.0, 86.0, 86.0, 172.0, 430.0, 860.0, 860.0, 1720.0])  # tests/mod.rs:50-53
    assert np.array_equal(u.local_values, [3698.0, 14792.0, 369800.0, 1479200.0]) and u.sampled_scale == 1479200.0
    v = t4a.treeaci_local_update([r1, r2], [c1, c2])  # (fused product)
    assert np.array_equal(v.local_values, u.local_values)
    for zero in (False, True):  # tests/mod.rs:84-123
        w = t4a.treeaci_local_update([r1], [c1], (lambda x: np.zeros(x.shape[1])) if zero else (lambda x: x[0]))
        assert w.left.shape == (2, 1) and w.right.shape == (1, 2)
        assert np.abs((w.left @ w.right).reshape(-1, order="F") - w.local_values).max() < 1e-10
        if zero:
            assert w.row_indices == [0] and w.col_indices == [0] and not w.left.any() and not w.right.any()


@pytest.mark.parametrize("left_orthogonal", [True, False])
@pytest.mark.parametrize("op_name", ["product", "sum", "callback"])
def test_random_frames_match_the_oracle(t4a, left_orthogonal, op_name):
    rng = np.random.default_rng(11 + int(left_orthogonal))
    bonds, rows, cols = [5, 3, 4], 140, 96
    rf = [rng.standard_normal((b, rows)) for b in bonds]
    cf = [rng.standard_normal((b, cols)) for b in bonds]
    if op_name == "product":
        dop = oop = ob.ACI_PRODUCT
    elif op_name == "sum":
        dop = oop = ob.ACI_SUM
    else:
        dop = oop = lambda v: v[0] * v[1] - 0.5 * v[2]  # noqa: E731
    for kw in (dict(tolerance=1e-10), dict(max_bond_dim=7), dict(tolerance=1e-6, scale_tolerance=False)):
        d = t4a.treeaci_local_update(rf, cf, dop, left_orthogonal=left_orthogonal, **kw)
        o = ob.treeaci_local_update(rf, cf, oop, left_orthogonal=left_orthogonal, **kw)
        assert np.array_equal(d.local_values.view(np.uint64), o.local_values.view(np.uint64))  # same summation order: bitwise
        assert d.sampled_scale == o.sampled_scale
        assert d.rank == o.rank and d.row_indices == o.row_indices and d.col_indices == o.col_indices
        assert np.array_equal(d.pivot_errors, o.pivot_errors)
        scale = max(1.0, np.abs(o.left).max(), np.abs(o.right).max())
        assert np.abs(d.left - o.left).max() <= 1e-10 * scale and np.abs(d.right - o.right).max() <= 1e-10 * scale


def test_argument_errors(t4a):
    r1, c1 = two_node_frames(1.0)
    with pytest.raises(ValueError):
        t4a.treeaci_local_update([r1], [c1[:1]])
    with pytest.raises(t4a.T4aError):
        t4a.treeaci_local_update([], [])

    def failing(values):
        raise RuntimeError("sentinel")  # tests/mod.rs:125-140: a failing operator stops the step
    with pytest.raises(RuntimeError, match="sentinel"):
        t4a.treeaci_local_update([r1], [c1], failing)
