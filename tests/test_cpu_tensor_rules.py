"""Host-side pieces of the dense labelled-tensor seam through the C ABI (no GPU needed): the reference's rank rules
(defaults/svd.rs:150, defaults/qr.rs:74) against its own fixtures and the oracle, and index validation, which happens
before the device is touched."""
import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_tensor import upper


def test_rank_rules_match_reference_fixtures_and_oracle():
    import t4a_amd as t
    P = t.SvdTruncationPolicy
    assert t.svd_retained_rank([], P(1e-6)) == 1
    assert t.svd_retained_rank([0.0, 0.0], P(1e-6)) == 1
    assert t.svd_retained_rank([5.0, 1e-9], P(1e-6)) == 1
    assert t.svd_retained_rank([5.0, 1.0]) == 2
    assert t.svd_retained_rank([5.0, 1.0], P(1.5, scale=t.ABSOLUTE)) == 1
    assert t.svd_retained_rank([10.0, 1.0, 0.1], P(0.05, measure=t.SQUARED_VALUE, rule=t.DISCARDED_TAIL_SUM)) == 1
    assert t.svd_retained_rank([1.0, 0.1, 0.1], P(0.02, t.ABSOLUTE, t.SQUARED_VALUE, t.DISCARDED_TAIL_SUM)) == 2
    assert t.qr_retained_rank([3.0, 0.0, 1.0, 1e-14], 2, 2, 1e-10) == 1
    assert t.qr_retained_rank([0.0] * 4, 2, 2, 1.0) == 1
    assert t.qr_retained_rank([], 0, 2, 1e-12) == 1
    tr = upper(3, 3, [(0, 0, 10.0), (0, 1, 0.5), (0, 2, 0.1), (1, 1, 0.01), (2, 2, 0.001)])
    assert t.qr_retained_rank(tr, 3, 3, 0.01) == 1 and t.qr_retained_rank(tr, 3, 3, 1e-4) == 2
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 9))
        s = np.sort(np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-8, 2, size=n))[::-1]
        kw = dict(threshold=float(10.0 ** rng.integers(-10, 0)), scale=int(rng.integers(0, 2)), measure=int(rng.integers(0, 2)),
                  rule=int(rng.integers(0, 2)))
        assert t.svd_retained_rank(s, P(**kw)) == ob.svd_retained_rank(s, **kw)
        k, m = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        r = np.triu(rng.standard_normal((k, m)) * 10.0 ** rng.integers(-6, 1, size=(k, 1))).ravel(order="F")
        rtol = float(10.0 ** rng.integers(-8, 0))
        assert t.qr_retained_rank(r, k, m, rtol) == ob.qr_retained_rank(r, k, m, rtol)


def test_index_validation_precedes_the_device():
    import t4a_amd as t
    a = np.zeros((3, 4, 5))
    with pytest.raises(t.T4aError) as e:
        t.contract_pair(a, [1, 2, 3], np.zeros((5, 2)), [1, 9])  # common label 1: 3 vs 5
    assert e.value.code == t.INVALID_ARGUMENT
    with pytest.raises(t.T4aError) as e:
        t.contract_pair(a, [1, 1, 3], np.zeros((5, 2)), [7, 9])
    assert e.value.code == t.INVALID_ARGUMENT
    for left in ([], [1, 2, 3], [8], [1, 1]):
        with pytest.raises(t.T4aError) as e:
            t.tensor_svd(a, [1, 2, 3], left)
        assert e.value.code == t.INVALID_ARGUMENT
    for kw in (dict(max_bond_dim=0), dict(policy=t.SvdTruncationPolicy(float("inf"))), dict(policy=t.SvdTruncationPolicy(-1.0))):
        with pytest.raises(t.T4aError) as e:
            t.tensor_svd(a, [1, 2, 3], [1], **kw)
        assert e.value.code == t.INVALID_ARGUMENT
    with pytest.raises(t.T4aError) as e:
        t.tensor_qr(a, [1, 2, 3], [2], rtol=float("nan"))
    assert e.value.code == t.INVALID_ARGUMENT
    # a well-formed call reaches the device: no CPU fallback
    with pytest.raises(t.T4aError) as e:
        t.tensor_svd(np.ones((2, 2)), [1, 2], [1])
    assert e.value.code == t.NO_DEVICE
    with pytest.raises(t.T4aError) as e:
        t.contract_pair(np.ones((2, 2)), [1, 2], np.ones((2, 2)), [2, 3])
    assert e.value.code == t.NO_DEVICE
    with pytest.raises(t.T4aError) as e:
        t.LabelledTensor(np.ones((2, 2)), [1, 1])
    assert e.value.code == t.INVALID_ARGUMENT
    with pytest.raises(t.T4aError) as e:
        t.LabelledTensor(np.ones((2, 2)), [1, 2])
    assert e.value.code == t.NO_DEVICE
