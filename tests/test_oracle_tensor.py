"""CPU oracle of the dense labelled-tensor layer (oracle/t4a_oracle_tensor.hpp) against the reference's own tests
(crates/tensor4all-core/src/defaults/{svd,qr}/tests/mod.rs) and numpy."""
import numpy as np
import pytest

import oracle_binding as ob


def test_svd_retained_rank_fixtures():
    # svd/tests/mod.rs:7-41
    assert ob.svd_retained_rank([], 1e-6) == 1
    assert ob.svd_retained_rank([0.0, 0.0], 1e-6) == 1
    assert ob.svd_retained_rank([5.0, 1e-9], 1e-6) == 1
    assert ob.svd_retained_rank([5.0, 1.0], 1e-12) == 2
    assert ob.svd_retained_rank([5.0, 1.0], 1.5, scale=1) == 1
    assert ob.svd_retained_rank([10.0, 1.0, 0.1], 0.05, measure=1, rule=1) == 1
    assert ob.svd_retained_rank([1.0, 0.1, 0.1], 0.02, scale=1, measure=1, rule=1) == 2


def upper(nrows, ncols, entries):
    d = np.zeros(nrows * ncols)
    for i, j, v in entries:
        d[i + j * nrows] = v
    return d


def test_qr_retained_rank_fixtures():
    # qr/tests/mod.rs:7-17, 159-229
    assert ob.qr_retained_rank([3.0, 0.0, 1.0, 1e-14], 2, 2, 1e-10) == 1
    assert ob.qr_retained_rank([0.0] * 4, 2, 2, 1.0) == 1
    assert ob.qr_retained_rank([], 0, 2, 1e-12) == 1
    r = upper(3, 4, [(0, 0, 10.0), (0, 1, 1.0), (0, 2, 1.0), (0, 3, 1.0), (1, 2, 5.0), (1, 3, 5.0), (2, 3, 1.0)])
    assert ob.qr_retained_rank(r, 3, 4, 1e-15) == 3
    assert ob.qr_retained_rank(upper(3, 3, [(0, 0, 5.0), (0, 1, 3.0), (0, 2, 1.0)]), 3, 3, 1e-15) == 1
    full = upper(3, 3, [(0, 0, 10.0), (0, 1, 1.0), (0, 2, 1.0), (1, 1, 8.0), (1, 2, 1.0), (2, 2, 6.0)])
    assert ob.qr_retained_rank(full, 3, 3, 1e-15) == 3
    tr = upper(3, 3, [(0, 0, 10.0), (0, 1, 0.5), (0, 2, 0.1), (1, 1, 0.01), (2, 2, 0.001)])
    assert ob.qr_retained_rank(tr, 3, 3, 0.01) == 1 and ob.qr_retained_rank(tr, 3, 3, 1e-4) == 2
    assert ob.qr_retained_rank([0.0] * 9, 3, 3, 1e-15) == 1


def test_unfold_order_and_truncating_factorisations():
    # qr/tests/mod.rs:42-60: column-major linearisation survives unit dimensions
    t = np.array([1.0, 2.0, 3.0, 4.0]).reshape((1, 2, 2), order="F")
    q, r = ob.tensor_qr(t, [1, 2, 3], [2, 3], truncate=False)
    assert q.shape == (2, 2, 1) and r.shape == (1, 1)
    assert np.allclose((q.reshape(4, 1, order="F") @ r.reshape(1, 1, order="F")).ravel(), [1, 2, 3, 4])
    # svd/tests/mod.rs:100-118 and qr/tests/mod.rs:102-119
    u, s, v = ob.tensor_svd(np.array([[3.0, 0.0], [0.0, 1.0]]), [10, 11], [10], max_bond_dim=1)
    assert u.shape == (2, 1) and s.shape == (1,) and v.shape == (2, 1) and s[0] == pytest.approx(3.0)
    q, r = ob.tensor_qr(np.array([[1.0, 0.0], [0.0, 1e-14]]), [10, 11], [10], rtol=1e-10)
    assert q.shape == (2, 1) and r.shape == (1, 2)
    for bad in (dict(max_bond_dim=0), dict(threshold=float("inf")), dict(threshold=-1.0)):
        with pytest.raises(ob.OracleError):
            ob.tensor_svd(np.eye(2), [1, 2], [1], **bad)
    with pytest.raises(ob.OracleError):
        ob.tensor_qr(np.eye(2), [1, 2], [1], rtol=float("nan"))
    for left in ([], [1, 2], [3], [1, 1]):
        with pytest.raises(ob.OracleError):
            ob.tensor_svd(np.eye(2), [1, 2], left)


def test_factorisations_reconstruct_random_tensors():
    rng = np.random.default_rng(0)
    t = rng.standard_normal((3, 4, 2, 5))
    labels = [7, 3, 9, 1]
    u, s, v = ob.tensor_svd(t, labels, [9, 7])  # left = (index 9, index 7) in THAT order
    rec = np.einsum("cak,k,bdk->abcd", u, s, v)
    assert np.abs(rec - t).max() < 1e-12
    q, r = ob.tensor_qr(t, labels, [3])
    assert np.abs(np.einsum("bk,kacd->abcd", q, r) - t).max() < 1e-12
    low = np.einsum("ia,ja->ij", rng.standard_normal((6, 2)), rng.standard_normal((8, 2))).reshape(6, 4, 2, order="F")
    u, s, v = ob.tensor_svd(low, [1, 2, 3], [1], threshold=1e-10)
    assert len(s) == 2 and np.abs(np.einsum("ik,k,abk->iab", u, s, v) - low).max() < 1e-12


def test_contract_pair_matches_einsum_and_index_order():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 4, 5))
    b = rng.standard_normal((5, 2, 3))
    c, labels = ob.tensor_contract(a, [1, 2, 3], b, [3, 4, 1])  # common: 1 and 3
    assert labels == [2, 4] and np.abs(c - np.einsum("iaj,jbi->ab", a, b)).max() < 1e-12
    c, labels = ob.tensor_contract(a, [1, 2, 3], b, [6, 7, 8])  # outer product
    assert labels == [1, 2, 3, 6, 7, 8] and c.shape == (3, 4, 5, 5, 2, 3)
    assert np.abs(c - np.einsum("abc,def->abcdef", a, b)).max() < 1e-12
    c, labels = ob.tensor_contract(a, [1, 2, 3], a, [1, 2, 3])  # full contraction -> scalar
    assert labels == [] and c.shape == () and abs(float(c) - np.sum(a * a)) < 1e-10
    with pytest.raises(ob.OracleError):
        ob.tensor_contract(a, [1, 2, 3], b, [1, 4, 6])  # common index 1 has dims 3 vs 5
    with pytest.raises(ob.OracleError):
        ob.tensor_contract(a, [1, 1, 3], b, [6, 7, 8])


@pytest.mark.parametrize("alg", [ob.SVD, ob.QR, ob.LU, ob.CI])
@pytest.mark.parametrize("canonical", [ob.LEFT, ob.RIGHT])
def test_factorize_reconstructs_and_is_canonical(alg, canonical):
    # defaults/factorize.rs: every algorithm returns left [left.., bond] * right [bond, right..] == t
    rng = np.random.default_rng(10 * alg + canonical)
    t = rng.standard_normal((4, 3, 5, 2))
    labels = [1, 2, 3, 4]
    l, r, sv = ob.tensor_factorize(t, labels, [3, 1], alg=alg, canonical=canonical)
    k = l.shape[-1]
    assert l.shape == (5, 4, k) and r.shape == (k, 3, 2) and k == 6
    assert np.abs(np.einsum("cak,kbd->abcd", l, r) - t).max() < 1e-11
    lm, rm = l.reshape(20, k, order="F"), r.reshape(k, 6, order="F")
    if alg == ob.SVD:
        assert sv is not None and np.all(np.diff(sv) <= 0)
        iso = lm.T @ lm if canonical == ob.LEFT else rm @ rm.T
        assert np.abs(iso - np.eye(k)).max() < 1e-11
    if alg == ob.QR:
        assert np.abs(lm.T @ lm - np.eye(k)).max() < 1e-11
    low = np.einsum("ia,ja->ij", rng.standard_normal((12, 2)), rng.standard_normal((10, 2))).reshape(12, 5, 2, order="F")
    l, r, sv = ob.tensor_factorize(low, [7, 8, 9], [7], alg=alg, canonical=canonical, qr_rtol=1e-10)
    assert l.shape[-1] == 2 and np.abs(np.einsum("ik,kab->iab", l, r) - low).max() < 1e-10
    l, r, sv = ob.tensor_factorize(low, [7, 8, 9], [7], alg=alg, canonical=canonical, full_rank=True)
    assert l.shape[-1] == (10 if alg in (ob.SVD, ob.QR) else l.shape[-1])
    assert np.abs(np.einsum("ik,kab->iab", l, r) - low).max() < 1e-10
    if alg != ob.QR:
        l, r, sv = ob.tensor_factorize(t, labels, [3, 1], alg=alg, canonical=canonical, max_bond_dim=3)
        assert l.shape[-1] == 3
