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


# ---- N-ary contraction (defaults/contract.rs:283-298; reference cases: contract/tests/mod.rs) ----
def _arange_tensor(shape):
    return np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape, order="F")  # make_test_tensor: data = 0, 1, 2, ... column-major


def test_contract_network_reference_cases():
    a, b, c, d = _arange_tensor((2, 3)), _arange_tensor((3, 4)), _arange_tensor((4, 5)), _arange_tensor((5, 6))
    r3, l3 = ob.tensor_contract_many([a, b, c], [[1, 2], [2, 3], [3, 4]])  # test_contract_three :173-183
    assert l3 == [1, 4] and np.array_equal(r3, a @ b @ c)
    r4, l4 = ob.tensor_contract_many([a, b, c, d], [[1, 2], [2, 3], [3, 4], [4, 5]])  # test_contract_four :185-196
    assert l4 == [1, 5] and np.array_equal(r4, a @ b @ c @ d)
    one, lone = ob.tensor_contract_many([a], [[7, 8]])  # test_contract_single :82-87
    assert lone == [7, 8] and np.array_equal(one, a)
    with pytest.raises(ob.OracleError):  # test_contract_empty :75-80
        ob.tensor_contract_many([], [])
    with pytest.raises(ob.OracleError, match="Disconnected tensor network: 2 components"):  # :98-108, :222-233
        ob.tensor_contract_many([a, c], [[1, 2], [3, 4]])


def test_contract_network_retained_indices_like_the_reference():
    # test_contract_with_options_retains_shared_batch_index :235-281
    a = np.arange(1, 13, dtype=np.float64).reshape((2, 2, 3), order="F")
    b = 0.5 * np.arange(1, 13, dtype=np.float64).reshape((2, 3, 2), order="F")
    r, labels = ob.tensor_contract_many([a, b], [[10, 11, 12], [10, 12, 13]], retain=[10])
    assert labels == [10, 11, 13] and np.array_equal(r, np.einsum("bik,bkj->bij", a, b))
    # test_contract_with_options_supports_three_way_retained_label :338-392
    x = np.array([1.0, 2.0, 3.0, 4.0]).reshape((2, 2), order="F")
    y = np.array([5.0, 6.0, 7.0, 8.0, 9.0, 10.0]).reshape((2, 3), order="F")
    z = np.array([11.0, 12.0, 13.0, 14.0]).reshape((2, 2), order="F")
    r, labels = ob.tensor_contract_many([x, y, z], [[20, 21], [20, 22], [20, 23]], retain=[20])
    assert labels == [20, 21, 22, 23] and np.array_equal(r, np.einsum("bi,bj,bk->bijk", x, y, z))
    # test_contract_with_options_retained_index_connects_components :410-449
    r, labels = ob.tensor_contract_many([x, y], [[40, 41], [40, 42]], retain=[40])
    assert labels == [40, 41, 42] and np.array_equal(r, np.einsum("bi,bj->bij", x, y))
    # a three-way label that is NOT retained is summed over all three operands
    r, labels = ob.tensor_contract_many([x, y, z], [[20, 21], [20, 22], [20, 23]])
    assert labels == [21, 22, 23] and np.array_equal(r, np.einsum("bi,bj,bk->ijk", x, y, z))
    with pytest.raises(ob.OracleError):  # test_contract_with_options_errors_for_missing_retained_index :394-408
        ob.tensor_contract_many([np.array([1.0, 2.0]), np.array([3.0, 4.0, 5.0])], [[30], [31]], retain=[32])
