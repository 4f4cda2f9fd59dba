"""GPU tests of the dense labelled-tensor seam through the C ABI: contract_pair, svd_with, qr_with against
the CPU oracle, numpy and the reference's fixtures (crates/tensor4all-core/src/defaults/{svd,qr}/tests/mod.rs)."""
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


def test_reference_fixtures(t4a):
    # qr/tests/mod.rs:42-60, 102-119; svd/tests/mod.rs:100-118
    t = np.array([1.0, 2.0, 3.0, 4.0]).reshape((1, 2, 2), order="F")
    q, r = t4a.tensor_qr(t, [1, 2, 3], [2, 3], truncate=False)
    assert q.shape == (2, 2, 1) and r.shape == (1, 1)
    assert np.allclose((q.reshape(4, 1, order="F") @ r.reshape(1, 1, order="F")).ravel(), [1, 2, 3, 4])
    u, s, v = t4a.tensor_svd(np.array([[3.0, 0.0], [0.0, 1.0]]), [10, 11], [10], max_bond_dim=1)
    assert u.shape == (2, 1) and s.shape == (1,) and v.shape == (2, 1) and s[0] == pytest.approx(3.0)
    q, r = t4a.tensor_qr(np.array([[1.0, 0.0], [0.0, 1e-14]]), [10, 11], [10], rtol=1e-10)
    assert q.shape == (2, 1) and r.shape == (1, 2)


def test_contract_pair_matches_einsum_and_oracle(t4a):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 4, 5))
    b = rng.standard_normal((5, 2, 3))
    c, labels = t4a.contract_pair(a, [1, 2, 3], b, [3, 4, 1])
    oc, ol = ob.tensor_contract(a, [1, 2, 3], b, [3, 4, 1])
    assert labels == ol == [2, 4] and np.abs(c - np.einsum("iaj,jbi->ab", a, b)).max() < 1e-12
    assert np.abs(c - oc).max() < 1e-12
    c, labels = t4a.contract_pair(a, [1, 2, 3], b, [6, 7, 8])
    assert labels == [1, 2, 3, 6, 7, 8] and np.abs(c - np.einsum("abc,def->abcdef", a, b)).max() < 1e-12
    c, labels = t4a.contract_pair(a, [1, 2, 3], a, [1, 2, 3])
    assert labels == [] and abs(float(c) - np.sum(a * a)) < 1e-10
    # a chain of environment-style contractions at tensor-network sizes: (chi, d, chi) cores
    chi, d = 48, 4
    x, y = rng.standard_normal((chi, d, chi)), rng.standard_normal((chi, d, chi))
    c, labels = t4a.contract_pair(x, [0, 1, 2], y, [2, 3, 4])
    assert labels == [0, 1, 3, 4] and np.abs(c - np.einsum("abc,cde->abde", x, y)).max() < 1e-10
    big_a, big_b = rng.standard_normal((7, 3, 11, 2)), rng.standard_normal((2, 5, 3, 6))
    c, labels = t4a.contract_pair(big_a, [10, 20, 30, 40], big_b, [40, 50, 20, 60])
    assert labels == [10, 30, 50, 60] and np.abs(c - np.einsum("ajck,kbjd->acbd", big_a, big_b)).max() < 1e-11


def test_svd_and_qr_reconstruct_and_match_oracle(t4a):
    rng = np.random.default_rng(0)
    t = rng.standard_normal((3, 4, 2, 5))
    labels = [7, 3, 9, 1]
    u, s, v = t4a.tensor_svd(t, labels, [9, 7])
    ou, os_, ov = ob.tensor_svd(t, labels, [9, 7])
    assert u.shape == ou.shape == (2, 3, 6) and v.shape == (4, 5, 6)
    assert np.abs(np.einsum("cak,k,bdk->abcd", u, s, v) - t).max() < 1e-12
    assert np.abs(s - os_).max() < 1e-12 * os_[0]
    uu = u.reshape(6, 6, order="F")
    assert np.abs(uu.T @ uu - np.eye(6)).max() < 1e-12
    q, r = t4a.tensor_qr(t, labels, [3])
    oq, orr = ob.tensor_qr(t, labels, [3])
    assert q.shape == oq.shape == (4, 4) and r.shape == orr.shape == (4, 3, 2, 5)
    assert np.abs(np.einsum("bk,kacd->abcd", q, r) - t).max() < 1e-12
    assert np.abs(q - oq).max() < 1e-11 and np.abs(r - orr).max() < 1e-11
    low = np.einsum("ia,ja->ij", rng.standard_normal((60, 3)), rng.standard_normal((80, 3))).reshape(60, 8, 10, order="F")
    for kw in (dict(policy=t4a.SvdTruncationPolicy(1e-10)), dict(policy=t4a.SvdTruncationPolicy(1e-20), max_bond_dim=3),
               dict(policy=t4a.SvdTruncationPolicy(1e-8, t4a.RELATIVE, t4a.SQUARED_VALUE, t4a.DISCARDED_TAIL_SUM))):
        u, s, v = t4a.tensor_svd(low, [1, 2, 3], [1], **kw)
        assert len(s) == 3 and np.abs(np.einsum("ik,k,abk->iab", u, s, v) - low).max() < 1e-10
    u, s, v = t4a.tensor_svd(low, [1, 2, 3], [3, 2], truncate=False)  # wide unfolding 80 x 60, all 60 kept
    assert len(s) == 60 and np.abs(np.einsum("bak,k,ik->iab", u, s, v) - low).max() < 1e-10
    q, r = t4a.tensor_qr(low, [1, 2, 3], [1], rtol=1e-10)
    assert np.abs(np.einsum("ik,kab->iab", q, r) - low).max() < 1e-10 and q.shape[1] == ob.tensor_qr(low, [1, 2, 3], [1], rtol=1e-10)[0].shape[1]


def test_random_contractions_and_factorisations(t4a):
    rng = np.random.default_rng(7)
    letters = "abcdefghij"
    for case in range(25):
        ra, rb = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        n_common = int(rng.integers(0, min(ra, rb) + 1))
        la = list(rng.choice(10, size=ra, replace=False))
        free_b = [x for x in range(10, 20)]
        lb = list(rng.choice(la, size=n_common, replace=False)) + list(rng.choice(free_b, size=rb - n_common, replace=False))
        lb = [int(x) for x in rng.permutation(lb)]
        la = [int(x) for x in la]
        dim_of = {x: int(rng.integers(1, 5)) for x in set(la) | set(lb)}
        a = rng.standard_normal([dim_of[x] for x in la])
        b = rng.standard_normal([dim_of[x] for x in lb])
        c, labels = t4a.contract_pair(a, la, b, lb)
        names = {x: letters[i] for i, x in enumerate(sorted(set(la)))}
        names.update({x: letters[len(names) + i] for i, x in enumerate(sorted(set(lb) - set(la)))})
        out = [x for x in la if x not in lb] + [x for x in lb if x not in la]
        assert labels == out, (case, la, lb)
        spec = "".join(names[x] for x in la) + "," + "".join(names[x] for x in lb) + "->" + "".join(names[x] for x in out)
        ref = np.einsum(spec, a, b)
        assert c.shape == ref.shape and np.abs(c - ref).max() < 1e-11, (case, spec)
    for case in range(12):
        rank = int(rng.integers(2, 5))
        dims = [int(rng.integers(1, 6)) for _ in range(rank)]
        labels = [int(x) for x in rng.choice(50, size=rank, replace=False)]
        t = rng.standard_normal(dims)
        nl = int(rng.integers(1, rank))
        left = [int(x) for x in rng.choice(labels, size=nl, replace=False)]
        u, s, v = t4a.tensor_svd(t, labels, left, truncate=False)
        perm = [labels.index(x) for x in left] + [k for k in range(rank) if labels[k] not in left]
        mat = np.transpose(t, perm).reshape(int(np.prod([dims[p] for p in perm[:nl]])), -1, order="F")
        k = min(mat.shape)
        um, vm = u.reshape(mat.shape[0], k, order="F"), v.reshape(mat.shape[1], k, order="F")
        assert np.abs((um * s) @ vm.T - mat).max() < 1e-11, (case, dims, labels, left)
        assert np.abs(s - np.linalg.svd(mat, compute_uv=False)).max() < 1e-11
        q, r = t4a.tensor_qr(t, labels, left, truncate=False)
        assert np.abs(q.reshape(mat.shape[0], k, order="F") @ r.reshape(k, mat.shape[1], order="F") - mat).max() < 1e-11


def test_device_resident_tensor_chain(t4a):
    """MPS-style chain on device handles: environment contraction, QR sweep (left-canonical form), SVD truncation —
    no host copies between the steps; checked against numpy at the end."""
    rng = np.random.default_rng(3)
    n, d, chi = 6, 3, 7
    bonds = [1] + [chi] * (n - 1) + [1]
    cores = [rng.standard_normal((bonds[k], d, bonds[k + 1])) for k in range(n)]
    # labels: bond k -> 100 + k, site k -> k
    T = [t4a.LabelledTensor(cores[k], [100 + k, k, 101 + k]) for k in range(n)]
    assert T[2].dims == [chi, d, chi] and T[2].labels == [102, 2, 103]
    # full contraction of the chain into one tensor over the site labels
    full = T[0]
    for k in range(1, n):
        full = full * T[k]
    ref = cores[0]
    for k in range(1, n):
        ref = np.tensordot(ref, cores[k], axes=([-1], [0]))
    assert full.labels == [100] + list(range(n)) + [100 + n]
    assert np.abs(full.to_numpy() - ref).max() < 1e-9 * np.abs(ref).max()
    # norm via environments <psi|psi>: bra labels primed (+1000 on the bonds)
    env = None
    for k in range(n):
        bra = t4a.LabelledTensor(cores[k], [1100 + k, k, 1101 + k])
        env = (T[k] * bra) if env is None else ((env * T[k]) * bra)
    nrm = float(env.to_numpy().ravel()[0])
    assert abs(nrm - np.sum(ref * ref)) < 1e-9 * nrm
    # QR sweep to the right: Q replaces the core, R is absorbed into the next one
    W = list(T)
    for k in range(n - 1):
        q, r = W[k].qr([100 + k, k], 500 + k, truncate=False)
        nxt = r * W[k + 1]                      # [500+k, k+1, 102+k]
        W[k] = q.relabel(500 + k, 101 + k)
        W[k + 1] = nxt.relabel(500 + k, 101 + k).permute([101 + k, k + 1, 102 + k])
        qm = W[k].to_numpy().reshape(-1, W[k].dims[-1], order="F")
        assert np.abs(qm.T @ qm - np.eye(qm.shape[1])).max() < 1e-11
    again = W[0]
    for k in range(1, n):
        again = again * W[k]
    assert np.abs(again.to_numpy() - ref).max() < 1e-9 * np.abs(ref).max()
    # SVD of the two-site tensor at the centre with a rank cap, reconstruction error = discarded weight
    two = T[2] * T[3]                           # [102, 2, 3, 104]
    u, s, v = two.svd([102, 2], 700, 701, policy=t4a.SvdTruncationPolicy(1e-14), max_bond_dim=4)
    assert u.labels == [102, 2, 700] and s.labels == [700] and v.labels == [3, 104, 701] and s.dims == [4]
    sm = np.linalg.svd(two.to_numpy().reshape(chi * d, d * chi, order="F"), compute_uv=False)
    assert np.abs(s.to_numpy() - sm[:4]).max() < 1e-11
    rec = np.einsum("abk,k,cdk->abcd", u.to_numpy(), s.to_numpy(), v.to_numpy())
    assert abs(np.linalg.norm(rec - two.to_numpy()) - np.sqrt(np.sum(sm[4:] ** 2))) < 1e-9
    with pytest.raises(t4a.T4aError):
        T[0].permute([100, 0])
    with pytest.raises(t4a.T4aError):
        T[0].relabel(0, 100)


@pytest.mark.parametrize("alg", [0, 1, 2, 3])
@pytest.mark.parametrize("canonical", [0, 1])
def test_factorize_matches_oracle(t4a, alg, canonical):
    # defaults/factorize.rs:86-834 — SVD / QR / LU / CI with Left / Right canonical forms on device handles
    rng = np.random.default_rng(10 * alg + canonical)
    t = rng.standard_normal((4, 3, 5, 2))
    labels = [1, 2, 3, 4]
    T = t4a.LabelledTensor(t, labels)
    l, r, rank, sv = T.factorize([3, 1], 900, alg=alg, canonical=canonical)
    ol, orr, osv = ob.tensor_factorize(t, labels, [3, 1], alg=alg, canonical=canonical)
    assert l.labels == [3, 1, 900] and r.labels == [900, 2, 4] and rank == 6 == ol.shape[-1]
    assert np.abs(np.einsum("cak,kbd->abcd", l.to_numpy(), r.to_numpy()) - t).max() < 1e-11
    if alg in (2, 3):  # rrLU based: bit-exact pivots -> identical factors up to the triangular solves
        assert np.abs(l.to_numpy() - ol).max() < 1e-10 and np.abs(r.to_numpy() - orr).max() < 1e-10
    if alg == 0:
        assert np.abs(sv - osv).max() < 1e-12 * osv[0]
        lm, rm = l.to_numpy().reshape(20, 6, order="F"), r.to_numpy().reshape(6, 6, order="F")
        iso = lm.T @ lm if canonical == 0 else rm @ rm.T
        assert np.abs(iso - np.eye(6)).max() < 1e-11
    low = np.einsum("ia,ja->ij", rng.standard_normal((40, 3)), rng.standard_normal((30, 3))).reshape(40, 6, 5, order="F")
    L = t4a.LabelledTensor(low, [7, 8, 9])
    l, r, rank, sv = L.factorize([7], 901, alg=alg, canonical=canonical, qr_rtol=1e-10)
    assert rank == 3 == ob.tensor_factorize(low, [7, 8, 9], [7], alg=alg, canonical=canonical, qr_rtol=1e-10)[0].shape[-1]
    assert np.abs(np.einsum("ik,kab->iab", l.to_numpy(), r.to_numpy()) - low).max() < 1e-10
    l, r, rank, sv = L.factorize([7], 902, alg=alg, canonical=canonical, full_rank=True)
    assert np.abs(np.einsum("ik,kab->iab", l.to_numpy(), r.to_numpy()) - low).max() < 1e-10
    if alg != 1:
        l, r, rank, sv = T.factorize([3, 1], 903, alg=alg, canonical=canonical, max_bond_dim=3)
        assert rank == 3 and l.dims == [5, 4, 3]
    with pytest.raises(t4a.T4aError):
        T.factorize([3, 1], 904, alg=7)
    if alg in (2, 3):
        with pytest.raises(t4a.T4aError):
            t4a.LabelledTensor(np.zeros((3, 4)), [1, 2]).factorize([1], 905, alg=alg)


def test_chain_bridge_between_tensor_trains_and_labelled_tensors(t4a):
    # treetn/src/simplett_bridge.rs: tensor_train_to_treetn (:706-794) and treetn_to_tensor_train (:172-277) on the dense seam
    rng = np.random.default_rng(8)
    shapes = [(1, 2, 3), (3, 4, 2), (2, 3, 4), (4, 2, 1)]
    cores = [rng.standard_normal(s) for s in shapes]
    tt = t4a.SimpleTensorTrain(cores)
    ts = t4a.tensor_train_to_tensors(tt, [10, 11, 12, 13], [20, 21, 22])
    assert [t.labels for t in ts] == [[10, 20], [20, 11, 21], [21, 12, 22], [22, 13]]
    assert np.array_equal(ts[0].to_numpy(), cores[0][0]) and np.array_equal(ts[1].to_numpy(), cores[1])
    assert np.array_equal(ts[3].to_numpy(), cores[3][:, :, 0])
    # contract the whole chain on the device and compare with the dense train
    full = ts[0]
    for t in ts[1:]:
        full = full * t
    ref = np.einsum("aib,bjc,ckd,dle->ijkl", *cores)
    assert full.labels == [10, 11, 12, 13] and np.abs(full.to_numpy() - ref).max() < 1e-12
    # back: legs in any order
    shuffled = [ts[0].permute([20, 10]), ts[1].permute([21, 20, 11]), ts[2], ts[3].permute([13, 22])]
    back = t4a.tensors_to_tensor_train(shuffled)
    assert [tuple(d) for d in back.dims()] == shapes
    for s in range(4):
        assert np.array_equal(back.site_tensor(s), cores[s])
    # the doc test of the bridge: a single site
    one = t4a.SimpleTensorTrain([np.array([1.0, 2.0]).reshape(1, 2, 1)])
    (t,) = t4a.tensor_train_to_tensors(one, [5], [])
    assert t.labels == [5] and np.array_equal(t.to_numpy(), [1.0, 2.0])
    assert np.array_equal(t4a.tensors_to_tensor_train([t]).site_tensor(0).ravel(), [1.0, 2.0])
    with pytest.raises(t4a.T4aError):  # not a chain: the first and the last tensor share an index
        t4a.tensors_to_tensor_train([ts[0], ts[1], ts[0]])
    with pytest.raises(t4a.T4aError):  # two site indices on one node
        t4a.tensors_to_tensor_train([full, ts[3]])
    with pytest.raises(t4a.T4aError):
        t4a.tensor_train_to_tensors(tt, [10, 11, 12, 13], [20, 21, 10])


# ---- N-ary contraction, outer product, tensordot (defaults/contract.rs:283-447; reference cases contract/tests/mod.rs) ----
def _arange_tensor(shape):
    return np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape, order="F")


def test_contract_network_reference_cases_on_the_device(t4a):
    a, b, c, d = _arange_tensor((2, 3)), _arange_tensor((3, 4)), _arange_tensor((4, 5)), _arange_tensor((5, 6))
    T = t4a.LabelledTensor
    r = t4a.contract([T(a, [1, 2]), T(b, [2, 3]), T(c, [3, 4])])  # test_contract_three
    assert r.labels == [1, 4] and np.array_equal(r.to_numpy(), a @ b @ c)
    r = t4a.contract([T(a, [1, 2]), T(b, [2, 3]), T(c, [3, 4]), T(d, [4, 5])])  # test_contract_four
    assert r.labels == [1, 5] and np.array_equal(r.to_numpy(), a @ b @ c @ d)
    r = t4a.contract([T(a, [7, 8])])  # test_contract_single
    assert r.labels == [7, 8] and np.array_equal(r.to_numpy(), a)
    with pytest.raises(t4a.T4aError):
        t4a.contract([])
    with pytest.raises(t4a.T4aError, match="Disconnected tensor network: 2 components"):
        t4a.contract([T(a, [1, 2]), T(c, [3, 4])])
    # retained indices (:235-281, :338-392, :410-449, :394-408)
    x3 = np.arange(1, 13, dtype=np.float64).reshape((2, 2, 3), order="F")
    y3 = 0.5 * np.arange(1, 13, dtype=np.float64).reshape((2, 3, 2), order="F")
    r = t4a.contract([T(x3, [10, 11, 12]), T(y3, [10, 12, 13])], retain=[10])
    assert r.labels == [10, 11, 13] and np.array_equal(r.to_numpy(), np.einsum("bik,bkj->bij", x3, y3))
    x = np.array([1.0, 2.0, 3.0, 4.0]).reshape((2, 2), order="F")
    y = np.array([5.0, 6.0, 7.0, 8.0, 9.0, 10.0]).reshape((2, 3), order="F")
    z = np.array([11.0, 12.0, 13.0, 14.0]).reshape((2, 2), order="F")
    r = t4a.contract([T(x, [20, 21]), T(y, [20, 22]), T(z, [20, 23])], retain=[20])
    assert r.labels == [20, 21, 22, 23] and np.array_equal(r.to_numpy(), np.einsum("bi,bj,bk->bijk", x, y, z))
    r = t4a.contract([T(x, [40, 41]), T(y, [40, 42])], retain=[40])
    assert r.labels == [40, 41, 42] and np.array_equal(r.to_numpy(), np.einsum("bi,bj->bij", x, y))
    with pytest.raises(t4a.T4aError):
        t4a.contract([T(np.array([1.0, 2.0]), [30]), T(np.array([3.0, 4.0, 5.0]), [31])], retain=[32])


def test_contract_network_random_networks_match_the_oracle(t4a):
    rng = np.random.default_rng(21)
    T = t4a.LabelledTensor
    cases = [
        # a ring of four with two dangling legs, a star with a hyper-edge (label 5 in three operands), a full contraction to a scalar pair
        ([(3, 4, 2), (4, 5), (5, 6, 3), (6, 3)], [[1, 2, 9], [2, 3], [3, 4, 8], [4, 1]], []),
        ([(4, 3), (4, 5), (4, 2), (3, 5, 2)], [[5, 1], [5, 2], [5, 3], [1, 2, 3]], []),
        ([(4, 3), (4, 5), (4, 2)], [[5, 1], [5, 2], [5, 3]], [5]),
        ([(6, 7), (7, 8), (8, 6)], [[1, 2], [2, 3], [3, 1]], [2]),
        ([(16, 24, 3), (24, 20), (20, 16, 5)], [[1, 2, 7], [2, 3], [3, 1, 8]], []),
    ]
    for shapes, labels, retain in cases:
        arrs = [rng.standard_normal(s) for s in shapes]
        want, wl = ob.tensor_contract_many(arrs, labels, retain)
        got = t4a.contract([T(a, l) for a, l in zip(arrs, labels)], retain=retain)
        assert got.labels == wl
        g = got.to_numpy()
        assert g.shape == want.shape and np.abs(g - want).max() <= 1e-11 * max(1.0, np.abs(want).max())


def test_outer_product_and_tensordot(t4a):
    rng = np.random.default_rng(5)
    T = t4a.LabelledTensor
    a, b = rng.standard_normal((3, 4)), rng.standard_normal((2, 5))
    o = T(a, [1, 2]).outer_product(T(b, [3, 4]))  # test_outer_product_matrix_matrix :198-208
    assert o.labels == [1, 2, 3, 4] and np.allclose(o.to_numpy(), np.einsum("ij,kl->ijkl", a, b), rtol=0, atol=1e-14)
    with pytest.raises(t4a.T4aError):
        T(a, [1, 2]).outer_product(T(b, [2, 4]))
    c = rng.standard_normal((4, 3, 6))
    d = T(a, [1, 2]).tensordot(T(c, [7, 8, 9]), [(2, 7), (1, 8)])  # differently labelled axes, paired explicitly
    assert d.labels == [9] and np.allclose(d.to_numpy(), np.einsum("ij,jik->k", a, c), rtol=0, atol=1e-12)
    with pytest.raises(t4a.T4aError, match="No pairs"):
        T(a, [1, 2]).tensordot(T(c, [7, 8, 9]), [])
    with pytest.raises(t4a.T4aError, match="Dimension mismatch"):
        T(a, [1, 2]).tensordot(T(c, [7, 8, 9]), [(1, 7)])
    with pytest.raises(t4a.T4aError, match="Index not found"):
        T(a, [1, 2]).tensordot(T(c, [7, 8, 9]), [(5, 7)])
    with pytest.raises(t4a.T4aError, match="Batch contraction"):
        T(a, [1, 2]).tensordot(T(c, [2, 1, 9]), [(1, 1)])
