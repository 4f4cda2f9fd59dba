"""GPU parity tests of the tensor-train rows of SURVEY.md §8 (a14-a18), through the C ABI:
t4a_gpu_svd_f64 / _qr_f64 / _full_piv_lu_f64, t4a_gpu_tt_* and t4a_gpu_tci2_from_tensor_train / _to_tensor_train.

Contract: chain contractions (evaluate, sum, norm2, evaluate_many) are bit-identical to the CPU oracle
(same summation order, no FMA contraction); pivot selections / index sets / bond dimensions are exact; values
produced through tenferro-level ops in the reference (gemm, svd, qr) agree to 1e-10 (stated at each assert).
"""
import math

import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_tt import (dense, random_tt, tt_preserves_values, tt_rank3, tt_two_scale)

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


# ---------------------------------------------------------------------------------------------- a14
@pytest.mark.parametrize("shape", [(2, 2), (5, 3), (3, 5), (16, 8), (8, 16), (64, 64), (130, 40), (300, 200),
                                   (200, 300), (1, 4), (4, 1), (512, 256)])
def test_svd_properties_and_oracle_singular_values(t4a, shape):
    a = RNG.standard_normal(shape)
    k = min(shape)
    u, s, vt = t4a.svd_backend(a)
    assert u.shape == (shape[0], k) and s.shape == (k,) and vt.shape == (k, shape[1])
    assert np.all(np.diff(s) <= 0) and np.all(s >= 0)
    scale = s[0]
    assert np.abs((u * s) @ vt - a).max() < 1e-10 * scale       # reference: reconstruction to 1e-10
    assert np.abs(u.T @ u - np.eye(k)).max() < 1e-11 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-11
    if max(shape) <= 130:
        _, so, _ = ob.svd(a)
        assert np.abs(s - so).max() < 1e-12 * scale
    else:
        assert np.abs(s - np.linalg.svd(a, compute_uv=False)).max() < 1e-11 * scale


def test_svd_reference_matrix_and_rank_deficient(t4a):
    a = np.array([[1.0, 2.0], [3.0, 4.0]])
    u, s, vt = t4a.svd_backend(a)
    assert np.abs((u * s) @ vt - a).max() < 1e-10
    low = RNG.standard_normal((40, 3)) @ RNG.standard_normal((3, 20))
    u, s, vt = t4a.svd_backend(low)
    assert np.abs((u * s) @ vt - low).max() < 1e-10 * s[0]
    assert s[3] < 1e-12 * s[0]
    z = np.zeros((6, 4))
    u, s, vt = t4a.svd_backend(z)
    assert np.all(s == 0)
    assert np.abs(u.T @ u - np.eye(4)).max() < 1e-12 and np.abs(vt @ vt.T - np.eye(4)).max() < 1e-12
    # exactly repeated columns: one singular value is exactly zero -> completion path
    rep = np.column_stack([np.arange(1.0, 6.0), np.arange(1.0, 6.0), np.ones(5)])
    u, s, vt = t4a.svd_backend(rep)
    assert np.abs((u * s) @ vt - rep).max() < 1e-10 * s[0]
    assert np.abs(u.T @ u - np.eye(3)).max() < 1e-10


def test_svd_rejects_non_finite_and_empty(t4a):
    bad = np.ones((4, 3))
    bad[2, 1] = np.nan
    with pytest.raises(t4a.T4aError) as e:
        t4a.svd_backend(bad)
    assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError):
        t4a.svd_backend(np.zeros((0, 3)))


@pytest.mark.parametrize("shape", [(2, 2), (5, 3), (3, 5), (16, 8), (64, 64), (300, 120), (120, 300), (1, 4), (4, 1),
                                   (512, 256), (700, 200), (560, 33), (561, 40)])
def test_qr_properties(t4a, shape):
    """(Panels of at most 560 rows are factorised in the LDS, taller ones in global memory: 700 x 200 starts with the second kind and
    ends with the first, 560 / 561 rows sit on the limit.)"""
    a = RNG.standard_normal(shape)
    k = min(shape)
    q, r = t4a.qr_backend(a)
    assert q.shape == (shape[0], k) and r.shape == (k, shape[1])
    assert np.abs(q @ r - a).max() < 1e-10 * np.abs(a).max() * math.sqrt(max(shape))
    assert np.abs(q.T @ q - np.eye(k)).max() < 1e-11
    assert np.abs(np.tril(r, -1)).max() == 0.0
    if max(shape) <= 64:  # same Householder convention as the oracle
        qo, ro = ob.qr(a)
        assert np.abs(q - qo).max() < 1e-10 and np.abs(r - ro).max() < 1e-10 * np.abs(ro).max()


def test_qr_reference_matrix_and_zero_column(t4a):
    a = np.array([[1.0, 2.0], [3.0, 4.0]])
    q, r = t4a.qr_backend(a)
    assert np.abs(q @ r - a).max() < 1e-10
    z = np.array([[0.0, 1.0, 2.0], [0.0, 3.0, 4.0], [0.0, 5.0, 7.0]])
    q, r = t4a.qr_backend(z)
    assert np.abs(q @ r - z).max() < 1e-12 and np.abs(q.T @ q - np.eye(3)).max() < 1e-12


@pytest.mark.parametrize("n", [1, 2, 7, 40, 129])
def test_full_piv_lu_matches_oracle(t4a, n):
    a = RNG.standard_normal((n, n))
    p, l, u, q = t4a.full_piv_lu_matrix(a)
    po, lo, uo, qo = ob.full_piv_lu(a)
    assert np.array_equal(p, po) and np.array_equal(q, qo)      # pivot order is bit-exact
    assert np.array_equal(l, lo) and np.array_equal(u, uo)      # and so is the elimination arithmetic
    assert np.abs(p @ a @ q.T - l @ u).max() < 1e-12 * n
    sing = np.outer(np.arange(1.0, 5.0), np.arange(2.0, 6.0))   # rank 1: L completed by identity columns
    p, l, u, q = t4a.full_piv_lu_matrix(sing)
    po, lo, uo, qo = ob.full_piv_lu(sing)
    assert np.array_equal(p, po) and np.array_equal(q, qo) and np.array_equal(l, lo) and np.array_equal(u, uo)
    assert np.abs(p @ sing @ q.T - l @ u).max() < 1e-12


# ---------------------------------------------------------------------------------------------- a15
@pytest.mark.parametrize("site_dims,chi", [([2, 3, 2, 4], 3), ([2] * 10, 8), ([3, 1, 2], 2), ([2] * 6, 70)])
def test_evaluate_sum_norm2_bit_identical_to_oracle(t4a, site_dims, chi):
    cores = random_tt(site_dims, chi, np.random.default_rng(chi))
    g = t4a.SimpleTensorTrain(cores)
    o = ob.OracleTT(cores)
    assert len(g) == len(site_dims) and g.site_dims() == site_dims and g.link_dims() == o.link_dims()
    idx = np.stack([RNG.integers(0, d, size=257) for d in site_dims], axis=1)
    assert np.array_equal(g.evaluate(idx), o.evaluate(idx))
    assert g.sum() == o.sum()
    assert g.norm2() == o.norm2()
    for s in range(len(site_dims)):
        assert np.array_equal(g.site_tensor(s), cores[s])


def test_tt_new_validation_and_errors(t4a):
    with pytest.raises(t4a.T4aError) as e:
        t4a.SimpleTensorTrain([np.ones((2, 2, 1)), np.ones((1, 2, 1))])  # first left dim must be 1
    assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError):
        t4a.SimpleTensorTrain([np.ones((1, 2, 3)), np.ones((2, 2, 1))])  # bond mismatch
    tt = t4a.SimpleTensorTrain.constant([2, 3, 2], 2.5)
    with pytest.raises(t4a.T4aError):
        tt.evaluate([[0, 3, 0]])
    assert abs(tt.evaluate([[1, 2, 1]])[0] - 2.5) == 0.0 and tt.sum() == 30.0
    c = tt.clone()
    assert c.sum() == 30.0 and c.link_dims() == [1, 1]


# ---------------------------------------------------------------------------------------------- a16
@pytest.mark.parametrize("method", [0, 1, 2])
def test_compress_reference_cases(t4a, method):
    tt = t4a.SimpleTensorTrain.constant([2, 3, 2], 1.0)
    tt.compress(method=method)
    assert abs(tt.sum() - 12.0) < 1e-10
    cores = tt_preserves_values()
    src = t4a.SimpleTensorTrain(cores)
    full = src.full_tensor()
    tt = src.clone()
    tt.compress(method=method)
    o = ob.OracleTT(cores)
    o.compress(method=method)
    assert tt.link_dims() == o.link_dims()
    assert abs(tt.sum() - src.sum()) < 1e-8 and np.abs(tt.full_tensor() - full).max() < 1e-10


@pytest.mark.parametrize("method", [0, 2])
def test_compress_with_max_bond_dim(t4a, method):
    tt = t4a.SimpleTensorTrain(tt_rank3())
    n0 = tt.norm()
    tt.compress(method=method, max_bond_dim=2, tolerance=1e-12)
    assert tt.rank() <= 2 and abs(n0 - tt.norm()) < 0.1 * n0
    o = ob.OracleTT(tt_rank3())
    o.compress(method=method, max_bond_dim=2, tolerance=1e-12)
    assert np.abs(tt.full_tensor() - o.full_tensor()).max() < 1e-10 * n0


def test_compress_normalize_error_threshold(t4a):
    cores = tt_two_scale()
    rel = t4a.SimpleTensorTrain(cores)
    assert rel.rank() == 2
    rel.compress(method=2, tolerance=1e-6, normalize_error=True)
    assert rel.rank() == 1
    ab = t4a.SimpleTensorTrain(cores)
    ab.compress(method=2, tolerance=1e-6, normalize_error=False)
    assert ab.rank() == 2
    idx = [[l, r] for l in range(2) for r in range(2)]
    assert np.abs(ab.evaluate(idx) - t4a.SimpleTensorTrain(cores).evaluate(idx)).max() < 1e-6


@pytest.mark.parametrize("method", [2, 0, 1])
def test_cfg1_compress_roundtrip_matches_oracle(t4a, method):
    """BASELINE config 1: d=10, d_loc=2, chi=8 random train, compress(tol 1e-12), all 1024 values to 1e-10."""
    cores = random_tt([2] * 10, 8, np.random.default_rng(1))
    src = t4a.SimpleTensorTrain(cores)
    full = src.full_tensor()
    tt = src.clone()
    tt.compress(method=method, tolerance=1e-12)
    o = ob.OracleTT(cores)
    o.compress(method=method, tolerance=1e-12)
    assert tt.link_dims() == o.link_dims() == [2, 4, 8, 8, 8, 8, 8, 4, 2]
    scale = np.abs(full).max()
    assert np.abs(tt.full_tensor() - full).max() < 1e-10 * scale
    assert np.abs(tt.full_tensor() - o.full_tensor()).max() < 1e-10 * scale
    if method in (0, 1):  # LU / CI: cores themselves agree (same pivots, tolerance-level gemm)
        for a, b in zip(tt.site_tensors(), o.cores()):
            assert a.shape == b.shape and np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("method", [0, 2])
def test_compress_truncates_larger_train(t4a, method):
    """chi = 40 train that is exactly rank 6 inside: compress recovers the bond dimension 6."""
    rng = np.random.default_rng(5)
    n, chi, r = 8, 40, 6
    small = random_tt([2] * n, r, rng)
    cores = []
    for i, c in enumerate(small):  # embed into chi via random isometries on the bonds
        lq = np.eye(1) if i == 0 else qs
        qs = np.eye(1) if i == n - 1 else np.linalg.qr(rng.standard_normal((chi, r)))[0]
        cores.append(np.einsum("al,lsr,br->asb", lq, c, qs))
    src = t4a.SimpleTensorTrain(cores)
    assert src.rank() == chi
    tt = src.clone()
    tt.compress(method=method, tolerance=1e-10)
    assert tt.link_dims() == [2, 4, 6, 6, 6, 4, 2]
    full = src.full_tensor()
    assert np.abs(tt.full_tensor() - full).max() < 1e-9 * np.abs(full).max()


def test_rank_zero_bond_matrix_is_refused_by_lu_and_ci_compress_and_by_conversion(t4a):
    """DESIGN.md section 2, "a known device / oracle difference".  rrlu_mut stops at `p <= EPS` — an ABSOLUTE 2.2e-16 — before the first
    pivot (matrixlu.rs:735-819), and `factorize` hands the LU / CI pivot count through unclamped (compression.rs:165-227; only the SVD branch
    has `rank.max(1)`, :305): a bond matrix whose largest entry is <= 2.2e-16 — an identically zero core, or a train holding values of
    1e-20 — means bond dimension 0 and 0-row products in the un-vendored backend.  No test of the reference goes there.  The oracle follows
    that reading (link dimensions 0: the train becomes the zero function, values of 1e-20 included); the DEVICE refuses: INVALID_ARGUMENT,
    the train still a train of the same tensor (cores in front of the refused bond may have been re-factored), the handle usable.  SVD compression of the same trains is the oracle's, and so is everything for a train with a
    zero slice but bond matrices above the threshold."""
    rng = np.random.default_rng(3)
    zero_core = [np.zeros((1, 1, 4)), rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))]
    zero_mid = [rng.uniform(-1, 1, (1, 3, 4)), np.zeros((4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))]
    tiny = [rng.uniform(-1, 1, (1, 3, 4)) * 1e-20, rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))]
    for cores in (zero_core, zero_mid, tiny):
        full = ob.OracleTT(cores).full_tensor()
        for method in (0, 1):
            g = t4a.SimpleTensorTrain(cores)
            with pytest.raises(t4a.T4aError) as e:
                g.compress(method=method, tolerance=1e-10)
            assert e.value.code == t4a.INVALID_ARGUMENT and "zero bond matrix" in str(e.value)
            # (like the reference's in-place compress behind `?`, the sweep may have re-factored the cores in front of the refused bond:
            #  the train is still a valid train of the SAME tensor, not necessarily the same cores)
            assert len(g.link_dims()) == 2 and 0 not in g.link_dims()
            assert np.abs(g.full_tensor() - full).max() <= 1e-14 * max(np.abs(full).max(), 1e-300)
            o = ob.OracleTT(cores)
            o.compress(method=method, tolerance=1e-10)
            assert 0 in o.link_dims()
        g = t4a.SimpleTensorTrain(cores)
        g.compress(method=2, tolerance=1e-10)
        o = ob.OracleTT(cores)
        o.compress(method=2, tolerance=1e-10)
        assert g.link_dims() == o.link_dims() and 0 not in g.link_dims()
        assert np.abs(g.full_tensor() - full).max() <= 1e-12 * max(np.abs(full).max(), 1e-300)
    with pytest.raises(t4a.T4aError) as e:
        t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain(zero_core))
    assert e.value.code == t4a.INVALID_ARGUMENT
    part = [rng.uniform(-1, 1, (1, 3, 4)), rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))]
    part[1][:, 1, :] = 0.0
    for method in (0, 1, 2):
        g = t4a.SimpleTensorTrain(part)
        g.compress(method=method, tolerance=1e-10)
        o = ob.OracleTT(part)
        o.compress(method=method, tolerance=1e-10)
        assert g.link_dims() == o.link_dims()
        assert np.abs(g.full_tensor() - o.full_tensor()).max() < 1e-10


# ---------------------------------------------------------------------------------------------- a17
def test_evaluate_many_reference_cases(t4a):
    tt = t4a.SimpleTensorTrain.constant([2, 3, 2], 2.0)
    idx = [[0, 0, 0], [0, 1, 0], [1, 2, 1], [0, 0, 1]]
    v, split = tt.evaluate_many(idx, return_split=True)
    assert split == 2 and v.shape == (4,) and np.abs(v - 2.0).max() < 1e-10
    assert tt.evaluate_many(np.zeros((0, 3), dtype=int)).size == 0
    v = tt.evaluate_many(idx, split=1)
    assert np.abs(v - 2.0).max() < 1e-10
    two = t4a.SimpleTensorTrain.constant([2, 2], 1.0)
    for bad in (0, 10):
        with pytest.raises(t4a.T4aError):
            two.evaluate_many([[0, 0], [1, 1]], split=bad)
    with pytest.raises(t4a.T4aError):
        two.evaluate_many([[0]], split=1)
    with pytest.raises(t4a.T4aError):
        two.evaluate_many([[0, 0, 0]], split=1)


@pytest.mark.parametrize("site_dims,chi,npts", [([2, 3, 2, 2, 3, 2], 4, 300), ([2] * 20, 32, 5000),
                                                ([4, 4, 4], 9, 64)])
def test_evaluate_many_bit_identical_to_oracle(t4a, site_dims, chi, npts):
    cores = random_tt(site_dims, chi, np.random.default_rng(chi + npts))
    g = t4a.SimpleTensorTrain(cores)
    o = ob.OracleTT(cores)
    n = len(site_dims)
    # few distinct halves so that the unique-environment path is exercised
    base = np.stack([RNG.integers(0, d, size=40) for d in site_dims], axis=1)
    idx = base[RNG.integers(0, 40, size=npts)].copy()
    idx[:, n // 2:] = base[RNG.integers(0, 40, size=npts)][:, n // 2:]
    for split in (None, 1, n // 2, n - 1, n):
        v, used = g.evaluate_many(idx, split=split, return_split=True)
        vo, used_o = o.evaluate_many(idx, split)
        assert used == used_o
        assert np.array_equal(v, vo), f"split={split}"


# ---------------------------------------------------------------------------------------------- a18
def assert_conversion_matches_oracle(t4a, cores, **kw):
    g = t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain(cores), **kw)
    res = ob.OracleTT(cores).to_tci2(**kw)
    n = len(cores)
    for p in range(n):
        assert [tuple(int(x) for x in e) for e in g.i_set(p)] == res["i_set"][p], f"I set differs at site {p}"
        assert [tuple(int(x) for x in e) for e in g.j_set(p)] == res["j_set"][p], f"J set differs at site {p}"
        a, b = g.site_tensor(p), res["cores"][p]
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), f"core differs at site {p}"
    pe = g.pivot_errors()
    assert len(pe) == len(res["pivot_errors"])
    assert np.abs(pe - res["pivot_errors"]).max() <= 1e-10 * max(1.0, np.abs(res["pivot_errors"]).max())
    assert abs(g.max_sample_value() - res["max_sample_value"]) <= 1e-10 * max(1.0, res["max_sample_value"])
    assert np.all(g.bond_errors() == 0.0)
    return g, res


def test_from_tensor_train_reference_cases(t4a):
    tt = t4a.SimpleTensorTrain.constant([2, 3, 2], 2.5)
    tci = t4a.TensorCI2.from_tensor_train(tt)
    rt = tci.to_tensor_train()
    assert abs(rt.evaluate([[1, 2, 1]])[0] - 2.5) < 1e-12
    assert tci.link_dims() == [1, 1]
    tci = t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain.constant([2, 2, 2], 1.0), max_bond_dim=1)
    assert all(d <= 1 for d in tci.link_dims())
    with pytest.raises(t4a.T4aError):
        t4a.TensorCI2.from_tensor_train(tt, tolerance=-1.0)
    with pytest.raises(t4a.T4aError):
        t4a.TensorCI2.from_tensor_train(tt, max_iter=1)
    with pytest.raises(t4a.T4aError):
        t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain.constant([2], 1.0))


def test_from_tensor_train_nontrivial_and_lorentz(t4a):
    from test_oracle_tt import _tci_source
    src = _tci_source(lambda i: (i[0] + 1.0) * (i[1] + 2.0) + (i[2] + 3.0), [3, 3, 3], [2, 2, 2], tolerance=1e-12,
                      max_iter=10)
    cores = [src.site_tensor(p) for p in range(3)]
    g, _ = assert_conversion_matches_oracle(t4a, cores)
    full = ob.OracleTT(cores).full_tensor()
    assert g.link_dims() == src.link_dims()
    assert np.abs(g.to_tensor_train().full_tensor() - full).max() < 1e-10
    src = _tci_source(lambda i: 1.0 / (1.0 + sum((x + 1.0) ** 2 for x in i)), [4] * 4, [0] * 4, tolerance=1e-12,
                      max_iter=20, max_bond_dim=5)
    cores = [src.site_tensor(p) for p in range(4)]
    # The Lorentzian is symmetric under site permutations, so pivot candidates tie exactly in exact arithmetic and
    # the winner is decided by the rounding of the carried GEMM (tenferro-level, "parity unpinned"): only the
    # reference's own assertions (conversion/tests/mod.rs:80-88) apply, not set-by-set equality with the oracle.
    g = t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain(cores), tolerance=1e-12, max_bond_dim=5)
    assert g.link_dims() == src.link_dims()
    assert np.abs(g.to_tensor_train().full_tensor() - ob.OracleTT(cores).full_tensor()).max() < 1e-10
    for p in range(3):  # nested sets
        ip, ip1 = [tuple(e) for e in g.i_set(p)], [tuple(e) for e in g.i_set(p + 1)]
        jp, jp1 = [tuple(e) for e in g.j_set(p)], [tuple(e) for e in g.j_set(p + 1)]
        assert all(e[:-1] in ip for e in ip1) and all(e[1:] in jp1 for e in jp)


@pytest.mark.parametrize("max_iter", [2, 3, 6])
def test_from_tensor_train_random_train_matches_oracle(t4a, max_iter):
    cores = random_tt([2, 3, 2, 3, 2, 2, 3], 5, np.random.default_rng(11))
    g, res = assert_conversion_matches_oracle(t4a, cores, max_iter=max_iter)
    full = ob.OracleTT(cores).full_tensor()
    assert np.abs(g.to_tensor_train().full_tensor() - full).max() < 1e-10 * np.abs(full).max()
    # the converted object is a live TensorCI2: a 2-site sweep on the exact function keeps the error at zero
    tt = ob.OracleTT(cores)
    g.set_function(lambda i: float(tt.evaluate([list(i)])[0]))
    g.sweep2site(True, t4a.TCI2Options(tolerance=1e-12, nsearch=0, max_nglobal_pivot=0))
    assert g.max_bond_error() < 1e-10 * np.abs(full).max()


# ---- arithmetic: add / sub / scale / reverse / partial_sum (simplett/src/arithmetic.rs, tensortrain.rs:264-583) ----
def _grid_pts(dims):
    import itertools
    return np.array(list(itertools.product(*[range(d) for d in dims])), dtype=np.uint32)


def _rank1_cores(*vectors):
    return [np.asarray(v, dtype=float).reshape(1, -1, 1) for v in vectors]


def test_tt_arithmetic_reference_fixtures(t4a):
    # tensortrain/tests/mod.rs:64-90, :297-403, :441-512
    tt = t4a.SimpleTensorTrain([np.ones((1, 2, 1)), np.ones((1, 2, 1))])
    tt.scale_mut(3.0)
    assert abs(tt.sum() - 12.0) < 1e-10
    tt = t4a.SimpleTensorTrain(_rank1_cores([1, 2], [1, 2, 3]))
    rev = tt.reverse()
    assert rev.site_dims() == [3, 2] and rev.evaluate([[2, 1]])[0] == tt.evaluate([[1, 2]])[0]
    ones = t4a.SimpleTensorTrain([np.ones((1, 2, 1)), np.ones((1, 3, 1)), np.ones((1, 2, 1))])
    r = ones.partial_sum([0, 1, 2])
    assert len(r) == 1 and abs(r.sum() - 12.0) < 1e-12
    r = ones.partial_sum([])
    assert len(r) == 3 and np.abs(r.evaluate(_grid_pts([2, 3, 2])) - 1.0).max() < 1e-12
    tt = t4a.SimpleTensorTrain(_rank1_cores([1, 2, 3], [1, 2, 3, 4], [1, 2]))
    r = tt.partial_sum([1])
    assert len(r) == 2 and np.abs(r.evaluate(_grid_pts([3, 2])) - np.array([(1 + i) * 10.0 * (1 + k) for i, k in _grid_pts([3, 2])])).max() < 1e-10
    r = tt.partial_sum([0, 2])
    assert len(r) == 1 and np.abs(r.evaluate(_grid_pts([4])) - 18.0 * np.arange(1, 5)).max() < 1e-10
    a = t4a.SimpleTensorTrain(_rank1_cores([1, 2, 3], [1, 1, 1], [1, 1]))
    b = t4a.SimpleTensorTrain(_rank1_cores([1, 1, 1], [1, 2, 3], [1, 1]))
    pts = _grid_pts([3, 3, 2])
    assert np.abs((a + b).evaluate(pts) - np.array([2.0 + i + j for i, j, k in pts])).max() < 1e-12
    assert np.abs((a - b).evaluate(pts) - np.array([float(i) - float(j) for i, j, k in pts])).max() < 1e-12
    assert np.abs(a.scale(2.5).evaluate([[i, 0, 0] for i in range(3)]) - 2.5 * np.arange(1, 4)).max() < 1e-12
    one = t4a.SimpleTensorTrain(_rank1_cores([1, 2])) + t4a.SimpleTensorTrain(_rank1_cores([10, 20]))
    assert np.array_equal(one.evaluate([[0], [1]]), [11.0, 22.0]) and [tuple(d) for d in one.dims()] == [(1, 2, 1)]
    with pytest.raises(t4a.T4aError):
        a + t4a.SimpleTensorTrain(_rank1_cores([1, 2, 3], [1, 1, 1]))
    with pytest.raises(t4a.T4aError):
        a + t4a.SimpleTensorTrain(_rank1_cores([1, 2, 3], [1, 1], [1, 1]))
    with pytest.raises(t4a.T4aError):
        tt.partial_sum([3])


def test_tt_arithmetic_matches_oracle_bitwise(t4a):
    rng = np.random.default_rng(21)
    sa = [(1, 2, 3), (3, 4, 5), (5, 3, 4), (4, 2, 2), (2, 3, 1)]
    sb = [(1, 2, 2), (2, 4, 6), (6, 3, 3), (3, 2, 4), (4, 3, 1)]
    ca, cb = [rng.standard_normal(s) for s in sa], [rng.standard_normal(s) for s in sb]
    da, db = t4a.SimpleTensorTrain(ca), t4a.SimpleTensorTrain(cb)
    oa, ob_ = ob.OracleTT(ca), ob.OracleTT(cb)

    def same(d, o):
        oc = o.cores()
        assert [tuple(x) for x in d.dims()] == [c.shape for c in oc]
        for s, c in enumerate(oc):
            assert np.array_equal(d.site_tensor(s), c), s
    same(da.add(db), oa.add(ob_))
    same(da.sub(db), oa.sub(ob_))
    same(da.scale(-0.37), oa.scale(-0.37))
    same(da.reverse(), oa.reverse())
    for dims in ([], [0], [4], [1, 3], [0, 1], [3, 4], [0, 2, 4], [0, 1, 2, 3, 4]):
        same(da.partial_sum(dims), oa.partial_sum(dims))
    # add then compress: the usual pipeline around the sweep
    s = da.add(db)
    dense = np.einsum("aib,bjc,ckd,dle,emf->ijklm", *ca) + np.einsum("aib,bjc,ckd,dle,emf->ijklm", *cb)
    assert np.abs(s.evaluate(_grid_pts([2, 4, 3, 2, 3])).reshape(dense.shape) - dense).max() < 1e-12
    s.compress(tolerance=1e-13)
    assert np.abs(s.evaluate(_grid_pts([2, 4, 3, 2, 3])).reshape(dense.shape) - dense).max() < 1e-10


def test_tt_inner_product(t4a):
    # contraction/tests/mod.rs:8-68, :177-185 and parity with the oracle / the dense sum
    c = t4a.SimpleTensorTrain.constant
    assert abs(c([2, 3], 2.0).inner_product(c([2, 3], 3.0)) - 36.0) < 1e-10
    assert abs(c([2, 3, 2], 1.0).inner_product(c([2, 3, 2], 2.0)) - 24.0) < 1e-10
    rng = np.random.default_rng(31)
    sa = [(1, 2, 7), (7, 3, 20), (20, 2, 33), (33, 4, 9), (9, 2, 1)]
    sb = [(1, 2, 5), (5, 3, 18), (18, 2, 40), (40, 4, 6), (6, 2, 1)]
    ca, cb = [rng.standard_normal(s) for s in sa], [rng.standard_normal(s) for s in sb]
    da, db = t4a.SimpleTensorTrain(ca), t4a.SimpleTensorTrain(cb)
    pts = _grid_pts([2, 3, 2, 4, 2])
    exact = float(np.dot(da.evaluate(pts), db.evaluate(pts)))
    got, ref = da.inner_product(db), ob.OracleTT(ca).inner_product(ob.OracleTT(cb))
    assert abs(got - ref) < 1e-10 * abs(ref) and abs(got - exact) < 1e-10 * abs(exact)
    assert abs(da.inner_product(da) - da.norm2()) < 1e-10 * da.norm2()
    with pytest.raises(t4a.T4aError):
        da.inner_product(c([2, 3], 1.0))
    with pytest.raises(t4a.T4aError):
        da.inner_product(c([2, 3, 2, 4, 3], 1.0))


def test_svd_tiny_and_rank_deficient_matrices_on_both_jacobi_paths(t4a):
    """n <= 16 columns run the single-launch Jacobi (one workgroup, no host round trip per sweep), wider matrices the
    blocked tournament.  2 x 2 ... 16 x 16 and 17 ... 40 columns, tall / wide, rank-deficient (outer products, repeated columns, zero
    matrices): singular values against LAPACK to 1e-10 of the largest, U S Vt reconstructs A, U and V orthonormal on the numerical
    range (svd_backend, tensor4all-tensorbackend/src/backend.rs:709-742)."""
    rng = np.random.default_rng(21)
    shapes = [(2, 2), (3, 2), (2, 5), (4, 4), (7, 3), (8, 8), (16, 16), (40, 16), (16, 9), (17, 17), (33, 20), (20, 40), (64, 24)]
    for (m, n) in shapes:
        k = min(m, n)
        cases = [rng.standard_normal((m, n)),
                 np.outer(rng.standard_normal(m), rng.standard_normal(n)),                       # rank 1
                 rng.standard_normal((m, max(1, k // 2))) @ rng.standard_normal((max(1, k // 2), n)),  # rank k / 2
                 np.zeros((m, n))]
        dup = rng.standard_normal((m, n))
        dup[:, -1] = dup[:, 0]                                                                   # a repeated column
        cases.append(dup)
        for a in cases:
            u, s_, vt = t4a.svd_backend(a)
            ref = np.linalg.svd(a, compute_uv=False)
            scale = max(ref[0], 1e-300) if ref.size else 1.0
            assert u.shape == (m, k) and s_.shape == (k,) and vt.shape == (k, n)
            assert np.all(np.diff(s_) <= 1e-12 * max(scale, 1.0)) and np.all(s_ >= 0.0)
            assert np.abs(s_ - ref).max() <= 1e-10 * max(scale, 1.0), (m, n)
            assert np.abs((u * s_) @ vt - a).max() <= 1e-10 * max(scale, 1.0), (m, n)
            r = int((ref > 1e-10 * max(scale, 1e-300)).sum())
            if r:
                assert np.abs(u[:, :r].T @ u[:, :r] - np.eye(r)).max() <= 1e-10, (m, n)
                assert np.abs(vt[:r] @ vt[:r].T - np.eye(r)).max() <= 1e-10, (m, n)


def test_randomized_svd_against_the_dense_svd(t4a):
    """t4a_gpu_rsvd_f64 (north_star: "one-sided Jacobi / randomized SVD"): exact to rounding on matrices of rank <= k, and within
    the range-finder bound on a decaying spectrum; orthonormal factors; argument errors."""
    rng = np.random.default_rng(17)
    # rank-20 matrix, k = 20: the thin SVD of the matrix itself
    a = rng.standard_normal((300, 20)) @ rng.standard_normal((20, 180))
    u, s, vt = t4a.randomized_svd(a, 20, oversample=8, power_iters=1, seed=3)
    s_ref = np.linalg.svd(a, compute_uv=False)[:20]
    assert np.abs(s - s_ref).max() <= 1e-10 * s_ref[0]
    assert np.abs(u @ np.diag(s) @ vt - a).max() <= 1e-10 * np.abs(a).max() * 20
    assert np.abs(u.T @ u - np.eye(20)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(20)).max() < 1e-10
    assert np.all(np.diff(s) <= 1e-12 * s[0])
    # decaying spectrum sigma_i = 2^-i: rank-12 truncation with two power iterations
    q1, _ = np.linalg.qr(rng.standard_normal((256, 64)))
    q2, _ = np.linalg.qr(rng.standard_normal((200, 64)))
    sig = 2.0 ** -np.arange(64)
    b = (q1 * sig) @ q2.T
    u, s, vt = t4a.randomized_svd(b, 12, oversample=10, power_iters=2, seed=5)
    assert np.abs(s - sig[:12]).max() <= 1e-6 * sig[0]
    assert np.linalg.norm(u @ np.diag(s) @ vt - b, 2) <= 1.5 * sig[12]
    # the same seed gives the same factors
    u2, s2, vt2 = t4a.randomized_svd(b, 12, oversample=10, power_iters=2, seed=5)
    assert np.array_equal(s, s2) and np.array_equal(u, u2)
    with pytest.raises(t4a.T4aError):
        t4a.randomized_svd(a, 0)
    with pytest.raises(t4a.T4aError):
        t4a.randomized_svd(a, 181)


@pytest.mark.parametrize("shape", [(67, 67), (129, 65), (65, 129), (255, 77), (333, 131), (40, 33), (33, 40), (17, 9)])
def test_svd_and_qr_on_odd_and_ragged_shapes(t4a, shape):
    """Round 5 kernels: the blocked Jacobi round keeps PAIRS of rows per lane (odd column lengths are padded by a zero row), its last
    column block is partial when the column count is no multiple of the block width, the V update runs on 16-row MFMA tiles (row counts
    off a multiple of 16); the QR panel kernel is specialised by rows per lane and its last panel is narrow.  Singular values against
    LAPACK, orthonormal factors, reconstruction; rank-deficient variant with zero and repeated columns."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    m, n = shape
    k = min(m, n)
    full = rng.standard_normal((m, n))
    low = rng.standard_normal((m, 5)) @ rng.standard_normal((5, n))
    low[:, 1] = 0.0
    low[:, n - 1] = low[:, 0]
    for a in (full, low):
        u, s, vt = t4a.svd_backend(a)
        sref = np.linalg.svd(a, compute_uv=False)
        assert np.abs(s - sref).max() <= 1e-12 * sref[0]
        assert np.abs((u * s) @ vt - a).max() <= 1e-12 * sref[0] * k
        assert np.abs(u.T @ u - np.eye(k)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-10
        q, r = t4a.qr_backend(a)
        assert np.abs(q @ r - a).max() <= 1e-12 * sref[0] * k
        assert np.abs(q.T @ q - np.eye(k)).max() < 1e-11 and np.abs(np.tril(r, -1)).max() == 0.0


@pytest.mark.parametrize("shape", [(1000, 40), (1025, 33), (2000, 24), (3001, 17), (24, 2000), (9000, 20)])
def test_svd_and_qr_of_tall_and_skinny_matrices(t4a, shape):
    """Fewer than 64 columns: no QR preconditioner, the blocked Jacobi works on the long columns themselves — block width 8 with the 1 024-row
    register window (1 000 rows), narrower blocks with a tail beyond the window (1 025, 2 000, 3 001 rows: block width 8 / 4 / 2), columns too long
    for the LDS (9 000 rows: one launch per tournament round); the QR panels of more than 560 rows are factorised in global memory."""
    rng = np.random.default_rng(shape[0] + shape[1])
    a = rng.standard_normal(shape)
    k = min(shape)
    u, s, vt = t4a.svd_backend(a)
    sref = np.linalg.svd(a, compute_uv=False)
    assert np.abs(s - sref).max() <= 1e-12 * sref[0]
    assert np.abs((u * s) @ vt - a).max() <= 1e-12 * sref[0] * k
    assert np.abs(u.T @ u - np.eye(k)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-10
    q, r = t4a.qr_backend(a)
    assert np.abs(q @ r - a).max() <= 1e-12 * sref[0] * k
    assert np.abs(q.T @ q - np.eye(k)).max() < 1e-11 and np.abs(np.tril(r, -1)).max() == 0.0


def test_svd_and_qr_are_deterministic(t4a):
    """The same matrix twice: bitwise the same factors (the blocked Jacobi's tournament is fixed, its sweeps run in batches behind a
    device-side flag, the QR panel's tasks are assigned statically — no result may depend on which workgroup finishes first)."""
    rng = np.random.default_rng(77)
    for shape in [(300, 200), (64, 64), (40, 130), (1025, 33)]:
        a = rng.standard_normal(shape)
        u1, s1, v1 = t4a.svd_backend(a)
        u2, s2, v2 = t4a.svd_backend(a)
        assert np.array_equal(u1, u2) and np.array_equal(s1, s2) and np.array_equal(v1, v2)
        q1, r1 = t4a.qr_backend(a)
        q2, r2 = t4a.qr_backend(a)
        assert np.array_equal(q1, q2) and np.array_equal(r1, r2)


@pytest.mark.parametrize("shape", [(2, 2), (32, 32), (33, 33), (64, 32), (65, 32), (128, 31), (129, 17), (224, 32), (225, 32),   # V columns of 32
                                   (64, 64), (64, 33), (65, 64), (128, 64), (129, 63), (224, 64), (225, 64), (48, 200),              # V columns of 64
                                   (65, 65), (96, 96), (96, 65), (97, 96), (97, 65), (90, 81), (81, 90)])                            # 96 / beyond
def test_svd_one_launch_route_instantiations_and_their_edges(t4a, shape):
    """Round 6: matrices of up to 96 columns whose W and V fit one workgroup's LDS run the whole Jacobi iteration in one launch, sixteen
    lanes per column pair (jacobi_groups_kernel, eight instantiations by column length: V of 32 / 64 / 96 rows, W of 32 / 64 / 128 / 224
    / 96 rows), without the QR preconditioner.  Every instantiation, the shapes on both sides of every limit (225 rows, 97 rows next to
    more than 64 columns: the blocked tournament takes over), both orientations; a random matrix, a spectrum graded over twelve decades,
    exact rank deficiency (zero column, zero row, repeated column), the zero matrix, a non-finite entry; bitwise repeatability.
    svd_backend, tensor4all-tensorbackend/src/backend.rs:709-731; reconstruction tolerance of the reference's tests 1e-10
    (backend/tests/mod.rs:58-110)."""
    rng = np.random.default_rng(shape[0] * 977 + shape[1])
    m, n = shape
    k = min(m, n)
    q1, _ = np.linalg.qr(rng.standard_normal((m, k)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, k)))
    graded = (q1 * np.logspace(0, -12, k)) @ q2.T
    low = rng.standard_normal((m, max(1, k // 3))) @ rng.standard_normal((max(1, k // 3), n))
    low[:, 0] = 0.0
    low[m - 1, :] = 0.0
    if n > 2:
        low[:, n - 1] = low[:, 1]
    for a in (rng.standard_normal((m, n)), graded, low):
        u, s, vt = t4a.svd_backend(a)
        sref = np.linalg.svd(a, compute_uv=False)
        assert u.shape == (m, k) and s.shape == (k,) and vt.shape == (k, n)
        assert np.abs(s - sref).max() <= 1e-12 * sref[0]
        assert np.all(s >= 0.0) and np.all(np.diff(s) <= 1e-13 * s[0])
        assert np.abs((u * s) @ vt - a).max() <= 1e-12 * sref[0] * k
        assert np.abs(u.T @ u - np.eye(k)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-10
        u2, s2, vt2 = t4a.svd_backend(a)
        assert np.array_equal(u, u2) and np.array_equal(s, s2) and np.array_equal(vt, vt2)
    u, s, vt = t4a.svd_backend(np.zeros((m, n)))
    assert np.all(s == 0.0)
    assert np.abs(u.T @ u - np.eye(k)).max() < 1e-12 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-12
    for bad_value in (np.nan, np.inf, -np.inf):
        bad = rng.standard_normal((m, n))
        bad[m - 1, n // 2] = bad_value
        with pytest.raises(t4a.T4aError) as e:
            t4a.svd_backend(bad)
        assert e.value.code == t4a.INVALID_ARGUMENT
    # (the handle is usable after the error)
    u, s, vt = t4a.svd_backend(np.eye(m, n))
    assert np.abs(s - 1.0).max() < 1e-15


@pytest.mark.parametrize("scale", [1e120, 1e-120, 1e250, 1e-250, 2.0 ** 201, 2.0 ** -201, 2.0 ** 199])
def test_svd_of_matrices_far_from_unit_scale(t4a, scale):
    """Round 6 (found by tools/soak_svd_small.py): the pair test of the Jacobi rotations forms alpha * beta, the product of two squared
    column norms — from |a| ~ 1e77 on it overflowed, no pair rotated and U / V came back non-orthogonal WITHOUT an error; below 1e-77 the
    product underflowed.  A matrix whose largest entry is outside 2^-200 .. 2^200 is now scaled by a power of two (exact) and the singular
    values are scaled back (Engine::svd for the blocked and QR-preconditioned routes, inside jacobi_groups_kernel for the one-launch
    route).  Every route, both orientations, both sides of the threshold; singular values against LAPACK relative to the largest."""
    rng = np.random.default_rng(5)
    for (m, n) in [(40, 20), (20, 40), (64, 64), (96, 96), (1000, 40), (40, 1000), (200, 100), (100, 200), (300, 70)]:
        a = rng.standard_normal((m, n)) * scale
        k = min(m, n)
        u, s, vt = t4a.svd_backend(a)
        sref = np.linalg.svd(a, compute_uv=False)
        assert np.all(np.isfinite(s)) and np.all(np.isfinite(u)) and np.all(np.isfinite(vt))
        assert np.abs(s / sref[0] - sref / sref[0]).max() <= 1e-12, (m, n)
        assert np.abs(u.T @ u - np.eye(k)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-10, (m, n)
        assert np.abs(((u * (s / sref[0])) @ vt) - a / sref[0]).max() <= 1e-12 * k, (m, n)
    # one huge entry in an otherwise ordinary matrix, and a zero matrix next to the scaling logic
    a = rng.standard_normal((50, 30))
    a[7, 3] = scale
    u, s, vt = t4a.svd_backend(a)
    sref = np.linalg.svd(a, compute_uv=False)
    assert np.abs(s / sref[0] - sref / sref[0]).max() <= 1e-12
    assert np.abs(u.T @ u - np.eye(30)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(30)).max() < 1e-10


@pytest.mark.parametrize("scale", [1e-200, 1e-250, 1e200, 1e250, 2.0 ** -201, 2.0 ** 201, 2.0 ** 199, 1.0])
def test_qr_of_matrices_far_from_unit_scale(t4a, scale):
    """Round 6 (found by tools/soak_dense_small.py): the Householder column norms are sums of squares — with entries of 1e-200 every
    column was "zero" and qr_backend returned R = 0 (Q R = 0, no error); from ~1e154 on the squares overflow.  Engine::qr now factorises
    2^-e A for a matrix whose largest entry is outside 2^-200 .. 2^200 (exact, found and applied on the device) and puts the factor back on
    R.  Panels factorised in the LDS and in global memory (more than 560 rows), wide and tall; Q R = A relative to the largest entry, Q
    orthonormal, R upper triangular; the R of a scaled matrix is the scaled R, bit for bit, when the factor is a power of two."""
    rng = np.random.default_rng(9)
    for (m, n) in [(40, 20), (20, 40), (64, 64), (300, 120), (120, 300), (700, 60), (1, 5), (5, 1)]:
        a0 = rng.standard_normal((m, n))
        a = a0 * scale
        k = min(m, n)
        q, r = t4a.qr_backend(a)
        assert np.all(np.isfinite(q)) and np.all(np.isfinite(r)), (m, n)
        amax = np.abs(a).max()
        assert np.abs(q @ (r / amax) - a / amax).max() <= 1e-13 * k, (m, n)
        assert np.abs(q.T @ q - np.eye(k)).max() < 1e-11 and np.abs(np.tril(r, -1)).max() == 0.0, (m, n)
        if scale in (2.0 ** -201, 2.0 ** 201, 2.0 ** 199):
            q0, r0 = t4a.qr_backend(a0)
            assert np.array_equal(q, q0) and np.array_equal(r, r0 * scale), (m, n)


@pytest.mark.parametrize("shape", [(200, 100), (100, 200), (128, 128)])
def test_preconditioned_svd_on_rank_deficient_and_graded_inputs(t4a, shape):
    """Engine::svd runs the Jacobi iteration on L = R^T of a Householder QR from 64 columns on (round 5): exact rank deficiency (zero
    columns / rows, repeated columns), a graded spectrum over twelve decades, both orientations — singular values against LAPACK,
    orthonormal factors, reconstruction."""
    rng = np.random.default_rng(31)
    m, n = shape
    k = min(m, n)
    low = rng.standard_normal((m, 30)) @ rng.standard_normal((30, n))
    low[:, 3] = 0.0
    low[5, :] = 0.0
    low[:, 7] = low[:, 8]
    q1, _ = np.linalg.qr(rng.standard_normal((m, k)))
    q2, _ = np.linalg.qr(rng.standard_normal((n, k)))
    graded = (q1 * np.logspace(0, -12, k)) @ q2.T
    for a in (low, graded):
        u, s, vt = t4a.svd_backend(a)
        sref = np.linalg.svd(a, compute_uv=False)
        assert u.shape == (m, k) and vt.shape == (k, n)
        assert np.abs(s - sref).max() <= 1e-12 * sref[0]
        assert np.all(np.diff(s) <= 1e-13 * s[0])
        assert np.abs((u * s) @ vt - a).max() <= 1e-12 * sref[0] * k
        assert np.abs(u.T @ u - np.eye(k)).max() < 1e-10 and np.abs(vt @ vt.T - np.eye(k)).max() < 1e-10
