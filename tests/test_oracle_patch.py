"""Pins the oracle's adaptive patching driver (oracle/t4a_oracle_patch.hpp) against the reference's own tests
(crates/tensor4all-partitionedtt/src/adaptive_interpolation/tests.rs).  CPU only."""
import numpy as np
import pytest

import oracle_binding as ob
from t4a_amd import TCI2Options


def disjoint(projs):
    for i, a in enumerate(projs):
        for b in projs[i + 1:]:
            if all(a[k] == b[k] for k in a.keys() & b.keys()):
                return False
    return True


def test_low_rank_function_without_splitting():
    f = lambda i: (i[0] + 1.0) * (i[1] + 2.0) * (i[2] + 3.0)
    r = ob.adaptiveinterpolate(f, [2, 2, 2], [[1, 1, 1]], TCI2Options())
    assert len(r) == 1 and r.projector(0) == {}
    assert np.allclose(r.dense([2, 2, 2]), [6.0, 12.0, 9.0, 18.0, 8.0, 16.0, 12.0, 24.0], rtol=1e-12, atol=0)


def test_single_active_site_is_exact():
    f = lambda i: 10.0 if i[0] == 3 else float(i[0] * i[0] + 1)
    r = ob.adaptiveinterpolate(f, [4], [], TCI2Options(), n_initial_pivots=1)
    assert len(r) == 1
    assert list(r.dense([4])) == [1.0, 2.0, 5.0, 10.0]


def test_rank_cap_forces_disjoint_exact_child_patches():
    f = lambda i: 2.0 if all(v == i[0] for v in i) else 0.5
    opt = TCI2Options(tolerance=1e-14, max_bond_dim=1, max_iter=4, ncheck_history=1, nsearch=0, max_nglobal_pivot=0)
    r = ob.adaptiveinterpolate(f, [2, 2, 2], [[0, 0, 0], [1, 1, 1]], opt, patch_order=[0, 1, 2], recycle_pivots=True)
    assert len(r) >= 2
    assert disjoint([r.projector(k) for k in range(len(r))])
    assert np.allclose(r.dense([2, 2, 2]), [2.0, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 2.0], rtol=1e-12, atol=0)
    # projected middle / edge sites carry delta tensors with the carried bond dimension
    for k in range(len(r)):
        cores = r.cores(k)
        for pos, val in r.projector(k).items():
            c = cores[pos]
            assert c.shape[0] == c.shape[2]
            assert np.array_equal(c[:, val, :], np.eye(c.shape[0])) and np.abs(c).sum() == c.shape[0]


def test_batched_callback_is_used_and_zero_function_is_zero():
    calls = [0]

    class F:
        def __call__(self, i):
            return float(i[0] + i[1] + 1)

        def batch(self, pts):
            calls[0] += 1
            return [float(p[0] + p[1] + 1) for p in pts]

    fn = F()
    fn.batched = fn.batch
    r = ob.adaptiveinterpolate(fn, [2, 2], [[1, 1]], TCI2Options())
    assert list(r.dense([2, 2])) == pytest.approx([1.0, 2.0, 2.0, 3.0], abs=1e-12)
    z = ob.adaptiveinterpolate(lambda i: 0.0, [2, 2], [], TCI2Options())
    assert len(z) == 1 and list(z.dense([2, 2])) == [0.0] * 4


def test_numerically_zero_child_patches_are_accepted():
    """issue #598 regression (tests.rs:434-491): fused-quantics Gaussian mixture on sites of dimension 4 (R = 7 and a
    rank cap of 12 instead of R = 10 / 64 so that the scalar Python callback keeps the CPU suite short)."""
    W, A, C, L, R = [1.3, 0.9, 0.9], [2.8, 5.4, 0.7], [(0.4, 0.1), (3.8, -0.8), (-5.5, -2.1)], 12.0, 7

    def batch(pts):
        pts = np.asarray(pts, dtype=np.int64).reshape(-1, R)
        sh = (R - 1 - np.arange(R))[None, :]
        ix = ((pts & 1) << sh).sum(axis=1)
        iy = (((pts >> 1) & 1) << sh).sum(axis=1)
        step = 2.0 * L / (1 << R)
        x, y = -L + ix * step, -L + iy * step
        return sum(W[i] * np.exp(-A[i] * ((x - C[i][0]) ** 2 + (y - C[i][1]) ** 2)) for i in range(3))

    class Mixture:
        def __call__(self, index):
            return float(batch([list(index)])[0])

        def batched(self, pts):
            return batch(pts)

    f = Mixture()
    opt = TCI2Options(tolerance=1e-8, max_bond_dim=12, max_iter=20, normalize_error=False, seed=1)
    r = ob.adaptiveinterpolate(f, [4] * R, [], opt)
    assert len(r) > 1 and disjoint([r.projector(k) for k in range(len(r))])
    # the reference only requires a valid, complete partition here (near-zero children must not crash): the patches
    # tile the whole index space exactly once
    vol = sum(4 ** (R - len(r.projector(k))) for k in range(len(r)))
    assert vol == 4 ** R
    pts = np.random.default_rng(0).integers(0, 4, size=(300, R))
    assert np.all(np.isfinite(r.evaluate(pts)))


def test_invalid_inputs_are_rejected():
    f = lambda i: 1.0
    with pytest.raises(ob.OracleError):
        ob.adaptiveinterpolate(f, [2, 2], [], TCI2Options(), n_initial_pivots=0)
    with pytest.raises(ob.OracleError):
        ob.adaptiveinterpolate(f, [2, 2], [], TCI2Options(), patch_order=[0, 0])
    with pytest.raises(ob.OracleError):
        ob.adaptiveinterpolate(f, [2, 2], [[0, 2]], TCI2Options())
