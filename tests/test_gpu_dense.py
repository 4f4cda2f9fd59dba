"""GPU parity tests of the dense kernels under the TCI2 sweep, through the C ABI (include/t4a_gpu.h).

rrLU: bit-exact against the CPU oracle (pivot order, factored matrix, errors) — the oracle itself is pinned to
the reference's Hilbert table and known-answer tests in test_oracle_golden.py.
solve / trsm / gemm / LUCI factor VALUES: tolerance level (the reference delegates them to tenferro;
SURVEY.md §8c "parity unpinned"), tolerances written next to each assert.
"""
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


def hilbert(n):
    i = np.arange(n)
    return 1.0 / (i[:, None] + i[None, :] + 1.0)


def assert_rrlu_bit_exact(t4a, a, **kw):
    ref_f, ref_rp, ref_cp, ref_np, ref_err = ob.rrlu(a, **kw)
    lu = t4a.rrlu(a, **kw)
    assert lu.npivots() == ref_np
    assert np.array_equal(lu.row_permutation, ref_rp)
    assert np.array_equal(lu.col_permutation, ref_cp)
    # bitwise equality of the whole factored buffer (NaN-safe)
    assert np.array_equal(lu.factored.view(np.uint64), ref_f.view(np.uint64))
    if np.isnan(ref_err):
        assert np.isnan(lu.error)
    else:
        assert lu.error == ref_err
    return lu


@pytest.mark.parametrize("n,rank,err", [(16, 10, 2.198484e-12), (32, 11, 4.197675e-11), (64, 13, 9.601802e-12),
                                        (128, 14, 3.690140e-11)])
@pytest.mark.parametrize("left", [True, False])
def test_hilbert_table(t4a, n, rank, err, left):
    # reference: benchmarks/results/2026-05-22-matrix-lu-hilbert.md:44-51
    lu = assert_rrlu_bit_exact(t4a, hilbert(n), rel_tol=0.0, abs_tol=1e-10, left_orthogonal=left)
    assert lu.npivots() == rank
    assert float("%.6e" % lu.last_pivot_error()) == err


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (3, 7), (7, 3), (17, 33), (64, 64), (96, 100), (130, 70),
                                   (200, 200), (257, 255), (512, 512), (768, 768), (640, 300)])
@pytest.mark.parametrize("left", [True, False])
def test_random_full_rank_bit_exact(t4a, shape, left):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    a = rng.uniform(-1, 1, size=shape)
    assert_rrlu_bit_exact(t4a, a, left_orthogonal=left)


@pytest.mark.parametrize("shape,r", [((300, 280), 40), ((512, 512), 256), ((768, 700), 256)])
def test_low_rank_and_truncation(t4a, shape, r):
    rng = np.random.default_rng(7)
    a = rng.normal(size=(shape[0], r)) @ rng.normal(size=(r, shape[1]))
    assert_rrlu_bit_exact(t4a, a, rel_tol=1e-12)
    assert_rrlu_bit_exact(t4a, a, max_bond_dim=r // 2, rel_tol=1e-12, left_orthogonal=False)
    assert_rrlu_bit_exact(t4a, a, rel_tol=0.0, abs_tol=1e-3)


@pytest.mark.parametrize("shape,maxb", [((1024, 1024), 300), ((1370, 1376), 200), ((1536, 1536), 128),
                                        ((2048, 1000), 150), ((3000, 500), 100), ((500, 3000), 100)])
@pytest.mark.parametrize("left", [True, False])
def test_cfg4_sized_matrices(t4a, shape, maxb, left):
    """BASELINE config 4 (chi = 512) candidate matrices are 1024..1536 on a side: register-resident kernel with
    3 rows x 8 columns per thread up to M = 1536, LDS-resident fallback beyond."""
    rng = np.random.default_rng(shape[0] + maxb)
    a = rng.uniform(-1, 1, size=shape)
    assert_rrlu_bit_exact(t4a, a, max_bond_dim=maxb, left_orthogonal=left)


def test_ties_follow_permuted_positions(t4a):
    # many exactly equal |entries|: the winner must be the first in column-major order of the permuted block
    rng = np.random.default_rng(3)
    a = rng.integers(-2, 3, size=(150, 140)).astype(float)
    assert_rrlu_bit_exact(t4a, a, rel_tol=0.0, abs_tol=0.0)
    a = np.ones((130, 129))
    assert_rrlu_bit_exact(t4a, a)


def test_edge_cases(t4a):
    z = np.zeros((3, 3))
    lu = assert_rrlu_bit_exact(t4a, z)
    assert lu.npivots() == 0  # matrixlu/tests/mod.rs:140-151
    eye = np.eye(2)
    lu = assert_rrlu_bit_exact(t4a, eye)
    assert np.allclose(lu.pivot_errors(), [1.0, 1.0, 0.0], atol=1e-14)  # :283-295
    tiny = np.array([[1e-20, 1.0, 0.0], [1.0, 1e-20, 0.0], [0.0, 0.0, 1e-20]])
    assert_rrlu_bit_exact(t4a, tiny)  # NaN regression :124-138
    # max_bond_dim = 1 on a 2x2 (doc example matrixlu.rs:728-733)
    lu = assert_rrlu_bit_exact(t4a, np.array([[1.0, 2.0], [3.0, 4.0]]), max_bond_dim=1)
    assert lu.npivots() == 1


def test_nan_is_reported(t4a):
    a = np.eye(4)
    a[0, 0] = np.nan
    with pytest.raises(ob.OracleError):
        ob.rrlu(a)
    with pytest.raises(t4a.T4aError) as e:
        t4a.rrlu(a)
    assert e.value.code == t4a.NAN_ENCOUNTERED


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,maxb", [((8, 6), 4), ((40, 50), None), ((300, 200), 64), ((512, 512), 256)])
def test_luci_factors(t4a, left, shape, maxb):
    rng = np.random.default_rng(11)
    a = rng.uniform(-1, 1, size=shape)
    ref = ob.luci(a, max_bond_dim=maxb, left_orthogonal=left)
    f = t4a.matrix_luci_factors_from_matrix(a, max_bond_dim=maxb, left_orthogonal=left)
    assert f.rank == ref["rank"]
    assert np.array_equal(f.row_indices, ref["rows"])
    assert np.array_equal(f.col_indices, ref["cols"])
    assert np.array_equal(f.pivot_errors, ref["pivot_errors"])
    # factor values: tolerance-level (trsm/gemm belong to tenferro in the reference). 1e-10 relative to the
    # factor magnitude; the factors of a random matrix are O(1)..O(10).
    scale_l = max(1.0, np.abs(ref["left"]).max())
    scale_r = max(1.0, np.abs(ref["right"]).max())
    assert np.abs(f.left - ref["left"]).max() <= 1e-10 * scale_l
    assert np.abs(f.right - ref["right"]).max() <= 1e-10 * scale_r
    # interpolation property: left * right reproduces the pivot rows/columns (matrix_luci/tests/mod.rs:65-133)
    rec = f.left @ f.right
    assert np.abs(rec[f.row_indices, :] - a[f.row_indices, :]).max() < 1e-9 * scale_l * scale_r


@pytest.mark.parametrize("m,k,n", [(2, 3, 2), (64, 64, 64), (100, 37, 51), (256, 256, 512), (513, 129, 65)])
def test_gemm(t4a, m, k, n):
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, size=(m, k))
    b = rng.uniform(-1, 1, size=(k, n))
    c = t4a.mat_mul(a, b)
    ref = ob.gemm(a, b)
    assert np.abs(c - ref).max() <= 1e-12 * k  # fp64 accumulation-order tolerance
    # exact small integers (matrix/tests/mod.rs:488-553)
    ai = rng.integers(-3, 4, size=(m, k)).astype(float)
    bi = rng.integers(-3, 4, size=(k, n)).astype(float)
    assert np.array_equal(t4a.mat_mul(ai, bi), ai @ bi)


@pytest.mark.parametrize("m,k,n", [(1536, 40, 1536),   # 24 x 24 tiles of 64 x 64: wide-tile kernel, XCD-aware tile order
                                   (1000, 72, 1100),   # 64 x 32 tiles (16 x 35 of them), ragged edges, remapped tile order
                                   (1030, 33, 1100),   # 17 x 35 tiles: not a multiple of eight -> plain tile order, partial k-tile
                                   (64, 4096, 32)])    # one tile, long k (pointer-stepped interior path)
def test_gemm_tile_variants(t4a, m, k, n):
    """The two tile widths of gemm_kernel, its interior fast path and both tile orders against exact integer products (every
    partial sum stays far below 2^53, so any summation order gives the same bits)."""
    rng = np.random.default_rng(m + 7 * k + 13 * n)
    ai = rng.integers(-3, 4, size=(m, k)).astype(float)
    bi = rng.integers(-3, 4, size=(k, n)).astype(float)
    assert np.array_equal(t4a.mat_mul(ai, bi), ai @ bi)


def test_gemm_batched(t4a):
    rng = np.random.default_rng(6)
    batch, m, k, n = 5, 7, 9, 4
    a = rng.integers(-3, 4, size=(batch, k, m)).astype(float)  # memory: [m,k,batch] column-major
    b = rng.integers(-3, 4, size=(batch, n, k)).astype(float)
    c = t4a.batched_mat_mul_same_shape(batch, m, k, n, a.ravel(), b.ravel()).reshape(batch, n, m)
    for q in range(batch):
        assert np.array_equal(c[q].T, a[q].T @ b[q].T)


@pytest.mark.parametrize("left_side", [True, False])
@pytest.mark.parametrize("lower", [True, False])
@pytest.mark.parametrize("trans", [True, False])
@pytest.mark.parametrize("unit", [True, False])
def test_trsm(t4a, left_side, lower, trans, unit):
    rng = np.random.default_rng(9)
    n, other = 70, 45
    a = rng.uniform(-1, 1, size=(n, n)) + 4 * np.eye(n)
    b = rng.uniform(-1, 1, size=(n, other) if left_side else (other, n))
    x = t4a.triangular_solve_matrix(a, b, left_side, lower, trans, unit)
    ref = ob.trsm(a, b, left_side, lower, trans, unit)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    # closed-form 2x2 (backend/tests/mod.rs:119-325): L = [[2,0],[1,4]], L x = b
    l = np.array([[2.0, 0.0], [1.0, 4.0]])
    bb = np.array([[2.0], [9.0]])
    assert np.abs(t4a.triangular_solve_matrix(l, bb, True, True, False, False) - np.array([[1.0], [2.0]])).max() < 1e-12


@pytest.mark.parametrize("n,nrhs", [(2, 1), (10, 3), (100, 40), (256, 512), (32, 16), (33, 17), (64, 64), (200, 100), (300, 70), (500, 33), (512, 1024), (513, 64)])
def test_solve(t4a, n, nrhs):
    rng = np.random.default_rng(13)
    a = rng.uniform(-1, 1, size=(n, n))
    b = rng.uniform(-1, 1, size=(n, nrhs))
    x = t4a.solve_matrix(a, b)
    ref = ob.solve(a, b)
    assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(a @ x - b).max() <= 1e-9 * max(1.0, np.abs(x).max()) * n
    # closed form (backend/tests/mod.rs): [[2,1],[1,3]] x = [3,5] -> [0.8, 1.4]
    xs = t4a.solve_matrix(np.array([[2.0, 1.0], [1.0, 3.0]]), np.array([[3.0], [5.0]]))
    assert np.abs(xs.ravel() - np.array([0.8, 1.4])).max() < 1e-12


def test_solve_singular(t4a):
    with pytest.raises(t4a.T4aError) as e:
        t4a.solve_matrix(np.zeros((3, 3)), np.ones((3, 1)))
    assert e.value.code == t4a.SINGULAR_MATRIX


def test_function_workload_bit_exact(t4a):
    from t4a_amd.functions import quantics_trig_exp, quantics_osc2d, lorentz
    rng = np.random.default_rng(0)
    for spec in (quantics_trig_exp(20), quantics_osc2d(30, k4=97, delta=0.3), lorentz([10] * 5)):
        idx = np.stack([rng.integers(0, d, size=4000) for d in spec.local_dims], axis=1)
        gpu = t4a.fn_eval(spec, spec.local_dims, idx)
        cpu = ob.fn_eval(spec, idx)
        assert np.array_equal(gpu.view(np.uint64), cpu.view(np.uint64))


def test_single_xcd_kernel_falls_back_when_its_placement_assumption_fails():
    """The single-XCD rrLU kernel elects its workgroups by HW_REG_XCC_ID.  With an XCC id that no workgroup reports
    (T4A_XCD_ID=8) nobody takes part: the launch ends without the completion token, the engine re-runs the factorisation with
    the chip-wide kernel and keeps the single-XCD path off — results are bitwise those of the default run and of T4A_NO_XCD=1."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(%r, "tensor4all-rs_amd", "python"))
import numpy as np, t4a_amd
rng = np.random.default_rng(3)
h = hashlib.sha256()
for shape in ((300, 260), (130, 512), (700, 690)):
    a = rng.uniform(-1, 1, size=shape)
    for left in (True, False):
        lu = t4a_amd.rrlu(a, max_bond_dim=96, left_orthogonal=left)
        h.update(np.ascontiguousarray(lu.factored).tobytes()); h.update(lu.row_permutation.tobytes()); h.update(lu.col_permutation.tobytes())
        h.update(np.float64(lu.error).tobytes())
print("DIGEST", h.hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for name, env in (("default", {}), ("no_xcd", {"T4A_NO_XCD": "1"}), ("bad_xcc", {"T4A_XCD_ID": "8"})):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[name] = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]
    assert digests["default"] == digests["no_xcd"] == digests["bad_xcc"], digests
