"""Randomised bitwise parity of the rrLU entry point against the oracle: mixed shapes (single- and multi-workgroup
plans back to back, which exercises the alternating key tables and the in-kernel clean-up), rank-deficient inputs,
ties, all stop rules, both orthogonalities."""
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


def random_case(rng):
    kind = rng.integers(0, 5)
    m = int(rng.integers(1, 420)) if kind != 4 else int(rng.integers(300, 900))
    n = int(rng.integers(1, 420)) if kind != 4 else int(rng.integers(300, 900))
    if kind == 0:
        a = rng.uniform(-1, 1, size=(m, n))
    elif kind == 1:  # low rank
        r = int(rng.integers(1, max(2, min(m, n))))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    elif kind == 2:  # many exact ties
        a = rng.integers(-2, 3, size=(m, n)).astype(float)
    elif kind == 3:  # wide dynamic range
        a = rng.standard_normal((m, n)) * 10.0 ** rng.integers(-12, 3, size=(m, 1))
    else:
        a = rng.uniform(-1, 1, size=(m, n))
    opts = {}
    if rng.random() < 0.5:
        opts["max_bond_dim"] = int(rng.integers(1, min(m, n) + 1))
    mode = rng.integers(0, 4)
    if mode == 0:
        opts.update(rel_tol=0.0, abs_tol=0.0)
    elif mode == 1:
        opts.update(rel_tol=float(10.0 ** rng.integers(-14, -2)), abs_tol=0.0)
    elif mode == 2:
        opts.update(rel_tol=0.0, abs_tol=float(10.0 ** rng.integers(-10, 0)))
    opts["left_orthogonal"] = bool(rng.integers(0, 2))
    if kind == 4:
        opts["max_bond_dim"] = int(rng.integers(8, 120))  # keeps the oracle fast on the large shapes
    return a, opts


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_rrlu_random_cases_bitwise(t4a, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(60):
        a, opts = random_case(rng)
        lu = t4a.rrlu(a, **opts)
        f, rp, cp, npiv, err = ob.rrlu(a, **opts)
        ctx = f"seed {seed} case {case} shape {a.shape} opts {opts}"
        assert lu.npivots() == npiv, ctx
        assert np.array_equal(lu.row_permutation, rp) and np.array_equal(lu.col_permutation, cp), ctx
        assert np.array_equal(lu.factored, f), ctx
        assert lu.error == err or (np.isnan(lu.error) and np.isnan(err)), ctx


def _same_outcome(t4a, a, **opts):
    """Device and oracle either both refuse (NaN in L / U: MatrixCIError::NaNEncountered) or agree bitwise."""
    try:
        f, rp, cp, npiv, err = ob.rrlu(a, **opts)
    except ob.OracleError:
        with pytest.raises(t4a.T4aError) as e:
            t4a.rrlu(a, **opts)
        assert e.value.code == t4a.NAN_ENCOUNTERED
        return None
    lu = t4a.rrlu(a, **opts)
    assert lu.npivots() == npiv
    assert np.array_equal(lu.row_permutation, rp) and np.array_equal(lu.col_permutation, cp)
    assert np.array_equal(lu.factored.view(np.uint64), f.view(np.uint64))
    assert lu.error == err or (np.isnan(lu.error) and np.isnan(err))
    return lu


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape", [(1024, 1024), (1000, 900), (1024, 700), (800, 1000), (760, 1024)])
def test_rrlu_on_the_widest_single_xcd_plans(t4a, left, shape):
    """Up to 64 matrix entries per lane since the end of round 4 (12 x 4, 16 x 3, 16 x 4 rows x columns per lane: 1024 x 1024 in
    the registers of one XCD): random, tie-ridden and low-rank matrices at those shapes, rank capped so that the oracle stays
    fast, against the oracle bitwise."""
    m, n = shape
    rng = np.random.default_rng(9000 + m + n)
    kw = dict(left_orthogonal=left)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), max_bond_dim=48, rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.integers(-2, 3, size=(m, n)).astype(float), max_bond_dim=24, rel_tol=0.0, abs_tol=0.0, **kw)
    r = 20
    _same_outcome(t4a, rng.standard_normal((m, r)) @ rng.standard_normal((r, n)), max_bond_dim=40, **kw)
    bad = rng.uniform(-1, 1, size=(m, n))
    bad[m // 2, n // 2] = np.inf
    _same_outcome(t4a, bad, max_bond_dim=5, **kw)  # (handed to the first generation at the same plan)


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape", [(1464, 1448), (1024, 1428), (1428, 1024), (1424, 512), (1100, 900), (1536, 1536), (1300, 200), (600, 1500)])
def test_rrlu_beyond_one_xcd(t4a, left, shape):
    """Round 5: matrices beyond one XCD's 1024 x 1024 (kernels_rrlu_xcd2m.hip) — 24 row slots per lane (up to 1 536 rows) and the
    columns over the agents of two or three XCDs with one polling wave per XCD; BASELINE.json configs[3]'s saturated bonds with the
    history extras are ~1 450 x 1 450.  Random, tie-ridden (every step through the exact walk over all full keys), low-rank, zero and
    wide-range matrices at those shapes against the oracle bitwise, rank capped so that the oracle stays fast; non-finite inputs go
    back to the chip-wide kernel."""
    m, n = shape
    rng = np.random.default_rng(9100 + m + n)
    kw = dict(left_orthogonal=left)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), max_bond_dim=40, rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.integers(-2, 3, size=(m, n)).astype(float), max_bond_dim=16, rel_tol=0.0, abs_tol=0.0, **kw)
    r = 12
    _same_outcome(t4a, rng.standard_normal((m, r)) @ rng.standard_normal((r, n)), max_bond_dim=30, **kw)
    _same_outcome(t4a, np.zeros((m, n)), **kw)
    wide = rng.standard_normal((m, n)) * 10.0 ** rng.integers(-12, 3, size=(m, 1))
    _same_outcome(t4a, wide, max_bond_dim=24, rel_tol=1e-9, abs_tol=0.0, **kw)
    big = rng.uniform(-1, 1, size=(m, n))          # scores that overflow: the exact walk must see the infinity and hand over
    big[m // 2, n // 3] = 1e200
    big[m // 3, n // 2] = -3e199
    _same_outcome(t4a, big, max_bond_dim=6, rel_tol=0.0, abs_tol=0.0, **kw)
    bad = rng.uniform(-1, 1, size=(m, n))
    bad[m - 1, n - 1] = np.nan
    _same_outcome(t4a, bad, max_bond_dim=5, **kw)
    bad[m - 1, n - 1] = 0.5
    bad[0, 0] = np.nan                              # the NaN incumbent on the diagonal stays (matrixlu.rs:480-519)
    _same_outcome(t4a, bad, max_bond_dim=4, **kw)


@pytest.mark.parametrize("left", [True, False])
def test_rrlu_special_values_on_multi_workgroup_shapes(t4a, left):
    """Shapes that take the single-XCD kernel (more than 64 x 64 entries, up to 768 x 768) with the values that leave its
    fast paths: scores that overflow to +inf or underflow to 0 (ties between different |v|: exact sweep on the squares,
    matrixlu.rs:480-519), subnormal entries, exact zeros, a NaN incumbent on the diagonal, NaN / inf elsewhere, blocks of
    exact ties — and the pivot divisions that leave the shared-reciprocal fast path (zeros, huge ratios)."""
    rng = np.random.default_rng(77)
    kw = dict(left_orthogonal=left)
    # squares overflow: several distinct huge magnitudes all score +inf; the first in tie order must win
    a = rng.uniform(-1, 1, size=(90, 130))
    for (i, j, v) in [(7, 100, 1e200), (60, 3, -3e199), (61, 3, 2e180), (2, 2, 9e170)]:
        a[i, j] = v
    _same_outcome(t4a, a, max_bond_dim=12, rel_tol=0.0, abs_tol=0.0, **kw)
    # squares underflow to 0: every score ties at 0, the diagonal incumbent is kept at every step
    tiny = rng.uniform(0.5, 1, size=(100, 80)) * 1e-200
    _same_outcome(t4a, tiny, max_bond_dim=9, rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, tiny, **kw)                                     # default tolerances: stops on the first pivot
    # subnormal entries (the squares are 0, the quotients by the pivot are not)
    sub = rng.integers(1, 1000, size=(70, 140)).astype(float) * 5e-324 * 1e10
    _same_outcome(t4a, sub, max_bond_dim=6, rel_tol=0.0, abs_tol=0.0, **kw)
    # mixed scales in one matrix: quotients far outside 2^+-300 take the full division
    mixed = rng.uniform(-1, 1, size=(96, 96))
    mixed[:, ::7] *= 1e-180
    mixed[::5, :] *= 1e150
    _same_outcome(t4a, mixed, max_bond_dim=20, rel_tol=0.0, abs_tol=0.0, **kw)
    # exact zeros: an all-zero matrix, and a low-rank one whose trailing block becomes exactly zero
    _same_outcome(t4a, np.zeros((100, 100)), **kw)
    lowrank = np.outer(np.arange(1, 121), np.arange(1, 91)).astype(float)
    _same_outcome(t4a, lowrank, rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, lowrank, **kw)
    # blocks of exact ties (signed permutation-like matrix)
    perm = np.zeros((128, 128))
    perm[np.arange(128), rng.permutation(128)] = rng.choice([-1.0, 1.0], size=128)
    _same_outcome(t4a, perm, **kw)
    _same_outcome(t4a, np.kron(perm[:16, :16] + 0.0, np.ones((8, 8))), rel_tol=0.0, abs_tol=0.0, max_bond_dim=40, **kw)
    # NaN on the first diagonal element (the reference's initial incumbent stays), NaN elsewhere, inf
    nan0 = rng.uniform(-1, 1, size=(80, 120))
    nan0[0, 0] = np.nan
    _same_outcome(t4a, nan0, max_bond_dim=5, **kw)
    nan1 = rng.uniform(-1, 1, size=(80, 120))
    nan1[40, 77] = np.nan
    _same_outcome(t4a, nan1, max_bond_dim=5, **kw)
    inf1 = rng.uniform(-1, 1, size=(120, 80))
    inf1[11, 13] = np.inf
    _same_outcome(t4a, inf1, max_bond_dim=4, **kw)


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape", [(40, 100), (64, 128), (128, 60), (60, 30), (9, 120), (128, 8)])
def test_rrlu_special_values_on_one_workgroup_shapes(t4a, left, shape):
    """The shapes the one-workgroup kernel takes (kernels_rrlu_wg.hip: up to 64 x 128 / 128 x 64 as the kernel sees them, i.e.
    transposed for a right-orthogonal factorisation) with everything that leaves its fast paths: ties inside a wave and between
    waves, zero / subnormal / overflowing scores, low rank down to an exactly zero trailing block, all stop rules, quotients
    outside the shared-reciprocal range, and NaN / inf — which that kernel hands back to the first-generation kernel."""
    _special_value_suite(t4a, left, shape)


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape", [(64, 16), (33, 9), (12, 12), (50, 3), (16, 64), (3, 40), (9, 9)])
def test_rrlu_special_values_on_one_wave_shapes(t4a, left, shape):
    """The same suite on the shapes the ONE-WAVE kernel takes (kernels_rrlu_w1_body.hpp: at most 64 rows and 16 columns as the
    kernel sees them — a right-orthogonal factorisation of a 16 x 64 matrix is one): ties inside a lane, between lanes and between
    register groups, zero / subnormal maxima (the exact sweep with explicit position tests, cleared columns must never win),
    all stop rules, the factored matrix out of the LDS side buffers, NaN / inf handed back to the first-generation kernel."""
    _special_value_suite(t4a, left, shape)


def test_one_wave_kernel_on_wider_matrices_in_a_child_process():
    """T4A_W1_MAXN=64 lifts the plan limit so that the 32- and 64-column instantiations of the one-wave kernel (which the
    persistent half-sweep uses up to 32 columns) run the dense entry point; T4A_WG_MIN=0 sends even the tiniest matrices there."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "import t4a_amd, test_gpu_fuzz as f\n"
        "rng = np.random.default_rng(77)\n"
        "for (m, n) in [(64, 64), (40, 32), (64, 20), (7, 33), (2, 2), (1, 5), (5, 1)]:\n"
        "    for left in (True, False):\n"
        "        f._same_outcome(t4a_amd, rng.uniform(-1, 1, size=(m, n)), left_orthogonal=left)\n"
        "        f._same_outcome(t4a_amd, rng.integers(-2, 3, size=(m, n)).astype(float), rel_tol=0.0, abs_tol=0.0, left_orthogonal=left)\n"
        "        f._same_outcome(t4a_amd, np.outer(np.arange(1, m + 1), np.arange(1, n + 1)).astype(float), rel_tol=0.0, abs_tol=0.0, left_orthogonal=left)\n"
        "        f._same_outcome(t4a_amd, rng.uniform(-1, 1, size=(m, n)), max_bond_dim=max(1, min(m, n) // 2), rel_tol=1e-3, abs_tol=0.0, left_orthogonal=left)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, T4A_W1_MAXN="64", T4A_WG_MIN="0", PYTHONPATH=os.path.join(root, "tensor4all-rs_amd", "python"))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def _special_value_suite(t4a, left, shape):
    m, n = shape
    rng = np.random.default_rng(4000 + 10 * m + n)
    kw = dict(left_orthogonal=left)
    mn = min(m, n)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), **kw)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), max_bond_dim=max(1, mn // 3), rel_tol=1e-3, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.uniform(-1, 1, size=(m, n)), rel_tol=0.0, abs_tol=0.3, **kw)
    # exact ties everywhere: small integers, a signed permutation pattern, constant blocks
    _same_outcome(t4a, rng.integers(-2, 3, size=(m, n)).astype(float), rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.integers(-1, 2, size=(m, n)).astype(float), **kw)
    ones = np.ones((m, n))
    _same_outcome(t4a, ones, rel_tol=0.0, abs_tol=0.0, **kw)
    perm = np.zeros((m, n))
    perm[rng.permutation(m)[:mn], rng.permutation(n)[:mn]] = rng.choice([-1.0, 1.0], size=mn)
    _same_outcome(t4a, perm, rel_tol=0.0, abs_tol=0.0, **kw)
    # low rank: the trailing block becomes exactly zero (integers) / numerically tiny (floats)
    _same_outcome(t4a, np.outer(np.arange(1, m + 1), np.arange(1, n + 1)).astype(float), rel_tol=0.0, abs_tol=0.0, **kw)
    r = max(1, mn // 4)
    _same_outcome(t4a, rng.standard_normal((m, r)) @ rng.standard_normal((r, n)), **kw)
    _same_outcome(t4a, np.zeros((m, n)), **kw)
    # scores that underflow to 0 / overflow to +inf, subnormal entries, mixed scales (full division)
    _same_outcome(t4a, rng.uniform(0.5, 1, size=(m, n)) * 1e-200, max_bond_dim=min(5, mn), rel_tol=0.0, abs_tol=0.0, **kw)
    big = rng.uniform(-1, 1, size=(m, n))
    big[m // 2, n // 3] = 1e200
    big[m // 3, n // 2] = -3e199
    _same_outcome(t4a, big, max_bond_dim=min(6, mn), rel_tol=0.0, abs_tol=0.0, **kw)
    _same_outcome(t4a, rng.integers(1, 1000, size=(m, n)).astype(float) * 5e-324 * 1e10, max_bond_dim=min(4, mn), rel_tol=0.0, abs_tol=0.0, **kw)
    mixed = rng.uniform(-1, 1, size=(m, n))
    mixed[:, ::3] *= 1e-180
    mixed[::4, :] *= 1e150
    _same_outcome(t4a, mixed, max_bond_dim=min(12, mn), rel_tol=0.0, abs_tol=0.0, **kw)
    # non-finite input: the NaN incumbent on the diagonal stays, NaN / inf elsewhere never win (matrixlu.rs:480-519)
    for (i, j, v) in [(0, 0, np.nan), (m - 1, n - 1, np.nan), (m // 2, n // 2, np.inf), (0, n - 1, -np.inf)]:
        bad = rng.uniform(-1, 1, size=(m, n))
        bad[i, j] = v
        _same_outcome(t4a, bad, max_bond_dim=min(5, mn), **kw)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tci2_random_option_combinations_match_oracle(t4a, seed):
    """Random option combinations on built-in functions: index sets, error history and termination equal the oracle's."""
    from t4a_amd.functions import lorentz, quantics_osc2d, quantics_trig_exp
    rng = np.random.default_rng(500 + seed)
    for case in range(6):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            d = int(rng.integers(3, 6))
            dims = [int(rng.integers(2, 6)) for _ in range(d)]
            spec = lorentz(dims)
        elif kind == 1:
            d = 2 * int(rng.integers(3, 7))
            dims = [2] * d
            spec = quantics_osc2d(d, k1=int(rng.integers(1, 9)), k2=int(rng.integers(1, 9)), k3=int(rng.integers(1, 40)),
                                  eps=0.5, k4=int(rng.integers(1, 90)), delta=0.5)
        else:
            d = int(rng.integers(6, 14))
            dims = [2] * d
            spec = quantics_trig_exp(d)
        opt = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-11, -4)), max_iter=int(rng.integers(2, 7)),
                              max_bond_dim=(None if rng.random() < 0.3 else int(rng.integers(2, 20))),
                              normalize_error=bool(rng.integers(0, 2)), sweep_strategy=int(rng.integers(0, 3)),
                              strictly_nested=bool(rng.integers(0, 2)), ncheck_history=int(rng.integers(1, 4)),
                              nsearch=int(rng.integers(0, 4)), max_nglobal_pivot=int(rng.integers(0, 3)),
                              seed=int(rng.integers(0, 100)))
        g = t4a.TensorCI2(dims)
        g.set_function(spec)
        o = ob.OracleTCI2(dims)
        o.set_function(spec)
        piv = [[0] * d]
        g.crossinterpolate2(piv, opt)
        o.crossinterpolate2(piv, opt)
        ctx = f"seed {seed} case {case} kind {kind} dims {dims} opt {vars(opt)}"
        for p in range(d):
            assert np.array_equal(g.i_set(p), o.i_set(p)), ctx
            assert np.array_equal(g.j_set(p), o.j_set(p)), ctx
        rg, eg = g.history()
        ro, eo = o.history()
        assert list(rg) == list(ro), ctx
        assert np.allclose(eg, eo, rtol=1e-9, atol=1e-300), ctx
        assert g.termination() == o.termination(), ctx
        pts = np.stack([rng.integers(0, dd, size=50) for dd in dims], axis=1)
        gv, ov = g.evaluate(pts), o.evaluate(pts)
        assert np.abs(gv - ov).max() <= 1e-9 * max(1.0, np.abs(ov).max()), ctx
