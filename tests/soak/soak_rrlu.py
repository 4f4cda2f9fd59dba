"""Soak of the hot path's own entry point against the oracle, BITWISE: t4a_gpu_rrlu_f64 (rrlu_mut, tensor4all-core/src/matrixlu.rs:735-819)
on random matrices (the LUCI factors on top of it are covered through TCI2 and the tensor-train soaks) — the generator of
tests/test_gpu_fuzz.py::test_rrlu_random_cases_bitwise widened: shapes 1 .. 420 (every one-wave / one-workgroup / single-XCD plan back to
back), 300 .. 900 and a share up to 1 536 (the multi-XCD kernels), degenerate shapes (1 x n, m x 1, 2 x 2), uniform / low rank / exact
integer ties / 15 decades of row scaling / column scaling / entries around 1e-300 and 1e300 / sparse / matrices with NaN or Inf planted, all
stop rules, both orthogonalities.  Pivot order, factored buffer, last error: bitwise (NaN entries: same positions; the sign bit of a NaN the
elimination generates is the architecture's); a refusal (NaN in L or U) on both sides or neither.
usage: python3 tests/soak/soak_rrlu.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402
import oracle_binding as ob  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
refused_both = 0
nan_cases = 0
by_size = {"<=64": 0, "<=420": 0, "<=900": 0, "<=1536": 0}
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    size = rng.random()
    if size < 0.08:
        m, n = [(1, int(rng.integers(1, 300))), (int(rng.integers(1, 300)), 1), (2, 2), (1, 1), (int(rng.integers(1, 9)), int(rng.integers(1, 9)))][int(rng.integers(0, 5))]
    elif size < 0.45:
        m, n = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    elif size < 0.88:
        m, n = int(rng.integers(1, 421)), int(rng.integers(1, 421))
    elif size < 0.97:
        m, n = int(rng.integers(300, 901)), int(rng.integers(300, 901))
    else:
        m, n = int(rng.integers(900, 1537)), int(rng.integers(900, 1537))
    big = max(m, n) > 420
    by_size["<=64" if max(m, n) <= 64 else "<=420" if not big else "<=900" if max(m, n) <= 900 else "<=1536"] += 1
    kind = int(rng.integers(0, 9))
    if kind == 0:
        a = rng.uniform(-1, 1, size=(m, n))
    elif kind == 1:
        r = int(rng.integers(1, max(2, min(m, n))))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    elif kind == 2:
        a = rng.integers(-2, 3, size=(m, n)).astype(float)
    elif kind == 3:
        a = rng.standard_normal((m, n)) * 10.0 ** rng.integers(-12, 3, size=(m, 1))
    elif kind == 4:
        a = rng.standard_normal((m, n)) * 10.0 ** rng.integers(-12, 3, size=(1, n))
    elif kind == 5:
        a = rng.standard_normal((m, n)) * float(rng.choice([1e-300, 1e-160, 1e150, 1e300]))
    elif kind == 6:
        a = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.05)
    elif kind == 7:
        a = rng.uniform(-1, 1, size=(m, n))
        for _ in range(int(rng.integers(1, 4))):
            a[int(rng.integers(0, m)), int(rng.integers(0, n))] = float(rng.choice([np.nan, np.inf, -np.inf]))
    else:
        a = np.outer(rng.integers(1, 5, size=m), rng.integers(1, 5, size=n)).astype(float)  # rank 1 with ties everywhere
    opts = {}
    if rng.random() < 0.5 or big:
        opts["max_bond_dim"] = int(rng.integers(1, min(m, n, 120 if big else 10 ** 9) + 1))  # (keeps the oracle fast on the large shapes)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        opts.update(rel_tol=0.0, abs_tol=0.0)
    elif mode == 1:
        opts.update(rel_tol=float(10.0 ** rng.integers(-14, -2)), abs_tol=0.0)
    elif mode == 2:
        opts.update(rel_tol=0.0, abs_tol=float(10.0 ** rng.integers(-10, 0)))
    opts["left_orthogonal"] = bool(rng.integers(0, 2))
    ctx = f"seed {seed0 + case} {m} x {n} kind {kind} opts {opts}"
    try:
        try:
            f, rp, cp, npiv, err = ob.rrlu(a, **opts)
            o_ref = None
        except ob.OracleError as exc:
            o_ref = exc
        try:
            lu = t4a.rrlu(a, **opts)
            g_ref = None
        except t4a.T4aError as exc:
            g_ref = exc
        if o_ref is not None or g_ref is not None:
            if o_ref is not None and g_ref is not None and g_ref.code == t4a.NAN_ENCOUNTERED:
                refused_both += 1
            else:
                fails += 1
                print(f"FAIL {ctx}: device {'raised ' + str(g_ref)[:70] if g_ref else 'ok'}, oracle {'raised ' + str(o_ref)[:70] if o_ref else 'ok'}", flush=True)
            continue
        bad = []
        if lu.npivots() != npiv:
            bad.append(f"npivots {lu.npivots()} vs {npiv}")
        elif not (np.array_equal(lu.row_permutation, rp) and np.array_equal(lu.col_permutation, cp)):
            bad.append("permutations")
        else:
            # bitwise, except the SIGN of a NaN the elimination itself generates (0 * inf, inf - inf): 0xFFF8... on the x86 host of the
            # oracle, 0x7FF8... on the GPU — architecture-defined; the NaN POSITIONS must coincide (seed 2009: 24 such entries)
            gn, on = np.isnan(lu.factored), np.isnan(f)
            if not np.array_equal(gn, on):
                bad.append("NaN positions of the factored buffer")
            elif not np.array_equal(lu.factored.view(np.uint64)[~gn], f.view(np.uint64)[~on]):
                bad.append("factored buffer")
            elif gn.any():
                nan_cases += 1
        if not bad and not (lu.error == err or (np.isnan(lu.error) and np.isnan(err))):
            bad.append(f"last error {lu.error} vs {err}")
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {bad[0]}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; shapes {by_size}; refused on both sides (NaN in L / U) {refused_both}; accepted results holding NaN {nan_cases}; {time.perf_counter() - t0:.1f} s",
      flush=True)
sys.exit(1 if fails else 0)
