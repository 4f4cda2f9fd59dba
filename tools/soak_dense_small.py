"""Soak of the tolerance-level dense entry points next to the SVD (tensor4all-tensorbackend/src/backend.rs): qr_backend (:742), solve_matrix
(:865), triangular_solve_matrix (:924) and mat_mul (matrix.rs:1488) on random shapes and on inputs far from unit scale, rank deficient or
graded — the families that exposed the overflow in the SVD's pair test (profiles/r06_svd_small.txt).  Checks against numpy.
usage: python3 tools/soak_dense_small.py N [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
BIG = int(os.environ.get("T4A_SOAK_BIG", "1"))  # (T4A_SOAK_BIG=4: shapes up to four times larger — QR panels beyond 560 rows, larger LU plans)
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scales = [1.0, 1.0, 1.0, 1e-120, 1e120, 1e-200, 1e150]
fails = 0
worst = {}


def note(what, err, limit, info):
    global fails
    worst[what] = max(worst.get(what, 0.0), err)
    if not (err <= limit):
        fails += 1
        print(f"FAIL {what} {info}: {err:.3e} > {limit:.1e}", flush=True)


t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    scale = scales[case % len(scales)]
    # ---- QR
    m = int(rng.integers(1, 260 * BIG))
    n = int(rng.integers(1, 260 * BIG))
    k = min(m, n)
    kind = case % 4
    if kind == 0:
        a = rng.standard_normal((m, n))
    elif kind == 1:
        a = rng.standard_normal((m, n)) * np.logspace(0, -float(rng.integers(1, 14)), n)
    elif kind == 2:
        r = int(rng.integers(1, k + 1))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
        a[:, int(rng.integers(0, n))] = 0.0
    else:
        a = np.zeros((m, n))
        a[: min(m, 3), :] = rng.standard_normal((min(m, 3), n))
    a = a * scale
    amax = float(np.abs(a).max()) or 1.0
    info = f"seed {seed0 + case} {m} x {n} kind {kind} scale {scale:g}"
    q, r = t4a_amd.qr_backend(a)
    ok = bool(np.all(np.isfinite(q)) and np.all(np.isfinite(r)))
    note("qr finite", 0.0 if ok else 1.0, 0.0, info)
    if ok:
        note("qr |QR - A| / (amax k)", float(np.abs(q @ (r / amax) - a / amax).max() / k), 1e-13, info)
        note("qr |Q^T Q - I|", float(np.abs(q.T @ q - np.eye(k)).max()), 1e-11, info)
        note("qr tril(R)", float(np.abs(np.tril(r, -1)).max()), 0.0, info)
    # ---- solve (well conditioned: a random orthogonal matrix times a mild diagonal) and triangular solves
    ns = int(rng.integers(1, 200 * BIG))
    nrhs = int(rng.integers(1, 64))
    qq, _ = np.linalg.qr(rng.standard_normal((ns, ns)))
    A = (qq * rng.uniform(0.5, 2.0, ns)) @ np.linalg.qr(rng.standard_normal((ns, ns)))[0] * scale
    X = rng.standard_normal((ns, nrhs))
    B = ((A / scale) @ X) * scale
    info = f"seed {seed0 + case} n {ns} nrhs {nrhs} scale {scale:g}"
    xs = t4a_amd.solve_matrix(A, B)
    ok = bool(np.all(np.isfinite(xs)))
    note("solve finite", 0.0 if ok else 1.0, 0.0, info)
    if ok:
        note("solve |x - x_ref| / |x|", float(np.abs(xs - X).max() / np.abs(X).max()), 1e-10, info)
    T = np.tril(rng.standard_normal((ns, ns))) + np.eye(ns) * (3.0 + ns ** 0.5)
    for lower in (True, False):
        for transpose_a in (False, True):
            Tm = (T if lower else T.T) * scale
            op = Tm.T if transpose_a else Tm
            Bt = (op / scale) @ X * scale
            xt = t4a_amd.triangular_solve_matrix(Tm, Bt, True, lower, transpose_a, False)
            ok = bool(np.all(np.isfinite(xt)))
            note("trsm finite", 0.0 if ok else 1.0, 0.0, info + f" lower {lower} trans {transpose_a}")
            if ok:
                note("trsm |x - x_ref| / |x|", float(np.abs(xt - X).max() / np.abs(X).max()), 1e-10, info + f" lower {lower} trans {transpose_a}")
    # ---- mat_mul against numpy (the reference's own tests use exact small-integer products; here: relative to |A| |B| k)
    mm, kk, nn = (int(rng.integers(1, 300 * BIG)) for _ in range(3))
    sa = scale if abs(np.log10(scale)) <= 120 else 1.0  # (the product of two 1e150 matrices is not representable)
    Am = rng.standard_normal((mm, kk)) * sa
    Bm = rng.standard_normal((kk, nn)) / sa
    C = t4a_amd.mat_mul(Am, Bm)
    note("mat_mul |C - C_ref| / k", float(np.abs(C - Am @ Bm).max() / kk), 1e-14, f"seed {seed0 + case} {mm} x {kk} x {nn} scale {sa:g}")
print(f"{N} cases from seed {seed0}: {fails} failures; " + "; ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())) + f"; {time.perf_counter() - t0:.1f} s",
      flush=True)
sys.exit(1 if fails else 0)
