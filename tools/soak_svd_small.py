"""Soak of svd_backend (tensor4all-tensorbackend/src/backend.rs:709-731) on the shapes of the one-launch Jacobi route (up to 96 columns,
jacobi_groups_kernel) and a band around its limits: random shapes, both orientations, spectra drawn from {gaussian, graded over 1 - 15
decades, rank-deficient with zero / repeated columns, clustered singular values, badly scaled columns, tiny and huge overall scale}.
Checks against numpy.linalg.svd: singular values to 1e-12 of the largest, reconstruction, orthonormal factors, ordering.
usage: python3 tools/soak_svd_small.py N [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
BIG = bool(int(os.environ.get("T4A_SOAK_BIG", "0")))
kinds = ["gaussian", "graded", "low_rank", "clustered", "column_scaled", "tiny", "huge", "orthogonal"]
worst = {"s": 0.0, "rec": 0.0, "u": 0.0, "v": 0.0}
fails = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    if BIG:
        n = int(rng.integers(90, 420))                     # (T4A_SOAK_BIG=1: the blocked and the QR-preconditioned routes)
        m = int(rng.integers(n, min(900, 3 * n) + 1))
    else:
        n = int(rng.integers(2, 101))                      # columns: the route takes up to 96
        m = int(rng.integers(n, min(240, 3 * n + 8) + 1))  # rows: up to 224 (n <= 64) / 96 (n > 64) stay on the route, the rest next to it
    if rng.integers(0, 2):
        m, n = n, m
    k = min(m, n)
    kind = kinds[case % len(kinds)]
    if kind == "gaussian":
        a = rng.standard_normal((m, n))
    elif kind in ("graded", "clustered", "orthogonal"):
        q1, _ = np.linalg.qr(rng.standard_normal((m, k)))
        q2, _ = np.linalg.qr(rng.standard_normal((n, k)))
        if kind == "graded":
            sv = np.logspace(0, -float(rng.integers(1, 16)), k)
        elif kind == "clustered":
            sv = np.repeat(rng.uniform(0.5, 2.0, size=(k + 3) // 4), 4)[:k] * (1.0 + 1e-13 * rng.standard_normal(k))
        else:
            sv = np.ones(k)
        a = (q1 * sv) @ q2.T
    elif kind == "low_rank":
        r = int(rng.integers(1, max(2, k // 2) + 1))
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
        a[:, int(rng.integers(0, n))] = 0.0
        a[int(rng.integers(0, m)), :] = 0.0
        if n > 1:
            a[:, n - 1] = a[:, 0]
    elif kind == "column_scaled":
        a = rng.standard_normal((m, n)) * np.logspace(0, -float(rng.integers(1, 13)), n)[rng.permutation(n)]
    elif kind == "tiny":
        a = rng.standard_normal((m, n)) * 1e-120
    else:
        a = rng.standard_normal((m, n)) * 1e120
    u, s, vt = t4a_amd.svd_backend(a)
    sref = np.linalg.svd(a, compute_uv=False)
    scale = sref[0] if sref[0] > 0 else 1.0
    e_s = float(np.abs(s - sref).max() / scale)
    e_rec = float(np.abs((u * s) @ vt - a).max() / (scale * k))
    e_u = float(np.abs(u.T @ u - np.eye(k)).max())
    e_v = float(np.abs(vt @ vt.T - np.eye(k)).max())
    ordered = bool(np.all(s >= 0.0) and np.all(np.diff(s) <= 1e-13 * scale))
    worst["s"] = max(worst["s"], e_s)
    worst["rec"] = max(worst["rec"], e_rec)
    worst["u"] = max(worst["u"], e_u)
    worst["v"] = max(worst["v"], e_v)
    if not (e_s <= 1e-12 and e_rec <= 1e-12 and e_u < 1e-10 and e_v < 1e-10 and ordered):
        fails += 1
        print(f"FAIL seed {seed0 + case} {kind} {m} x {n}: |ds| {e_s:.2e} rec {e_rec:.2e} U {e_u:.2e} V {e_v:.2e} ordered {ordered}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; worst |s - s_ref| / s_max {worst['s']:.2e}, reconstruction / (s_max k) {worst['rec']:.2e}, "
      f"|U^T U - I| {worst['u']:.2e}, |V V^T - I| {worst['v']:.2e}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
