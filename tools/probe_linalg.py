"""Thin QR and thin SVD (qr_backend / svd_backend, tensor4all-tensorbackend/src/backend.rs:709-760) at the shapes the compression
path issues; run under `rocprofv3 --kernel-trace` and summarise with tools/linalg_trace_summary.py for device times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402

rng = np.random.default_rng(0)
for (m, n) in [(512, 256), (300, 200), (64, 64)]:
    a = rng.standard_normal((m, n))
    for what in ("qr", "svd"):
        f = t4a_amd.qr_backend if what == "qr" else t4a_amd.svd_backend
        f(a)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = f(a)
            ts.append(time.perf_counter() - t0)
        k = min(m, n)
        flops = (2.0 * m * n * n - 2.0 / 3.0 * n ** 3) * (2 if what == "qr" else 1) if what == "qr" else 6.0 * m * n * n * 8
        if what == "qr":
            q, r = out
            err = float(np.abs(q @ r - a).max())
        else:
            u, s, vt = out
            err = float(np.abs((u * s) @ vt - a).max())
        print(f"{what} {m} x {n}: call {min(ts) * 1e6:9.1f} us (host round trip incl. upload / download), ~{flops / min(ts) / 1e9:7.1f} GF/s, max residual {err:.2e}",
              flush=True)

# graded / rank-deficient inputs (what the unfoldings of a tensor train under compression look like): singular values 2^-i, a rank-40
# matrix over ten decades, the transposed (wide) orientation
for label, make in [("sigma_i = 2^-i", lambda m, n: (np.linalg.qr(rng.standard_normal((m, n)))[0] * 2.0 ** -np.arange(n)) @ np.linalg.qr(rng.standard_normal((n, n)))[0]),
                    ("rank 40, ten decades", lambda m, n: rng.standard_normal((m, 40)) @ np.diag(np.logspace(0, -10, 40)) @ rng.standard_normal((40, n)))]:
    for (m, n) in [(512, 256), (256, 512)]:
        a = make(max(m, n), min(m, n))
        if m < n:
            a = a.T.copy()
        t4a_amd.svd_backend(a)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            u, s, vt = t4a_amd.svd_backend(a)
            ts.append(time.perf_counter() - t0)
        sref = np.linalg.svd(a, compute_uv=False)
        print(f"svd {m} x {n} [{label}]: call {min(ts) * 1e6:9.1f} us, max |s - s_numpy| / s_max {np.abs(s - sref).max() / sref[0]:.2e}, "
              f"max residual {float(np.abs((u * s) @ vt - a).max()):.2e}, |U^T U - I| {float(np.abs(u.T @ u - np.eye(u.shape[1])).max()):.2e}", flush=True)
