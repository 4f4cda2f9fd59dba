"""svd_backend (tensor4all-tensorbackend/src/backend.rs:709-731) at small shapes: wall time of one call (upload, decomposition, three
downloads) and the residuals.  With a library built with -DT4A_DIAG_SWITCHES (T4A_GPU_LIB=...) the routes inside Engine::svd can be
selected from the environment: T4A_SVD_NO_PRECOND=1 (no QR in front), T4A_SVD_SMALL_N=n (one-launch kernel up to n columns)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
if label != "default" and not t4a_amd.diag_switches_enabled():
    raise SystemExit("this library was built without -DT4A_DIAG_SWITCHES: the environment switches are ignored, refusing to label the run")
rng = np.random.default_rng(0)
for (m, n) in [(64, 64), (48, 32), (128, 64), (64, 128), (96, 96), (200, 100), (3, 2), (33, 33), (65, 65), (224, 64), (200, 31), (96, 48), (16, 8)]:
    a = rng.standard_normal((m, n))
    t4a_amd.svd_backend(a)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        u, s, vt = t4a_amd.svd_backend(a)
        ts.append(time.perf_counter() - t0)
    sref = np.linalg.svd(a, compute_uv=False)
    t0 = time.perf_counter()
    for _ in range(20):
        np.linalg.svd(a, full_matrices=False)
    lap = (time.perf_counter() - t0) / 20
    print(f"[{label}] svd {m} x {n}: best {min(ts) * 1e6:8.1f} us, median {sorted(ts)[len(ts) // 2] * 1e6:8.1f} us (numpy {lap * 1e6:7.1f} us); "
          f"|s - s_numpy| / s_max {np.abs(s - sref).max() / sref[0]:.1e}, residual {float(np.abs((u * s) @ vt - a).max()):.1e}, "
          f"|U^T U - I| {float(np.abs(u.T @ u - np.eye(u.shape[1])).max()):.1e}, |V V^T - I| {float(np.abs(vt @ vt.T - np.eye(vt.shape[0])).max()):.1e}",
          flush=True)
