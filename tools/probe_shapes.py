"""rrLU on very rectangular candidate matrices (tree hubs: d*chi*chi rows by d*chi columns).  Usage: probe_shapes.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4a_amd

shapes = [(2048, 64), (4096, 64), (8192, 128), (8300, 131), (128, 8192), (16384, 128), (32768, 256), (20000, 40), (40, 20000),
          (65535, 64), (3000, 3000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
rng = np.random.default_rng(1)
for (m, n) in shapes:
    r = min(m, n, 48)
    a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n)) + 1e-9 * rng.standard_normal((m, n))
    try:
        t0 = time.perf_counter()
        lu = t4a_amd.rrlu(a, max_bond_dim=64, rel_tol=1e-12)
        dt = time.perf_counter() - t0
        print(f"{m} x {n}: npivots={lu.npivots()} last_error={lu.last_pivot_error():.3e} {1e3 * dt:.1f} ms", flush=True)
    except t4a_amd.T4aError as e:
        print(f"{m} x {n}: FAILED {e}", flush=True)
