"""fill_site_tensors alone (no bond chain beside it) on a saturated handle: wall time per call (the call returns after the device has
finished) for BASELINE configs[2] (d = 30, chi = 256) and configs[3] (d = 40, chi = 512).  T4A_NO_FUSED_SOLVE=1: the two-step path."""
import sys
import time
sys.path.insert(0, "tensor4all-rs_amd/python")
import numpy as np
import t4a_amd
from t4a_amd.functions import quantics_osc2d

CFG = [(30, 256, 10), (40, 512, 12)]
if len(sys.argv) > 1:
    CFG = [c for c in CFG if c[0] == int(sys.argv[1])]
for (n, chi, iters) in CFG:
    spec = quantics_osc2d(n, k1=37, k2=53, k3=2111, eps=0.5, k4=16411, delta=0.5)
    t = t4a_amd.TensorCI2([2] * n)
    t.set_function(spec)
    t.add_global_pivots([[0] * n])
    o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=iters, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0)
    t.optimize(o, final_sweep1site=False)
    t.fill_site_tensors()
    rng = np.random.default_rng(1)
    pts = rng.integers(0, 2, size=(64, n))
    ref = t.evaluate(pts)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        t.fill_site_tensors()
        v = t.evaluate(pts[:1])  # (completes the fill)
        ts.append(time.perf_counter() - t0)
    print(f"d={n} chi={chi} link_max={max(t.link_dims())} fill_site_tensors + one evaluation: best {min(ts)*1e3:.3f} ms, median {sorted(ts)[3]*1e3:.3f} ms; "
          f"checksum {float(np.abs(ref).sum()):.12e}", flush=True)
