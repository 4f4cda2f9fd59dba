"""configs[1] host-side phases: T4A_OPT_PROF=1 (host time between chains), T4A_HOST_PROFILE=1 (host part of fill_site_tensors)."""
import os, sys, time
sys.path.insert(0, "tensor4all-rs_amd/python")
import t4a_amd
from t4a_amd.functions import quantics_trig_exp
spec = quantics_trig_exp(20)
opt = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0)
best = 1e9
for rep in range(25):
    g = t4a_amd.TensorCI2([2] * 20)
    g.set_function(spec)
    t0 = time.perf_counter()
    g.crossinterpolate2([[0] * 20], opt)
    dt = time.perf_counter() - t0
    best = min(best, dt)
print(f"best of 25: {best*1e3:.3f} ms", flush=True)
