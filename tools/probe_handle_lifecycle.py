"""Cost of a handle's life around a small solve (BASELINE configs[1]): construction, the solve, destruction (GPU box).
T4A_NO_POOL=1 shows the same without the process-wide resource cache (pool.hip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
from t4a_amd.functions import quantics_trig_exp
spec = quantics_trig_exp(20)
opt = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0)
acc = [0.0] * 4
for rep in range(12):
    t0 = time.perf_counter()
    g = t4a_amd.TensorCI2([2] * 20)
    t1 = time.perf_counter()
    g.set_function(spec)
    t2 = time.perf_counter()
    g.crossinterpolate2([[0] * 20], opt)
    t3 = time.perf_counter()
    del g
    t4 = time.perf_counter()
    if rep >= 2:
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += d * 1e3 / 10
print("construct %.3f ms, set_function %.3f ms, crossinterpolate2 %.3f ms, destroy %.3f ms" % tuple(acc))
