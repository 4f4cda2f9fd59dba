"""BASELINE configs[3] size (d = 40, chi = 512) at saturation: shapes of the last sweep and where the rrLU time of two half-sweeps
goes, per kernel instantiation."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402
from t4a_amd.functions import quantics_osc2d  # noqa: E402

d4, chi4 = 40, 512
t = t4a_amd.TensorCI2([2] * d4)
t.set_function(quantics_osc2d(d4, k1=37, k2=53, k3=20011, eps=0.5, k4=1048583, delta=0.5))
t.add_global_pivots([[0] * d4])
t.set_max_sample_value(1.0)
o4 = lambda it: t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi4, max_iter=it, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
t.optimize(o4(11), final_sweep1site=False)
t.profile_enable(True)
t.profile_reset()
t0 = time.perf_counter()
t.optimize(o4(2), final_sweep1site=False)
dt = time.perf_counter() - t0
vs = t.profile_variants()
print(f"full sweep {dt * 1e3:.1f} ms, chain {t.chain_stats()}")
print("shapes of the last half-sweep (M, N, rank):", [tuple(int(x) for x in s) for s in t.last_sweep_shapes()][::3])
for v in sorted(vs, key=lambda v: -v["ms"]):
    if v["code"] >= 10000000:
        continue
    print(f"{bench.rrlu_kernel_name(v['code']):55s} launches {v['launches']:4.0f} steps {v['steps']:7.0f} ms {v['ms']:8.3f} us/step {1e3 * v['ms'] / max(v['steps'], 1):.2f}")
