"""BASELINE configs[3] size (d = 40, chi = 512) through the PLAIN sweep2site API (tensorci2.rs:746-798: no history extras): the
mid-chain matrices are exactly 1024 x 1024 — what one XCD holds since the single-XCD kernels take 64 matrix entries per lane.
Prints the time of a forward + backward pair, the chain statistics and the rrLU time per kernel instantiation."""
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
o4 = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi4, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
t.optimize(o4, final_sweep1site=False)
for rep in range(3):
    t.profile_enable(True)
    t.profile_reset()
    t0 = time.perf_counter()
    t.sweep2site(True, o4)
    t.sweep2site(False, o4)
    dt = time.perf_counter() - t0
    vs = t.profile_variants()
    t.profile_enable(False)
    print(f"sweep2site forward + backward: {dt * 1e3:.1f} ms, chain {t.chain_stats()}", flush=True)
for v in sorted(vs, key=lambda v: -v["ms"])[:6]:
    if v["code"] < 10000000:
        print(f"{bench.rrlu_kernel_name(v['code']):55s} launches {v['launches']:4.0f} steps {v['steps']:7.0f} ms {v['ms']:8.3f} us/step {1e3 * v['ms'] / max(v['steps'], 1):.2f}")
