"""Determinism of the widest single-XCD plans (64 matrix entries per lane, 1024 x 1024 matrices): five times BASELINE configs[3] size
grown from scratch + two forward / backward sweep2site pairs, digests of all index sets and bond errors must agree."""
import hashlib, sys, os, time
sys.path.insert(0, "tensor4all-rs_amd/python")
import numpy as np, t4a_amd
from t4a_amd.functions import quantics_osc2d
d4, chi4 = 40, 512
def run():
    t = t4a_amd.TensorCI2([2] * d4)
    t.set_function(quantics_osc2d(d4, k1=37, k2=53, k3=20011, eps=0.5, k4=1048583, delta=0.5))
    t.add_global_pivots([[0] * d4]); t.set_max_sample_value(1.0)
    o4 = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi4, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
    t.optimize(o4, final_sweep1site=False)
    for _ in range(2):
        t.sweep2site(True, o4); t.sweep2site(False, o4)
    h = hashlib.sha256()
    for p in range(d4):
        h.update(np.asarray(t.i_set(p), dtype=np.int64).tobytes()); h.update(np.asarray(t.j_set(p), dtype=np.int64).tobytes())
    h.update(np.asarray(t.bond_errors()).tobytes())
    return h.hexdigest()[:16]
t0 = time.time()
ds = [run() for _ in range(5)]
print(ds, "identical" if len(set(ds)) == 1 else "DIFFERENT", f"{time.time()-t0:.1f} s")
