"""Fused small-problem fill (fill_small_kernel) against the general path: site tensors bitwise, wall time of configs[1].
The T4A_NO_SMALL_FILL arm needs a library built with -DT4A_DIAG_SWITCHES (T4A_GPU_LIB); run once per arm and compare the hashes:
    T4A_GPU_LIB=.../libt4a_gpu_diag.so [T4A_NO_SMALL_FILL=1] python tools/probe_small_fill.py"""
import sys, time, hashlib
import numpy as np
sys.path.insert(0, "tensor4all-rs_amd/python")
import t4a_amd
from t4a_amd.functions import quantics_trig_exp, quantics_osc2d

def run(spec, L, opt, reps):
    best, cores = 1e9, None
    for rep in range(reps):
        g = t4a_amd.TensorCI2([2] * L)
        g.set_function(spec)
        t0 = time.perf_counter()
        g.crossinterpolate2([[0] * L], opt)
        best = min(best, time.perf_counter() - t0)
        if cores is None:
            cores = [np.asarray(g.site_tensor(s)) for s in range(L)]
    return best, cores

h = hashlib.sha256()
opt = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0)
t, cores = run(quantics_trig_exp(20), 20, opt, 25)
print(f"configs[1] best of 25: {t*1e3:.3f} ms, ranks {[c.shape[-1] for c in cores]}", flush=True)
for c in cores:
    h.update(np.ascontiguousarray(c).tobytes())
opt = t4a_amd.TCI2Options(tolerance=1e-10, max_bond_dim=24, max_iter=6, nsearch=0, max_nglobal_pivot=0)
t, cores = run(quantics_osc2d(16), 16, opt, 3)
print(f"osc2d L=16 chi<=24: {t*1e3:.3f} ms, ranks {[c.shape[-1] for c in cores]}", flush=True)
for c in cores:
    h.update(np.ascontiguousarray(c).tobytes())
print("sha256 of all site tensors:", h.hexdigest(), flush=True)
