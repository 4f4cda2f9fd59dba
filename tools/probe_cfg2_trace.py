"""Kernel-level view of BASELINE config 2 on the device (run under `rocprofv3 --kernel-trace --stats`): four solves of
d = 20, cos(10x) exp(-x), tol 1e-8, chi <= 64; prints the wall time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
from t4a_amd.functions import quantics_trig_exp
spec = quantics_trig_exp(20)
opt = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0)
for rep in range(4):
    g = t4a_amd.TensorCI2([2] * 20)
    g.set_function(spec)
    t0 = time.perf_counter()
    g.crossinterpolate2([[0] * 20], opt)
    print(f"solve {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms, iterations {len(g.history()[0])}, rank {g.rank()}", flush=True)
