"""The 7-site sample tree (graph/tests.rs) through TreeTCI on the device: wall time per solve; under rocprofv3 --kernel-trace the kernel
pattern of an edge update."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
from t4a_amd.functions import lorentz
dims = [4] * 7
edges = [(0, 1), (1, 2), (1, 3), (3, 4), (4, 5), (4, 6)]
opt = t4a_amd.TreeTciOptions(tolerance=1e-9, max_iter=6, max_bond_dim=16, enable_global_pivots=False)
for rep in range(4):
    t = t4a_amd.TreeTCI2(dims, edges)
    t.set_function(lorentz(dims))
    t0 = time.perf_counter()
    ranks, errors = t.crossinterpolate2([[0] * 7], opt)
    t1 = time.perf_counter()
    t.materialize(0)
    t2 = time.perf_counter()
    print(f"solve {rep}: crossinterpolate2 {(t1 - t0) * 1e3:.3f} ms ({len(ranks)} sweeps, rank {ranks[-1]}), materialize {(t2 - t1) * 1e3:.3f} ms", flush=True)
