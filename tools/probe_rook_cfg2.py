"""configs[1] with PivotSearchStrategy::Rook (device-resident search): wall time per solve, host synchronisations per bond; under
rocprofv3 --kernel-trace the per-bond kernel pattern."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
from t4a_amd.functions import quantics_trig_exp
spec = quantics_trig_exp(20)
o = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=8, nsearch=0, max_nglobal_pivot=0, seed=42, pivot_search=t4a_amd.TCI2Options.ROOK)
for rep in range(4):
    t = t4a_amd.TensorCI2([2] * 20)
    t.set_function(spec)
    t0 = time.perf_counter()
    t.crossinterpolate2([[0] * 20], o)
    dt = time.perf_counter() - t0
    st = t.rook_stats() if hasattr(t, "rook_stats") else None
    print(f"solve {rep}: {dt * 1e3:.2f} ms, rank {max(t.link_dims())}, rook stats {st}", flush=True)
