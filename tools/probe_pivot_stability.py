"""Do the pivots of a saturated sweep repeat?  cfg3 workload of bench.py grown to saturation, then K more iterations (half-sweeps):
after each one the nested index sets are hashed per bond and compared with those two half-sweeps earlier (same direction).  If the
sets have converged, the candidate matrix of a bond and therefore its whole pivot sequence repeat from sweep to sweep — the
premise of a rrLU that follows a PREDICTED pivot sequence and only verifies it."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import t4a_amd  # noqa: E402

n = bench.N_SITES
t = t4a_amd.TensorCI2([2] * n)
t.set_function(bench.patch_spec(0, 1))
t.add_global_pivots([[0] * n])
t.set_max_sample_value(1.0)


def opts(it):
    return t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=bench.CHI, max_iter=it, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)


def digests():
    out = []
    for s in range(n):
        out.append((hashlib.sha1(np.ascontiguousarray(t.i_set(s)).tobytes()).hexdigest()[:10],
                    hashlib.sha1(np.ascontiguousarray(t.j_set(s)).tobytes()).hexdigest()[:10]))
    return out


t.optimize(opts(10), final_sweep1site=False)
hist = [digests()]
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t.optimize(opts(1), final_sweep1site=False)
    hist.append(digests())
    same_prev = sum(1 for a, b in zip(hist[-1], hist[-2]) if a == b)
    same_two = sum(1 for a, b in zip(hist[-1], hist[-3]) if a == b) if len(hist) >= 3 else -1
    errs = np.asarray(t.bond_errors())
    print(f"half-sweep {k}: sites whose (I, J) equal the previous half-sweep's: {same_prev}/{n}, two half-sweeps ago: {same_two}/{n}; "
          f"max bond error {errs.max():.3e}", flush=True)
