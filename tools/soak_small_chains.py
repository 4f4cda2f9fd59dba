"""Robustness of the small-problem paths (one-wave kernel, persistent half-sweep, chained 1-site sweep with batched site tensors):
R repeats of (a) BASELINE configs[1] and (b) an oscillatory 14-site problem grown from scratch with a final 1-site sweep and
make_canonical — every repeat must reproduce the first one's index sets, errors and site tensors BITWISE (any race between the
phases of the persistent workgroup, a stale scalar-cache line, a prefetch overwriting a list that is still read, would show up as a
differing digest).   usage: python tools/soak_small_chains.py [R]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402
from t4a_amd.functions import quantics_osc2d, quantics_trig_exp  # noqa: E402


def digest(t, n):
    h = hashlib.sha256()
    for s in range(n):
        h.update(np.ascontiguousarray(t.site_tensor(s)).tobytes())
        h.update(np.asarray(t.i_set(s), dtype=np.int64).tobytes())
        h.update(np.asarray(t.j_set(s), dtype=np.int64).tobytes())
    h.update(np.asarray(t.bond_errors()).tobytes())
    h.update(np.asarray(t.pivot_errors()).tobytes())
    return h.hexdigest()[:16]


def run_a():
    n = 20
    t = t4a_amd.TensorCI2([2] * n)
    t.set_function(quantics_trig_exp(n))
    t.crossinterpolate2([[0] * n], t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0))
    return digest(t, n), t.chain_stats()


def run_b():
    n = 14
    t = t4a_amd.TensorCI2([2] * n)
    t.set_function(quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.3))
    t.crossinterpolate2([[0] * n, [1, 0] * (n // 2)], t4a_amd.TCI2Options(tolerance=1e-9, max_bond_dim=24, max_iter=6, nsearch=0, max_nglobal_pivot=0))
    t.make_canonical(1e-10, 1e-13, 16)
    return digest(t, n), t.chain_stats()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.perf_counter()
    ref = None
    bad = 0
    for r in range(reps):
        cur = (run_a(), run_b())
        if ref is None:
            ref = cur
            print("reference:", ref, flush=True)
        elif (cur[0][0], cur[1][0]) != (ref[0][0], ref[1][0]):
            bad += 1
            print(f"repeat {r}: DIFFERENT {cur}", flush=True)
    print(f"soak_small_chains.py {reps}: {reps} repeats x (configs[1] + a 14-site run with a final 1-site sweep and make_canonical): "
          f"{bad} mismatches, {time.perf_counter() - t0:.1f} s", flush=True)
    sys.exit(1 if bad else 0)
