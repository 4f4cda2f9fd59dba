#!/usr/bin/env python3
"""Determinism soak for the inter-workgroup pivot exchange: the same from-scratch TCI2 run (cfg3 workload of bench.py,
d = 30, chi = 256) is repeated R times and the nested index sets of every repeat are hashed; any difference between
repeats means a race in the speculative column publication / key gather (the arithmetic itself is deterministic).

    python tools/soak_determinism.py [repeats=20] [sweeps=14]
"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench
import t4a_amd


def run(sweeps):
    tci = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tci.set_function(bench.patch_spec(0, 1))
    tci.add_global_pivots([[0] * bench.N_SITES])
    tci.set_max_sample_value(1.0)
    o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=bench.CHI, max_iter=sweeps, ncheck_history=10 ** 6, nsearch=0,
                            max_nglobal_pivot=0, seed=42)
    tci.optimize(o, final_sweep1site=False)
    h = hashlib.sha256()
    for s in range(bench.N_SITES):
        h.update(np.ascontiguousarray(np.asarray(tci.i_set(s), dtype=np.int64)).tobytes())
        h.update(np.ascontiguousarray(np.asarray(tci.j_set(s), dtype=np.int64)).tobytes())
    h.update(np.asarray(tci.pivot_errors(), dtype=np.float64).tobytes())
    return h.hexdigest(), max(tci.link_dims())


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    t4a_amd.set_device(0)
    ref = None
    t0 = time.time()
    for r in range(repeats):
        digest, chi = run(sweeps)
        if ref is None:
            ref = digest
            print(f"repeat 0: chi={chi} digest={digest[:16]}", flush=True)
        elif digest != ref:
            print(f"MISMATCH at repeat {r}: {digest[:16]} != {ref[:16]}")
            sys.exit(1)
    print(f"{repeats} repeats x {sweeps} half-sweeps identical ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
