#!/usr/bin/env python3
"""ACI probe: Hadamard product of two random chi-bond trains on d binary sites (solution rank up to chi^2) on the device,
wall time per sweep and the rank / error history.   python tools/probe_aci.py [d=24] [chi=16] [cap=0] [guard=0]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np

import t4a_amd


def rand_tt(d, chi, seed):
    rng = np.random.default_rng(seed)
    link = [min(2 ** (b + 1), 2 ** (d - b - 1), chi) for b in range(d - 1)]
    return [rng.standard_normal((1 if s == 0 else link[s - 1], 2, link[s] if s < d - 1 else 1)) / np.sqrt(2.0) for s in range(d)]


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    chi = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    guard = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    t4a_amd.set_device(0)
    a, b = t4a_amd.SimpleTensorTrain(rand_tt(d, chi, 1)), t4a_amd.SimpleTensorTrain(rand_tt(d, chi, 2))
    o = t4a_amd.AciOptions(tolerance=1e-10, max_bond_dim=cap or None, enable_global_guard=guard, max_iters=8)
    for rep in range(2):
        t0 = time.perf_counter()
        r = t4a_amd.elementwise_batched(t4a_amd.ACI_PRODUCT, [a, b], o)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {dt * 1e3:.1f} ms, {len(r.ranks)} sweeps ({dt * 1e3 / max(len(r.ranks), 1):.1f} ms/sweep), ranks {r.ranks}, "
              f"errors {[f'{e:.1e}' for e in r.errors]}, termination {r.termination}", flush=True)
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 2, size=(2000, d))
    exact = a.evaluate(pts) * b.evaluate(pts)
    print("max rel err on 2000 random points:", np.abs(r.tensor_train.evaluate(pts) - exact).max() / np.abs(exact).max())


if __name__ == "__main__":
    main()
