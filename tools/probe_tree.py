"""Tree TCI probe: the 2-variable oscillatory integrand on a "two-chain" tree (x bits and y bits as two chains joined at
their most significant bits) versus the interleaved linear chain; device time per optimisation, optional oracle check.
Usage: python tools/probe_tree.py [n_sites] [max_bond_dim] [oracle|-] [hard]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4a_amd
from t4a_amd.functions import quantics_osc2d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
with_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
hard = len(sys.argv) > 4 and sys.argv[4] == "hard"
f = quantics_osc2d(n, 37, 53, 2111, 0.5, 16411, 0.5) if hard else quantics_osc2d(n, eps=0.1)
two_chain = [(0, 1)] + [(s, s + 2) for s in range(n - 2)]
# y chain hanging off the middle of the x chain: a genuine degree-3 vertex (candidate rows d * chi * chi there)
t_tree = [(2 * (n // 4), 1)] + [(s, s + 2) for s in range(n - 2)]
chain = [(s, s + 1) for s in range(n - 1)]
rng = np.random.default_rng(7)
pts = rng.integers(0, 2, size=(2000, n))
import oracle_binding as ob
exact = ob.fn_eval(f, pts)
for name, edges in (("two-chain tree", two_chain), ("T tree", t_tree), ("interleaved chain", chain)):
    opt = t4a_amd.TreeTciOptions(tolerance=1e-9, max_iter=6, max_bond_dim=chi, enable_global_pivots=False)
    t = t4a_amd.TreeTCI2([2] * n, edges)
    t.set_function(f)
    t0 = time.perf_counter()
    ranks, errors = t.crossinterpolate2([[0] * n], opt)
    t1 = time.perf_counter()
    t.materialize(0)
    t2 = time.perf_counter()
    err = np.abs(t.evaluate(pts) - exact).max()
    print(f"{name:18s} n={n} cap={chi}: sweeps={len(ranks)} ranks={ranks} err_est={errors[-1]:.2e} "
          f"optimize {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / (2 * len(ranks)):.1f} ms per edge pass) "
          f"materialize {1e3 * (t2 - t1):.1f} ms  max|f - tree| on 2000 pts = {err:.2e}", flush=True)
    if with_oracle:
        o = ob.OracleTreeTCI2([2] * n, edges, f)
        oo = ob.TreeOptions(tolerance=1e-9, max_iter=6, max_bond_dim=chi, enable_global_pivots=False)
        t0 = time.perf_counter()
        oranks, oerrors = o.crossinterpolate2([[0] * n], oo)
        t1 = time.perf_counter()
        same = all(t.pivots(k).tolist() == o.pivots(k).tolist()
                   for (u, v) in edges for k in o.subregion_vertices(u, v))
        print(f"   oracle: optimize {1e3 * (t1 - t0):.1f} ms, ranks={oranks}, pivot tables identical: {same}", flush=True)
