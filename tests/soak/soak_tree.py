"""Soak of the tree TCI driver (tensor4all-treetci/src/{optimize.rs:95-220, update.rs:22-115, proposer.rs}) against the CPU oracle: the body
of tests/test_gpu_tree.py::test_random_trees_and_options_match_oracle over many seeds and a wider range — random trees of 3 - 11 sites (every
new site attaches to a random earlier one), local dimensions 2 - 4, three function families through a Python callback, the three edge
proposers, random options (tolerance, iterations, rank cap, error normalisation, global pivot search).  Pivot tables of every subtree, the
rank / error histories, bond and pivot errors: IDENTICAL / 1e-11; the materialised tree tensor network at every point of the grid: 1e-9.
usage: python3 tests/soak/soak_tree.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402
import oracle_binding as ob  # noqa: E402
from oracle_binding import OracleTreeTCI2, TreeOptions  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def gopts(o):
    return t4a.TreeTciOptions(tolerance=o.tolerance, max_iter=o.max_iter, max_bond_dim=o.max_bond_dim, normalize_error=o.normalize_error,
                              enable_global_pivots=o.enable_global_pivots, nsearch=o.nsearch, max_nglobal_pivot=o.max_nglobal_pivot,
                              tol_margin_global_search=o.tol_margin_global_search, seed=o.seed)


fails = 0
refused_both = 0
max_rank = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(3, 12))
    dims = [int(rng.integers(2, 5)) for _ in range(n)]
    while int(np.prod(dims)) > 50000:  # (the whole grid is evaluated at the end)
        dims[int(rng.integers(0, n))] = 2
    edges = [(int(rng.integers(0, k)), k) for k in range(1, n)]
    w = rng.standard_normal(n)
    c = rng.standard_normal((n, n)) * 0.15
    kind = int(rng.integers(0, 3))

    def f(idx, w=w, c=c, kind=kind):
        x = np.asarray(idx, dtype=np.float64)
        if kind == 0:
            return float(np.cos(w @ x) + 0.3)
        if kind == 1:
            return float(1.0 / (1.5 + np.abs(w) @ x + x @ np.abs(c) @ x))
        return float(np.exp(-0.2 * (w @ x) ** 2) + 0.1 * x[0] * x[-1])

    prop = int(rng.choice([0, 0, 1, 2]))
    opt = TreeOptions(tolerance=float(10.0 ** rng.integers(-10, -3)), max_iter=int(rng.integers(2, 7)),
                      max_bond_dim=None if rng.random() < 0.5 else int(rng.integers(1, 9)),
                      normalize_error=bool(rng.integers(0, 2)), enable_global_pivots=bool(rng.integers(0, 2)),
                      nsearch=int(rng.integers(1, 6)), max_nglobal_pivot=int(rng.integers(1, 4)), seed=int(rng.integers(0, 100)))
    ctx = f"seed {seed0 + case}: dims {dims} edges {edges} kind {kind} proposer {prop} opts {vars(opt)}"
    try:
        g = t4a.TreeTCI2(dims, edges)
        g.set_function(f)
        o = OracleTreeTCI2(dims, edges, f)
        for t in (g, o):
            t.set_proposer(prop, 11 + case % 7)
        first = [int(rng.integers(0, d)) for d in dims]
        if f(first) == 0.0:
            first = [0] * n
        og = o.crossinterpolate2([first], opt)
        gg = g.crossinterpolate2([first], gopts(opt))
        bad = []
        if gg[0] != og[0]:
            bad.append(f"rank history {gg[0]} vs {og[0]}")
        elif not np.allclose(gg[1], og[1], rtol=0, atol=1e-11):
            bad.append("error history")
        for (u, v) in edges:
            l, r = o.subregion_vertices(u, v)
            for key in (tuple(l), tuple(r)):
                if g.pivots(key).tolist() != o.pivots(key).tolist():
                    bad.append(f"pivot table of subtree {key}")
                else:
                    max_rank = max(max_rank, len(o.pivots(key).tolist()))
        if g.max_sample_value() != o.max_sample_value():
            bad.append("max_sample_value")
        scale = max(1.0, o.max_sample_value())
        if not np.abs(g.bond_errors() - o.bond_errors()).max() <= 1e-10 * scale:
            bad.append("bond errors")
        ge, oe = g.pivot_errors(), o.pivot_errors()
        if len(ge) != len(oe) or (len(oe) and not np.abs(ge - oe).max() <= 1e-10 * scale):
            bad.append("pivot errors")
        if not bad:
            center = int(rng.integers(0, n))
            try:
                o.materialize(center)
                o_ok = True
            except ob.OracleError:
                o_ok = False
            if not o_ok:
                # (two sides of an edge hold different pivot counts after a saturated stop: to_treetn refuses, materialize.rs:40-48 — on both sides)
                try:
                    g.materialize(center)
                    bad.append("the oracle refuses to materialise, the device does not")
                except t4a.T4aError:
                    refused_both += 1
            else:
                g.materialize(center)
                pts = np.array(list(itertools.product(*[range(d) for d in dims])))
                gv, ov = g.evaluate(pts), o.evaluate(pts)
                if not np.abs(gv - ov).max() <= 1e-9 * max(1.0, np.abs(ov).max()):
                    bad.append(f"values differ by {np.abs(gv - ov).max():.2e}")
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {'; '.join(bad[:4])}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; largest pivot table {max_rank}; materialisation refused on both sides {refused_both}; "
      f"{time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
