"""One case of tests/soak/soak_rrlu.py in detail: where the factored buffers of the device and the oracle differ.  usage: probe_rrlu_case.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, t4a_amd as t4a, oracle_binding as ob
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
# (the draws of soak_rrlu.py for a mid-size kind-7 case, replayed)
size = rng.random()
assert 0.08 <= size < 0.45, size
m, n = int(rng.integers(1, 65)), int(rng.integers(1, 65))
kind = int(rng.integers(0, 9))
assert kind == 7, kind
a = rng.uniform(-1, 1, size=(m, n))
for _ in range(int(rng.integers(1, 4))):
    v = float(rng.choice([np.nan, np.inf, -np.inf]))  # (the soak's `a[i, j] = v` evaluates the right-hand side first)
    i, j = int(rng.integers(0, m)), int(rng.integers(0, n))
    a[i, j] = v
    print("planted", v, "at", (i, j))
opts = {}
if rng.random() < 0.5:
    opts["max_bond_dim"] = int(rng.integers(1, min(m, n) + 1))
mode = int(rng.integers(0, 4))
opts["left_orthogonal"] = bool(rng.integers(0, 2))
print(m, n, opts, "mode", mode)
f, rp, cp, npiv, err = ob.rrlu(a, **opts)
lu = t4a.rrlu(a, **opts)
print("npivots", lu.npivots(), npiv, "perms equal", np.array_equal(lu.row_permutation, rp), np.array_equal(lu.col_permutation, cp), "errors", lu.error, err)
d = lu.factored.view(np.uint64) != f.view(np.uint64)
print("entries that differ bitwise:", int(d.sum()), "of", d.size)
idx = np.argwhere(d)
both_nan = np.isnan(lu.factored[d]) & np.isnan(f[d])
print("of those, NaN on both sides:", int(both_nan.sum()))
print("input row / column of the planted entries and the pivot:", "pivot row", int(rp[0]), "pivot col", int(cp[0]))
for (i, j) in idx[:12]:
    print((int(i), int(j)), "device", lu.factored[i, j], hex(int(lu.factored.view(np.uint64)[i, j])), "oracle", f[i, j], hex(int(f.view(np.uint64)[i, j])))
