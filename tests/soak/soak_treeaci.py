"""Soak of the TreeACI local update (crates/tensor4all-treeaci/src/local_update: candidate values from row / column frames of K inputs, LUCI)
against the oracle: the assertions of tests/test_gpu_treeaci.py::test_random_frames_match_the_oracle over random shapes — one to four inputs
with cut bonds 1 - 9, 1 - 220 rows x 1 - 160 columns, product / sum / a Python callback operator, both orthogonalities, tolerance or rank cap
or an unscaled tolerance.  Local values BITWISE (same summation order), sampled scale, rank, row / column indices and pivot errors identical,
factors to 1e-9 of the largest entry and reproducing the local values.
usage: python3 tests/soak/soak_treeaci.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402
import oracle_binding as ob  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
max_rank = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    K = int(rng.integers(1, 5))
    bonds = [int(rng.integers(1, 10)) for _ in range(K)]
    rows, cols = int(rng.integers(1, 221)), int(rng.integers(1, 161))
    rf = [rng.standard_normal((b, rows)) for b in bonds]
    cf = [rng.standard_normal((b, cols)) for b in bonds]
    which = int(rng.integers(0, 3))
    if which == 0:
        dop = oop = ob.ACI_PRODUCT
    elif which == 1:
        dop = oop = ob.ACI_SUM
    else:
        wts = rng.uniform(-1, 1, K)
        dop = oop = lambda v, wts=wts: np.tensordot(wts, v, axes=(0, 0)) + 0.25 * v[0] * v[-1]  # noqa: E731
    left = bool(rng.integers(0, 2))
    kw = [dict(tolerance=float(10.0 ** rng.integers(-12, -4))), dict(max_bond_dim=int(rng.integers(1, 12))),
          dict(tolerance=float(10.0 ** rng.integers(-8, -2)), scale_tolerance=False)][int(rng.integers(0, 3))]
    ctx = f"seed {seed0 + case} K {K} bonds {bonds} {rows} x {cols} op {['product', 'sum', 'callback'][which]} left {left} {kw}"
    try:
        d = t4a.treeaci_local_update(rf, cf, dop, left_orthogonal=left, **kw)
        o = ob.treeaci_local_update(rf, cf, oop, left_orthogonal=left, **kw)
        bad = []
        if not np.array_equal(d.local_values.view(np.uint64), o.local_values.view(np.uint64)):
            bad.append("local values not bitwise equal")
        elif d.sampled_scale != o.sampled_scale:
            bad.append("sampled scale")
        elif d.rank != o.rank or d.row_indices != o.row_indices or d.col_indices != o.col_indices:
            bad.append(f"rank / indices: {d.rank} vs {o.rank}")
        elif not np.array_equal(d.pivot_errors, o.pivot_errors):
            bad.append("pivot errors")
        else:
            max_rank = max(max_rank, d.rank)
            scale = max(1.0, np.abs(o.left).max(), np.abs(o.right).max())
            if not (np.abs(d.left - o.left).max() <= 1e-9 * scale and np.abs(d.right - o.right).max() <= 1e-9 * scale):
                # (factors are solves against the pivot block: judge them by what they reproduce when they are not close entry by entry)
                rec_d = np.abs((d.left @ d.right).reshape(-1, order="F") - d.local_values).max()
                rec_o = np.abs((o.left @ o.right).reshape(-1, order="F") - o.local_values).max()
                if not rec_d <= max(10.0 * rec_o, 1e-9 * max(1.0, np.abs(o.local_values).max())):
                    bad.append(f"factors: device reproduces the local values to {rec_d:.2e}, the oracle to {rec_o:.2e}")
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {bad[0]}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; largest rank {max_rank}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
