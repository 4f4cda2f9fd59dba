"""Soak of the adaptive patching driver (tensor4all-partitionedtt/src/adaptive_interpolation.rs:151-330: FIFO patch queue, accept rule :417,
split order :303-326, candidate pivots :429-505, pivot recycling :537) against the CPU oracle — BASELINE config 5's semantics on random
problems: built-in functions (the 2-variable oscillatory quantics integrand with random wave numbers on 8 - 16 bits, cos(10 x) exp(-x), a
Lorentzian on sites of dimension 2 - 4) and a Python callback, rank caps far below / around / above the rank, random tolerance, patch order
(a random permutation or the default), 1 - 4 initial pivots per patch, recycling on / off, 1 - 3 random initial pivots.
The patch queue's outcome: the same NUMBER of patches, the same projectors IN THE SAME ORDER (identical), the same link dimensions per
patch, cores to 1e-8, values at random points to 1e-8 against the oracle's patches.
usage: python3 tests/soak/soak_patch.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402
import oracle_binding as ob  # noqa: E402
from t4a_amd.functions import lorentz, quantics_osc2d, quantics_trig_exp  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
n_patches_total = 0
most_patches = 0
both_raised = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        n = 2 * int(rng.integers(4, 9))
        dims = [2] * n
        f = quantics_osc2d(n, k1=int(rng.integers(1, 12)), k2=int(rng.integers(1, 12)), k3=int(rng.integers(1, 60)), eps=float(rng.choice([0.1, 0.5])),
                           k4=int(rng.integers(1, 200)), delta=float(rng.choice([0.0, 0.5])))
    elif kind == 1:
        n = int(rng.integers(6, 15))
        dims = [2] * n
        f = quantics_trig_exp(n)
    elif kind == 2:
        n = int(rng.integers(3, 7))
        dims = [int(rng.integers(2, 5)) for _ in range(n)]
        f = lorentz(dims)
    else:
        n = int(rng.integers(3, 6))
        dims = [int(rng.integers(2, 5)) for _ in range(n)]
        a = rng.uniform(0.1, 1.0, n)
        b = rng.uniform(0.0, 0.4, (n, n))

        def f(i, a=a, b=b):
            x = np.asarray(i, dtype=np.float64)
            return float(np.cos(a @ x + x @ b @ x) + 0.05 * x[0] * x[-1])
    opt = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-10, -4)), max_bond_dim=int(rng.integers(2, 20)), max_iter=int(rng.integers(3, 12)),
                          nsearch=0, max_nglobal_pivot=0, normalize_error=bool(rng.integers(0, 2)), strictly_nested=bool(rng.random() < 0.2),
                          seed=int(rng.integers(0, 100)))
    kw = dict(patch_order=(None if rng.random() < 0.5 else [int(v) for v in rng.permutation(n)]), n_initial_pivots=int(rng.integers(1, 5)),
              recycle_pivots=bool(rng.integers(0, 2)))
    piv = [[int(rng.integers(0, d)) for d in dims] for _ in range(int(rng.integers(1, 4)))]
    if kind != 3:
        piv[0] = [0] * n
    ctx = f"seed {seed0 + case} kind {kind} dims {dims if kind >= 2 else n} opt {vars(opt)} {kw} pivots {piv}"
    try:
        try:
            o = ob.adaptiveinterpolate(f, dims, piv, opt, **kw)
            o_err = None
        except ob.OracleError as exc:
            o_err = exc
        try:
            g = t4a.adaptiveinterpolate(f, dims, piv, opt, **kw)
            g_err = None
        except t4a.T4aError as exc:
            g_err = exc
        if o_err is not None or g_err is not None:
            if o_err is not None and g_err is not None:
                both_raised += 1
            else:
                fails += 1
                print(f"FAIL {ctx}: device {'raised ' + str(g_err)[:80] if g_err else 'ok'}, oracle {'raised ' + str(o_err)[:80] if o_err else 'ok'}", flush=True)
            continue
        bad = []
        if len(g) != len(o):
            bad.append(f"{len(g)} patches vs {len(o)}")
        else:
            n_patches_total += len(g)
            most_patches = max(most_patches, len(g))
            for k in range(len(g)):
                if g.projector(k) != o.projector(k):
                    bad.append(f"projector of patch {k}: {g.projector(k)} vs {o.projector(k)}")
                    break
                gc, oc = g.patch(k).site_tensors(), o.cores(k)
                for s, (a_, b_) in enumerate(zip(gc, oc)):
                    if a_.shape != b_.shape:
                        bad.append(f"patch {k} site {s}: shape {a_.shape} vs {b_.shape}")
                        break
                    if not np.abs(a_ - b_).max() <= 1e-8 * max(1.0, np.abs(b_).max()):
                        bad.append(f"patch {k} site {s}: cores differ by {np.abs(a_ - b_).max():.2e}")
                        break
                if bad:
                    break
            if not bad:
                pts = np.stack([rng.integers(0, d, size=200) for d in dims], axis=1)
                gv, ov = g.evaluate(pts), o.evaluate(pts)
                if not np.abs(gv - ov).max() <= 1e-8 * max(1.0, np.abs(ov).max()):
                    bad.append(f"values differ by {np.abs(gv - ov).max():.2e}")
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {'; '.join(bad[:3])}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; {n_patches_total} patches compared, at most {most_patches} in one run; both sides raised "
      f"{both_raised}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
