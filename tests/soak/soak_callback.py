"""Soak of the HOST-CALLBACK route of crossinterpolate2 (the reference's `f: Fn(&MultiIndex) -> T` / `batched_f`, tensorci2.rs:1513-1524,
:1859-1893) against the built-in device functor of the SAME function, device against device: a Python batch callback that returns the
oracle's evaluation of a built-in family (bit-identical to the device functor by construction) must drive the run to
exactly the same index sets, histories, termination and max_sample_value, and to the same site tensors (1e-12: the per-bond path and the
device chain issue the same kernels).  The callback route has its own plumbing — per-bond candidate matrices, index decoding into the
(n_points, n_sites) batch, upload of the evaluated matrix, the length check on what the callback returns, optional callback threads
(t4a_gpu_tci2_set_callback_threads) — and until now only fixed-case tests.  Random families, sizes (rank up to ~60), options incl. Rook
and global pivot search, 1 - 3 initial pivots, 1 / 2 / 4 callback threads.
usage: python3 tests/soak/soak_callback.py N [seed0]     (test infrastructure: oracle fn_eval inside the callback; not collected by pytest)"""
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
calls = points = 0
max_rank = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        d = int(rng.integers(3, 8))
        dims = [int(rng.integers(2, 5)) for _ in range(d)]
        spec = lorentz(dims)
    elif kind == 1:
        d = 2 * int(rng.integers(3, 10))
        dims = [2] * d
        spec = quantics_osc2d(d, k1=int(rng.integers(1, 30)), k2=int(rng.integers(1, 30)), k3=int(rng.integers(1, 500)), eps=float(rng.choice([0.1, 0.5])),
                              k4=int(rng.integers(1, 3000)), delta=float(rng.choice([0.0, 0.5])))
    else:
        d = int(rng.integers(6, 20))
        dims = [2] * d
        spec = quantics_trig_exp(d)
    rook = bool(rng.random() < 0.2)
    opt = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-11, -5)), max_iter=int(rng.integers(2, 7)),
                          max_bond_dim=(None if rng.random() < 0.2 else int(rng.integers(4, 30 if rook else 60))), pivot_search=1 if rook else 0,
                          normalize_error=bool(rng.integers(0, 2)), sweep_strategy=int(rng.integers(0, 3)), strictly_nested=bool(rng.random() < 0.3),
                          ncheck_history=int(rng.integers(1, 4)), nsearch=int(rng.integers(0, 3)) if rng.random() < 0.3 else 0,
                          max_nglobal_pivot=int(rng.integers(0, 3)), seed=int(rng.integers(0, 1000)))
    threads = int(rng.choice([1, 1, 2, 4]))
    piv = [[0] * d] + [[int(rng.integers(0, dd)) for dd in dims] for _ in range(int(rng.integers(0, 3)))]
    ctx = f"seed {seed0 + case} kind {kind} dims {dims if kind == 0 else d} threads {threads} opt {vars(opt)}"
    try:
        a = t4a.TensorCI2(dims)
        a.set_function(spec)

        def scalar(i, spec=spec):
            return float(ob.fn_eval(spec, np.asarray([i]))[0])
        scalar.batched = lambda pts, spec=spec: ob.fn_eval(spec, np.asarray(pts))
        b = t4a.TensorCI2(dims)
        b.set_function(scalar)
        if threads > 1:
            b.set_callback_threads(threads)  # (a Python callable serialises on the interpreter lock: this exercises the splitting, not speed)
        a.crossinterpolate2(piv, opt)
        b.crossinterpolate2(piv, opt)
        calls += b.n_callback_calls
        bad = []
        for p in range(d):
            if not (np.array_equal(a.i_set(p), b.i_set(p)) and np.array_equal(a.j_set(p), b.j_set(p))):
                bad.append(f"index sets at site {p}")
                break
        if not bad:
            if list(a.history()[0]) != list(b.history()[0]):
                bad.append(f"rank history {a.history()[0]} vs {b.history()[0]}")
            elif not (np.array_equal(a.history()[1], b.history()[1]) if not rook else np.allclose(a.history()[1], b.history()[1], rtol=1e-9, atol=1e-12)):
                bad.append(f"error history {list(a.history()[1])} vs {list(b.history()[1])}")
            elif a.termination() != b.termination() or a.max_sample_value() != b.max_sample_value():
                bad.append("termination / max_sample_value")
        if not bad:
            max_rank = max(max_rank, max(a.link_dims()))
            for p in range(d):
                x, y = a.site_tensor(p), b.site_tensor(p)
                if x.shape != y.shape or not np.abs(x - y).max() <= (1e-8 if rook else 1e-12) * max(1.0, np.abs(y).max()):
                    bad.append(f"site tensor {p} differs by {np.abs(x - y).max():.2e}" if x.shape == y.shape else f"site tensor {p}: shapes {x.shape} vs {y.shape}")
                    break
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {bad[0][:300]}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; {calls} callback invocations; largest link dimension {max_rank}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
