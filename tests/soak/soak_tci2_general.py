"""Soak of crossinterpolate2 (tensor4all-tensorci/src/tensorci2.rs:1513-1802) on MID-SIZE problems against the CPU oracle: ranks of 10 - 100,
6 - 24 sites of dimension 2 - 4, so the bonds run through the device chain, the one-wave / one-workgroup / single-XCD rrLU kernels, the
per-bond path, the Rook search and the global pivot search — the territory between the one-launch engine (tests/test_gpu_small.py: 20 000
random cases) and the five BASELINE configs.  Random options (tolerance, max_bond_dim, max_iter, sweep strategy, strictly nested sets, error
normalisation, history length, Full / Rook, global pivot search on / off); with the device chain on and off and the engine on and off.
Index sets, ranks, termination: IDENTICAL to the oracle; errors to 1e-9 relative (Rook: plus 1e-12 absolute); values at random points to 1e-9.
usage: python3 tests/soak/soak_tci2_general.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
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
ROOK_SHARE = float(os.environ.get("T4A_SOAK_ROOK_SHARE", "0.25"))  # (1.0: every case with PivotSearchStrategy::Rook)
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
t_dev = t_orc = 0.0
max_rank_seen = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        d = int(rng.integers(4, 9))
        dims = [int(rng.integers(2, 5)) for _ in range(d)]
        spec = lorentz(dims)
    elif kind in (1, 2):
        d = 2 * int(rng.integers(4, 13))
        dims = [2] * d
        spec = quantics_osc2d(d, k1=int(rng.integers(1, 60)), k2=int(rng.integers(1, 60)), k3=int(rng.integers(1, 3000)),
                              eps=float(rng.choice([0.1, 0.5])), k4=int(rng.integers(1, 20000)), delta=float(rng.choice([0.0, 0.5])))
    else:
        d = int(rng.integers(8, 25))
        dims = [2] * d
        spec = quantics_trig_exp(d)
    rook = bool(rng.random() < ROOK_SHARE)
    opt = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-12, -5)), max_iter=int(rng.integers(2, 8)),
                          max_bond_dim=(None if rng.random() < 0.15 else int(rng.integers(8, 40 if rook else 100))),
                          pivot_search=1 if rook else 0,
                          normalize_error=bool(rng.integers(0, 2)), sweep_strategy=int(rng.integers(0, 3)),
                          strictly_nested=bool(rng.integers(0, 2)), ncheck_history=int(rng.integers(1, 4)),
                          nsearch=int(rng.integers(0, 4)) if rng.random() < 0.4 else 0, max_nglobal_pivot=int(rng.integers(0, 4)),
                          seed=int(rng.integers(0, 1000)))
    chain = bool(rng.random() < 0.8)
    engine = bool(rng.random() < 0.5)
    ctx = f"seed {seed0 + case} kind {kind} dims {dims if kind == 0 else d} chain {chain} engine {engine} opt {vars(opt)}"
    try:
        g = t4a.TensorCI2(dims)
        g.set_function(spec)
        g.set_chain(chain, small_engine=engine)
        o = ob.OracleTCI2(dims)
        o.set_function(spec)
        o.set_pivot_search(opt.pivot_search)  # (the oracle takes the strategy from the handle, not from the options)
        piv = [[0] * d]
        ta = time.perf_counter()
        g.crossinterpolate2(piv, opt)
        tb = time.perf_counter()
        o.crossinterpolate2(piv, opt)
        tc = time.perf_counter()
        t_dev += tb - ta
        t_orc += tc - tb
        bad = []
        for p in range(d):
            if not np.array_equal(g.i_set(p), o.i_set(p)):
                bad.append(f"i_set[{p}]")
            if not np.array_equal(g.j_set(p), o.j_set(p)):
                bad.append(f"j_set[{p}]")
        rg, eg = g.history()
        ro, eo = o.history()
        if list(rg) != list(ro):
            bad.append(f"ranks {list(rg)} vs {list(ro)}")
        # (Full search: the bond errors are pivots of a bit-exact rrLU; Rook: residuals A[r, c] - A[r, J] (A[I, J]^-1 A[I, c]) of O(1) values,
        #  equal to rounding of those values — seed 1002604 differs by 4.7e-15 on an error of 2.35e-9)
        elif not np.allclose(eg, eo, rtol=1e-9, atol=1e-12 if rook else 1e-300):
            bad.append(f"errors {list(eg)} vs {list(eo)}")
        if g.termination() != o.termination():
            bad.append(f"termination {g.termination()} vs {o.termination()}")
        if list(g.link_dims()) != list(o.link_dims()):
            bad.append("link dims")
        max_rank_seen = max(max_rank_seen, max(list(g.link_dims()) or [0]))
        if not bad:
            pts = np.stack([rng.integers(0, dd, size=64) for dd in dims], axis=1)
            gv, ov = g.evaluate(pts), o.evaluate(pts)
            if not np.abs(gv - ov).max() <= 1e-9 * max(1.0, np.abs(ov).max()):
                bad.append(f"values differ by {np.abs(gv - ov).max():.3e}")
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {'; '.join(bad[:4])}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0} (Rook share {ROOK_SHARE}): {fails} failures; largest link dimension {max_rank_seen}; device {t_dev:.1f} s, oracle {t_orc:.1f} s, "
      f"total {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
