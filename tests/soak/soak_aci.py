"""Soak of the ACI path (crates/tensor4all-aci: element-wise product / sum of tensor trains by alternating cross interpolation) against
oracle/t4a_oracle_aci.hpp: the body of tests/test_gpu_aci.py::test_random_shapes_match_the_oracle over many seeds and a wider range — 2 - 9
sites of dimension 2 - 4, one to three inputs with bond dimensions 1 - 6, a random initial guess of bond dimension 1 - 8, random tolerance
and iteration cap, the global guard on / off.  WITHOUT the guard: rank, error and termination histories IDENTICAL (the errors bitwise: the
candidate matrices are accumulated in the same order on both sides), the solution at every grid point 1e-10 of the largest value.  WITH the
guard the same is expected but not guaranteed — its walk takes an arg-max over |exact - solution|, which is rounding noise wherever the
solution is already exact, and the two solutions agree only to rounding: such runs are counted, and must both be good.
usage: python3 tests/soak/soak_aci.py N [seed0]     (test infrastructure: the oracle is the checker; not collected by pytest)"""
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
from test_oracle_aci import dense, lcg_tt  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
converged = 0
counts = {}
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(2, 10))
    site_dims = [int(x) for x in rng.integers(2, 5, size=n)]
    while int(np.prod(site_dims)) > 40000:  # (the whole grid is compared)
        site_dims[int(rng.integers(0, n))] = 2

    def capped(xs):
        return [int(min(x, np.prod(site_dims[:b + 1]), np.prod(site_dims[b + 1:]))) for b, x in enumerate(xs)]
    K = int(rng.integers(1, 4))
    ins = [lcg_tt(site_dims, capped(rng.integers(1, 7, size=n - 1)), 1000 * case + k + 1) for k in range(K)]
    guess = [rng.standard_normal(c.shape) for c in lcg_tt(site_dims, capped(rng.integers(1, 9, size=n - 1)), 1)]
    product = bool(rng.integers(0, 2))
    kw = dict(initial_guess=guess, tolerance=float(10.0 ** rng.integers(-13, -5)), enable_global_guard=bool(rng.random() < 0.4),
              max_iters=int(rng.integers(2, 16)))
    ctx = f"seed {seed0 + case} dims {site_dims} inputs {[[c.shape[2] for c in t[:-1]] for t in ins]} guess {[c.shape[2] for c in guess[:-1]]} " \
          f"{'product' if product else 'sum'} tol {kw['tolerance']:g} guard {kw['enable_global_guard']} max_iters {kw['max_iters']}"
    try:
        rd = t4a.elementwise_batched(t4a.ACI_PRODUCT if product else t4a.ACI_SUM, ins, t4a.AciOptions(**kw))
        ro = ob.aci_elementwise(ob.ACI_PRODUCT if product else ob.ACI_SUM, ins, ob.AciOptions(**kw))
        bad = []
        guard = kw["enable_global_guard"]
        dn = [dense(t) for t in ins]
        exact = (np.prod(dn, axis=0) if product else np.sum(dn, axis=0)).ravel()
        grid = np.array(list(itertools.product(*[range(d) for d in site_dims])), dtype=np.uint32)
        vd = np.asarray(rd.tensor_train.evaluate(grid))
        vo = np.asarray(ro.tensor_train.evaluate(grid))
        scale = max(np.abs(exact).max(), 1e-300)
        same_hist = (list(rd.ranks) == list(ro.ranks) and list(rd.nglobal_pivots) == list(ro.nglobal_pivots) and rd.termination == ro.termination
                     and np.array_equal(rd.errors, ro.errors))
        if not guard:
            # without the guard nothing discrete depends on tolerance-level values: histories identical, errors BITWISE
            if not same_hist:
                bad.append(f"histories differ without the guard: ranks {list(rd.ranks)} vs {list(ro.ranks)}, termination {rd.termination} vs "
                           f"{ro.termination}, errors {list(rd.errors)} vs {list(ro.errors)}")
            elif not np.abs(vd - vo).max() <= 1e-10 * scale:
                bad.append(f"solution differs from the oracle's by {np.abs(vd - vo).max() / scale:.2e}")
        elif same_hist:
            counts["guard_identical"] = counts.get("guard_identical", 0) + 1
            if not np.abs(vd - vo).max() <= 1e-10 * scale:
                bad.append(f"solution differs from the oracle's by {np.abs(vd - vo).max() / scale:.2e}")
        else:
            # The guard's floating-zone walk (global_guard.rs:49-181, floating_zone.rs:46-103) takes an arg-max over |op(inputs) - solution|:
            # where the solution is exact to rounding that is an arg-max over NOISE, and the device's solution cores equal the oracle's only
            # to rounding (solves and GEMMs) — the walks may part and find different pivots.  Both runs must then still be good ones: the
            # same exact result within the larger of the two runs' own accuracy and the tolerance.
            counts["guard_histories_differ"] = counts.get("guard_histories_differ", 0) + 1
            noise = bool(max(np.abs(rd.errors).max(initial=0.0), np.abs(ro.errors).max(initial=0.0)) <= 1e-12 * max(scale, 1.0))
            counts["...of which all errors at rounding level"] = counts.get("...of which all errors at rounding level", 0) + int(noise)
            ed, eo = np.abs(vd - exact).max() / scale, np.abs(vo - exact).max() / scale
            if rd.termination == 0 and ro.termination == 0 and not ed <= max(10.0 * eo, 1e3 * kw["tolerance"], 1e-10):
                bad.append(f"guard histories differ and the device's solution is {ed:.2e} from the exact result (the oracle's {eo:.2e})")
        if rd.termination == 0:
            converged += 1
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {'; '.join(bad[:3])}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; {converged} converged runs; {counts}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
