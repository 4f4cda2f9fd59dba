"""Soak of t4a_gpu_tci2_optimize_group (up to eight TensorCI2 handles in lock-step from one thread, one XCD each: the per-GPU form of the
patch farm, BASELINE.json configs[4]) against the SAME handles solved one at a time with optimize(): device against device, everything
BITWISE — index sets, rank and error histories, termination, max_sample_value, every site tensor.  Random groups of 2 - 8 members that share
the number of sites (so their half-sweeps run as one chain of launches) with different functions (the oscillatory 2-variable quantics
integrand with random wave numbers, cos(10 x) exp(-x), members that converge early and drop out), random options (tolerance, rank cap 4 - 95,
iterations, sweep strategy, strictly nested sets, normalisation, history length), with and without the final 1-site sweep, 1 - 3 initial pivots.
usage: python3 tools/soak_group.py N [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
members_total = 0
grouped_half_sweeps = 0
max_rank = 0
t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    n = 2 * int(rng.integers(4, 13))
    size = int(rng.integers(2, 9))
    specs = []
    for p in range(size):
        if rng.random() < 0.2:
            specs.append(t4a.quantics_trig_exp(n))
        else:
            specs.append(t4a.quantics_osc2d(n, k1=int(rng.integers(1, 40)), k2=int(rng.integers(1, 40)), k3=int(rng.integers(1, 2000)),
                                            eps=float(rng.choice([0.1, 0.5])), k4=int(rng.integers(1, 9000)), delta=float(rng.choice([0.0, 0.5]))))
    opts = t4a.TCI2Options(tolerance=float(10.0 ** rng.integers(-12, -5)), max_bond_dim=int(rng.integers(4, 96)), max_iter=int(rng.integers(2, 9)),
                           nsearch=0, max_nglobal_pivot=0, normalize_error=bool(rng.integers(0, 2)), sweep_strategy=int(rng.integers(0, 3)),
                           strictly_nested=bool(rng.random() < 0.3), ncheck_history=int(rng.integers(1, 4)), seed=int(rng.integers(0, 100)))
    final = bool(rng.integers(0, 2))
    pivots = [[0] * n] + [[int(rng.integers(0, 2)) for _ in range(n)] for _ in range(int(rng.integers(0, 3)))]
    ctx = f"seed {seed0 + case} n {n} members {size} final {final} opts {vars(opts)}"
    try:
        solo, grouped = [], []
        for spec in specs:
            for dst in (solo, grouped):
                t = t4a.TensorCI2([2] * n)
                t.set_function(spec)
                t.add_global_pivots(pivots)
                dst.append(t)
        for t in solo:
            t.optimize(opts, final_sweep1site=final)
        t4a.optimize_group(grouped, opts, final_sweep1site=final)
        bad = []
        for k, (a, b) in enumerate(zip(solo, grouped)):
            for p in range(n):
                if not (np.array_equal(a.i_set(p), b.i_set(p)) and np.array_equal(a.j_set(p), b.j_set(p))):
                    bad.append(f"member {k}: index sets at site {p}")
                    break
            if bad:
                break
            if a.history()[0] != b.history()[0] or not np.array_equal(a.history()[1], b.history()[1]):
                bad.append(f"member {k}: histories {a.history()} vs {b.history()}")
                break
            if a.termination() != b.termination() or a.max_sample_value() != b.max_sample_value():
                bad.append(f"member {k}: termination / max_sample_value")
                break
            if final:
                for p in range(n):
                    if not np.array_equal(a.site_tensor(p), b.site_tensor(p)):
                        bad.append(f"member {k}: site tensor {p}")
                        break
            if bad:
                break
            grouped_half_sweeps += b.chain_stats()["group_half_sweeps"]
            max_rank = max(max_rank, max(a.link_dims()))
            members_total += 1
        if bad:
            fails += 1
            print(f"FAIL {ctx}: {bad[0][:300]}", flush=True)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL {ctx}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} groups from seed {seed0}: {fails} failures; {members_total} members compared, {grouped_half_sweeps} half-sweeps run as a group, largest link "
      f"dimension {max_rank}; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
