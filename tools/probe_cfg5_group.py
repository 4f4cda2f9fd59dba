"""Eight BASELINE configs[4] patches (chi = 128) on one GPU: one after the other, eight host threads, and t4a_gpu_tci2_optimize_group
(one thread, eight handles in lock-step, one XCD each)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402

OPT = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)


def make(p):
    tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tp.set_function(bench.patch_spec(p, 64))
    tp.add_global_pivots([[0] * bench.N_SITES])
    tp.set_max_sample_value(1.0)
    return tp


def finish(tp):
    tp.fill_site_tensors()
    return float(tp.sum())


n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ref = {}
for rnd in range(2):
    t0 = time.perf_counter()
    for p in range(n):
        tp = make(p)
        tp.optimize(OPT, final_sweep1site=False)
        ref[p] = finish(tp)
    print(f"one at a time, round {rnd}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per patch", flush=True)
for rnd in range(3):
    t0 = time.perf_counter()
    tps = [make(p) for p in range(n)]
    t4a_amd.optimize_group(tps, OPT, final_sweep1site=False)
    t4a_amd.fill_site_tensors_group(tps)  # (all fills issued, then completed)
    got = {p: float(tps[p].sum()) for p in range(n)}
    dt = time.perf_counter() - t0
    del tps
    print(f"optimize_group of {n}, round {rnd}: {dt * 1e3:.1f} ms wall, {dt / n * 1e3:.2f} ms per patch, speed-up {sum(1 for _ in ref) and 0 or 0}"
          f" identical {all(got[p] == ref[p] for p in ref)}", flush=True)
