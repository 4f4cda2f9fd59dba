"""Stress of concurrent handle lifecycles on one GPU (pool.hip, XcdArbiter): ROUNDS rounds of eight config-5 patches grown from
scratch on eight host threads (create, optimize, fill twice -> graph capture, checksum, destroy); every checksum must equal the
one of the same patch run alone.  Usage: python tools/stress_concurrent_handles.py [rounds=10]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import t4a_amd
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
errs = []


def grow(p, out):
    try:
        tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
        tp.set_function(bench.patch_spec(p, 64))
        tp.add_global_pivots([[0] * bench.N_SITES])
        tp.set_max_sample_value(1.0)
        tp.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0,
                                        max_nglobal_pivot=0, seed=42), final_sweep1site=False)
        tp.fill_site_tensors()
        tp.fill_site_tensors()
        out[p] = float(tp.sum())
    except Exception as e:  # noqa: BLE001
        errs.append((p, repr(e)[:200]))


seq = {}
for p in range(8):
    grow(p, seq)
t0 = time.perf_counter()
bad_total = 0
for rnd in range(rounds):
    par = {}
    ths = [threading.Thread(target=grow, args=(p, par)) for p in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    bad = [p for p in range(8) if par.get(p) != seq.get(p)]
    bad_total += len(bad)
    if bad or errs:
        print("round", rnd, "mismatch", bad, "errors", errs[-2:], flush=True)
print(f"{rounds} rounds x 8 patches side by side: {bad_total} mismatches, {len(errs)} errors, "
      f"{(time.perf_counter() - t0) / (8 * rounds) * 1e3:.1f} ms per patch")
