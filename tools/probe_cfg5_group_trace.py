"""Only the optimize_group part of probe_cfg5_group.py (for `rocprofv3 --kernel-trace`): `rounds` times eight configs[4] patches from scratch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402

OPT = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for rnd in range(rounds):
    t0 = time.perf_counter()
    tps = []
    for p in range(n):
        tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
        tp.set_function(bench.patch_spec(p, 64))
        tp.add_global_pivots([[0] * bench.N_SITES])
        tp.set_max_sample_value(1.0)
        tps.append(tp)
    t4a_amd.optimize_group(tps, OPT, final_sweep1site=False)
    t1 = time.perf_counter()
    for tp in tps:
        tp.fill_site_tensors()
    s = [float(tp.sum()) for tp in tps]
    dt = time.perf_counter() - t0
    print(f"round {rnd}: optimize_group {1e3 * (t1 - t0):.1f} ms, with fill + sum {dt * 1e3:.1f} ms ({dt / n * 1e3:.2f} ms per patch)", flush=True)
    del tps
