"""Eight BASELINE configs[4] patches (chi = 128) side by side on one GPU, one host thread and one handle (= one XCD) each, several
rounds in a row: the first round pays for the handles' first buffers (hipMalloc), later rounds find them in the process-wide cache."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402


def grow(p, out):
    tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tp.set_function(bench.patch_spec(p, 64))
    tp.add_global_pivots([[0] * bench.N_SITES])
    tp.set_max_sample_value(1.0)
    tp.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0,
                                    seed=42), final_sweep1site=False)
    tp.fill_site_tensors()
    out[p] = float(tp.sum())


seq = {}
for rnd in range(2):
    t0 = time.perf_counter()
    for p in range(2):
        grow(p, seq)
    print(f"sequential round {rnd}: {(time.perf_counter() - t0) / 2 * 1e3:.1f} ms per patch", flush=True)
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rnd in range(4):
    par = {}
    ths = [threading.Thread(target=grow, args=(p, par)) for p in range(nthreads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    print(f"{nthreads} threads round {rnd}: {dt * 1e3:.1f} ms wall, {dt / nthreads * 1e3:.1f} ms per patch, identical {all(par[p] == seq[p] for p in seq)}", flush=True)
