"""BASELINE config 5 on ONE GPU with patches run side by side: every TensorCI2 handle is assigned one of the eight XCDs, so
T host threads drive T independent patch interpolations concurrently (GPU only).
Usage: python tools/probe_cfg5_concurrent.py [threads=8] [patches=16]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import t4a_amd
import bench
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_run = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_patches, chi = 64, 128


def run_patch(p, out):
    t = t4a_amd.TensorCI2([2] * bench.N_SITES)
    t.set_function(bench.patch_spec(p, n_patches))
    t.add_global_pivots([[0] * bench.N_SITES])
    t.set_max_sample_value(1.0)
    o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
    t.optimize(o, final_sweep1site=False)
    t.fill_site_tensors()
    out[p] = (max(t.link_dims()), float(t.sum()))


for T in (1, n_threads):
    res = {}
    todo = list(range(n_run))
    lock = threading.Lock()

    def worker():
        while True:
            with lock:
                if not todo:
                    return
                p = todo.pop(0)
            run_patch(p, res)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker) for _ in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    print(f"{T} thread(s): {n_run} patches (chi={chi}, 11 iterations + fill) in {dt*1e3:.1f} ms = {dt*1e3/n_run:.1f} ms per patch; "
          f"checksum {sum(v[1] for v in res.values()):.12e} ranks {sorted(set(v[0] for v in res.values()))}", flush=True)
