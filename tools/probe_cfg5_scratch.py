"""One BASELINE configs[4] patch (30 active sites, chi = 128) grown from scratch, timed per optimize call; run under
`rocprofv3 --kernel-trace` and summarised by tools/trace_idle.py to see where a from-scratch patch spends its time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rep in range(n_rep):
    t0 = time.perf_counter()
    tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tp.set_function(bench.patch_spec(3 + rep, 64))
    tp.add_global_pivots([[0] * bench.N_SITES])
    tp.set_max_sample_value(1.0)
    t1 = time.perf_counter()
    tp.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0,
                                    seed=42), final_sweep1site=False)
    t2 = time.perf_counter()
    tp.fill_site_tensors()
    s = tp.sum()
    t3 = time.perf_counter()
    print(f"rep {rep}: setup {1e3*(t1-t0):.2f} ms, optimize(11) {1e3*(t2-t1):.2f} ms, fill+sum {1e3*(t3-t2):.2f} ms, chain {tp.chain_stats()}", flush=True)
