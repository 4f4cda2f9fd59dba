"""BASELINE config 5 on one GPU: 64 static patches (6 leading bits projected) of the d = 30 bench integrand,
per-patch max_bond_dim = 128; time of one full sweep per patch at saturated rank (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import t4a_amd
import bench
n_patches, chi = 64, 128
tot_sweep = tot_grow = 0.0
ranks = []
for p in range(n_patches):
    spec = bench.patch_spec(p, n_patches)
    t = t4a_amd.TensorCI2([2] * bench.N_SITES)
    t.set_function(spec)
    t.add_global_pivots([[0] * bench.N_SITES])
    t.set_max_sample_value(1.0)
    o = lambda it: t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=it, ncheck_history=10 ** 6, nsearch=0,
                                       max_nglobal_pivot=0, seed=42)
    t0 = time.perf_counter()
    t.optimize(o(9), final_sweep1site=False)
    t1 = time.perf_counter()
    t.optimize(o(2), final_sweep1site=False)
    t2 = time.perf_counter()
    tot_grow += t1 - t0
    tot_sweep += t2 - t1
    ranks.append(max(t.link_dims()))
print(f"64 patches chi=128: growth {tot_grow:.2f} s total, one full sweep per patch {tot_sweep*1e3/n_patches:.2f} ms avg "
      f"({tot_sweep:.2f} s for all 64), max link dims min/max {min(ranks)}/{max(ranks)}", flush=True)
