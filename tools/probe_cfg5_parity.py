import sys, os
sys.path.insert(0, "tensor4all-rs_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, t4a_amd, bench, oracle_binding as ob
n, chi, n_patches = bench.N_SITES, 128, 64
opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=9, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
pts = np.random.default_rng(5).integers(0, 2, size=(200, n))
for p in [0, 27, 63]:
    t = t4a_amd.TensorCI2([2] * n); t.set_function(bench.patch_spec(p, n_patches)); t.add_global_pivots([[0] * n]); t.set_max_sample_value(1.0)
    t.optimize(opt, final_sweep1site=False); t.fill_site_tensors()
    o = ob.OracleTCI2([2] * n); o.set_function(bench.patch_spec(p, n_patches)); o.add_global_pivots([[0] * n]); o.set_max_sample_value(1.0)
    o.optimize(opt, final_sweep1site=False); o.fill_site_tensors()
    cd = max(np.abs(t.site_tensor(s) - o.site_tensor(s)).max() / max(1.0, np.abs(o.site_tensor(s)).max()) for s in range(n))
    ev = np.abs(t.evaluate(pts) - o.evaluate(pts)).max()
    print(p, "max rel core diff", cd, "max eval diff", ev, "max|tt|", np.abs(o.evaluate(pts)).max(), flush=True)
