"""Checks that every patch bench.py hands to a rank at --gpus 2/4/8 saturates chi (untimed growth phase of bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import t4a_amd
import bench
for world in (2, 4, 8):
    for rank in range(world):
        spec = bench.patch_spec(rank, world)
        t = t4a_amd.TensorCI2([2] * bench.N_SITES)
        t.set_function(spec)
        t.add_global_pivots([[0] * bench.N_SITES])
        t.set_max_sample_value(1.0)
        o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=bench.CHI, max_iter=10, ncheck_history=10 ** 6, nsearch=0,
                                max_nglobal_pivot=0, seed=42)
        t0 = time.time()
        t.optimize(o, final_sweep1site=False)
        ld = t.link_dims()
        print(world, rank, "max link", max(ld), "saturated" if max(ld) == bench.CHI else "NOT SATURATED", "%.2fs" % (time.time() - t0), flush=True)
