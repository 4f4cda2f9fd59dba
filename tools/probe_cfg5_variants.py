"""Where a BASELINE configs[4] patch grown from scratch spends its rrLU time: per kernel instantiation (profile codes decoded by
bench.rrlu_kernel_name), launches, pivot steps, ms and us per step; and the ranks per iteration."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402

for rep in range(2):
    tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tp.set_function(bench.patch_spec(3 + rep, 64))
    tp.add_global_pivots([[0] * bench.N_SITES])
    tp.set_max_sample_value(1.0)
    tp.profile_enable(True)
    tp.profile_reset()
    t1 = time.perf_counter()
    tp.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0,
                                    seed=42), final_sweep1site=False)
    t2 = time.perf_counter()
    vs = tp.profile_variants()
    tp.profile_enable(False)
    if rep == 0:
        continue
    print(f"optimize(11) {1e3 * (t2 - t1):.2f} ms, ranks per iteration {tp.history()[0] if hasattr(tp, 'history') else ''}")
    tot = 0.0
    for v in sorted(vs, key=lambda v: -v["ms"]):
        if v["code"] >= 10000000:
            continue
        tot += v["ms"]
        print(f"{bench.rrlu_kernel_name(v['code']):50s} launches {v['launches']:5.0f} steps {v['steps']:7.0f} ms {v['ms']:7.3f} us/step {1e3 * v['ms'] / max(v['steps'], 1):.2f}")
    print(f"all rrLU kernels {tot:.2f} ms")
