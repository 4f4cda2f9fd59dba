"""One case of tests/test_gpu_small.py::test_fuzz_engine_general_path_and_oracle_agree, spelled out: who differs from whom.  usage: probe_fuzz_case.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import t4a_amd as t4a
import oracle_binding as ob
import test_gpu_small as T
seed = int(sys.argv[1])
spec, dims, opts, pivots, final = T._fuzz_case(t4a, seed)
n = len(dims)
s, g, o = T.three(t4a, spec, dims)
o.set_pivot_search(opts["pivot_search"])
for h in (s, g, o):
    h.add_global_pivots(pivots)
    h.set_max_sample_value(1.0)
    h.optimize(t4a.TCI2Options(**opts), final_sweep1site=final)
print("opts", opts, "dims", dims, "final", final)
print("small stats", s.small_stats())
print("ranks s/g/o", s.history()[0], g.history()[0], o.history()[0])
print("errors s", s.history()[1], "\nerrors g", g.history()[1], "\nerrors o", o.history()[1])
for p in range(n):
    a, b, c = s.j_set(p), g.j_set(p), o.j_set(p)
    ai, bi, ci = s.i_set(p), g.i_set(p), o.i_set(p)
    print(p, "J s==g", np.array_equal(a, b), "s==o", np.array_equal(a, c), "g==o", np.array_equal(b, c), "| I s==g", np.array_equal(ai, bi), "s==o", np.array_equal(ai, ci), "g==o", np.array_equal(bi, ci))
