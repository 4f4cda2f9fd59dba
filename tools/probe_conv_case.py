"""One case of tests/soak/soak_tt_ops.py in detail: the conversion's cores on the device against the oracle's.  usage: probe_conv_case.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "soak"))
import numpy as np, t4a_amd as t4a, oracle_binding as ob
seed = int(sys.argv[1])
src = open(os.path.join(ROOT, "tests", "soak", "soak_tt_ops.py")).read()
ns = {}
exec(src[src.index("def make_train"):src.index("t0 = time.perf_counter()")], {"np": np}, ns)
rng = np.random.default_rng(seed)
dims, cores, kind = ns["make_train"](rng)
print("dims", dims, "links", [c.shape[2] for c in cores[:-1]], "kind", kind)
kw = dict(tolerance=1e-08, max_bond_dim=14, max_iter=5)
g = t4a.TensorCI2.from_tensor_train(t4a.SimpleTensorTrain(cores), **kw)
res = ob.OracleTT(cores).to_tci2(**kw)
full = ob.OracleTT(cores).full_tensor()
print("max |tensor|", np.abs(full).max(), "pivot errors dev", g.pivot_errors(), "oracle", res["pivot_errors"])
for p in range(len(dims)):
    a, b = g.site_tensor(p), res["cores"][p]
    print(p, a.shape, "max|a| %.3e max|b| %.3e max|a-b| %.3e" % (np.abs(a).max(), np.abs(b).max(), np.abs(a - b).max()))
gt = g.to_tensor_train().full_tensor()
print("device TT vs tensor: %.3e ; oracle cores TT vs tensor: %.3e" % (np.abs(gt - full).max(), np.abs(ob.OracleTT(res["cores"]).full_tensor() - full).max()))
