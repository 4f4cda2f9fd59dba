import sys, os
sys.path.insert(0, "tensor4all-rs_amd/python"); sys.path.insert(0, "tests")
import numpy as np, t4a_amd as t4a, oracle_binding as ob
rng = np.random.default_rng(3)
trains = {
 "zero first core (1,1,4)": [np.zeros((1, 1, 4)), rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))],
 "zero middle core": [rng.uniform(-1, 1, (1, 3, 4)), np.zeros((4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))],
 "zero last core": [rng.uniform(-1, 1, (1, 3, 4)), rng.uniform(-1, 1, (4, 3, 4)), np.zeros((4, 2, 1))],
 "zero first core (1,3,4)": [np.zeros((1, 3, 4)), rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))],
 "all cores 1e-20": [rng.uniform(-1, 1, (1, 3, 4)) * 1e-20, rng.uniform(-1, 1, (4, 3, 4)) * 1e-20, rng.uniform(-1, 1, (4, 2, 1)) * 1e-20],
 "first core 1e-20": [rng.uniform(-1, 1, (1, 3, 4)) * 1e-20, rng.uniform(-1, 1, (4, 3, 4)), rng.uniform(-1, 1, (4, 2, 1))],
}
for name, cores in trains.items():
    for method in (0, 1, 2):
        o = ob.OracleTT(cores)
        try:
            o.compress(method=method, tolerance=1e-10); ores = f"links {o.link_dims()}"
        except Exception as e:
            ores = f"raised {e}"
        g = t4a.SimpleTensorTrain(cores)
        try:
            g.compress(method=method, tolerance=1e-10)
            f = g.full_tensor()
            gres = f"links {g.link_dims()} finite {bool(np.all(np.isfinite(f)))} max|v| {np.abs(f).max():.3e} (true max {np.abs(ob.OracleTT(cores).full_tensor()).max():.3e})"
        except t4a.T4aError as e:
            gres = f"raised: {str(e)[:70]}"
        print(f"{name:26s} method {method}: device {gres} | oracle {ores}")
