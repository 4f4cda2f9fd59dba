"""One case of tests/soak/soak_quantics.py in detail: device and oracle trains (identical pivots) against each other AND against the function
itself, and the conditioning of the fill's pivot blocks.  usage: probe_quantics_case.py SEED [SEED ...]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, t4a_amd as t4a, oracle_binding as ob
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    nv = int(rng.integers(1, 4))
    bits = [int(rng.integers(2, 10)) for _ in range(nv)]
    while sum(bits) > 20:
        bits[int(rng.integers(0, nv))] = 2
    lo = [float(rng.uniform(-1.0, 0.5)) for _ in range(nv)]
    hi = [lo[v] + float(rng.uniform(0.5, 3.0)) for v in range(nv)]
    w = rng.uniform(0.5, 4.0, nv); c = rng.uniform(-1.0, 1.0, nv); kind = int(rng.integers(0, 3))
    def f(x, w=w, c=c, kind=kind):
        x = np.asarray(x, dtype=np.float64)
        if kind == 0: return float(math.cos(float(w @ x)) + 0.3 * float(c @ x) + 1.0)
        if kind == 1: return float(1.0 / (1.5 + float(np.abs(w) @ (x * x))))
        return float(math.exp(-float(w @ ((x - c) ** 2))) + 0.1 * float(x[0]))
    scheme = int(rng.integers(0, 2)); endpoint = bool(rng.integers(0, 2))
    kw = dict(tolerance=float(10.0 ** rng.integers(-11, -5)), n_random_init_pivot=int(rng.integers(0, 5)), seed=int(rng.integers(0, 1000)), max_iter=int(rng.integers(3, 12)))
    g = t4a.quanticscrossinterpolate(bits, f, lo, hi, include_endpoint=endpoint, grid_unfolding=scheme, options=t4a.QtciOptions(**kw))
    o = ob.quanticscrossinterpolate(bits, f, lo, hi, include_endpoint=endpoint, grid_unfolding=scheme, options=ob.QtciOptions(**kw))
    pts = np.stack([rng.integers(0, 2 ** b, size=3000) for b in bits], axis=1)
    den = [(2 ** b - 1) if endpoint else 2 ** b for b in bits]
    exact = np.array([f([lo[v] + (hi[v] - lo[v]) * p[v] / den[v] for v in range(nv)]) for p in pts])
    gv, ov = g.evaluate(pts), o.evaluate(pts)
    sc = np.abs(exact).max()
    links = [c_.shape[2] for c_ in o.cores()[:-1]]
    print(f"seed {seed} bits {bits} tol {kw['tolerance']:g} unfolding {scheme}: max bond {max(links)}; last errors dev {g.history()[1][-1]:.2e} orc {o.history()[1][-1]:.2e}")
    print(f"   |device - f| {np.abs(gv - exact).max() / sc:.2e}   |oracle - f| {np.abs(ov - exact).max() / sc:.2e}   |device - oracle| {np.abs(gv - ov).max() / sc:.2e}")
    cd = max(np.abs(a - b).max() / max(1.0, np.abs(b).max()) for a, b in zip(g.tensor_train().site_tensors(), o.cores()))
    print(f"   largest core difference {cd:.2e}; largest |core entry| device {max(np.abs(a).max() for a in g.tensor_train().site_tensors()):.2e} oracle {max(np.abs(b).max() for b in o.cores()):.2e}")
