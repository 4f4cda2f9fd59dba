"""BASELINE config 2 (d = 20, cos(10x) exp(-x), tol 1e-8, chi <= 64): time to solution, device vs CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import t4a_amd
import oracle_binding as ob
from t4a_amd.functions import quantics_trig_exp, quantics_osc2d
for name, spec, n, chi in (("cfg2 cos(10x)exp(-x)", quantics_trig_exp(20), 20, 64),
                           ("osc2d d=20 chi=64", quantics_osc2d(20, k1=37, k2=53, k3=211, eps=0.1, k4=97, delta=0.3), 20, 64),
                           ("osc2d d=20 chi=16", quantics_osc2d(20, k1=37, k2=53, k3=211, eps=0.1, k4=97, delta=0.3), 20, 16),
                           ("osc2d d=20 chi=8", quantics_osc2d(20, k1=37, k2=53, k3=211, eps=0.1, k4=97, delta=0.3), 20, 8),
                           ("trig d=20 chi=6 (3 terms)", quantics_trig_exp(20, a=33.0, b=0.7, cc=0.4, cs=0.9), 20, 6),
                           ("osc2d d=30 chi=32", quantics_osc2d(30, k1=37, k2=53, k3=2111, eps=0.5, k4=16411, delta=0.5), 30, 32),
                           ("trig d=30 2 terms chi=8", quantics_trig_exp(30, a=33.0, b=0.7, cc=0.4, cs=0.9), 30, 8)):
    opt = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=chi, max_iter=20, nsearch=0, max_nglobal_pivot=0)
    tg = float("inf")
    for rep in range(6):
        g = t4a_amd.TensorCI2([2] * n)
        g.set_function(spec)
        t0 = time.perf_counter()
        g.crossinterpolate2([[0] * n], opt)
        tg = min(tg, time.perf_counter() - t0)
    tw = float("inf")
    for rep in range(4):
        w = t4a_amd.TensorCI2([2] * n)
        w.set_function(spec)
        w.set_chain(True, small_engine=False)
        t0 = time.perf_counter()
        w.crossinterpolate2([[0] * n], opt)
        tw = min(tw, time.perf_counter() - t0)
    st_ = g.small_stats()
    print(f"{name}: WITH engine (default: 16 x 16) {tg*1e3:.3f} ms [{st_['iterations']} iterations in the launch, completed {st_['completed']}, device us {st_['device_us']}] | WITHOUT {tw*1e3:.3f} ms", flush=True)
    t16 = float("inf")
    for rep in range(4):
        w = t4a_amd.TensorCI2([2] * n)
        w.set_function(spec)
        w.set_chain(True, small_tile32=True)
        t0 = time.perf_counter()
        w.crossinterpolate2([[0] * n], opt)
        t16 = min(t16, time.perf_counter() - t0)
    print(f"{name}: engine with the 32 x 32 tile allowed {t16*1e3:.3f} ms {w.small_stats()['iterations']} iterations in the launch, device us {w.small_stats()['device_us']}", flush=True)
    d = t4a_amd.TensorCI2([2] * n)
    d.set_function(spec)
    d.set_chain(True, small_stamps=True)
    d.crossinterpolate2([[0] * n], opt)
    print(f"{name}: stamped launch {d.small_stats()}", flush=True)
    o = ob.OracleTCI2([2] * n)
    o.set_function(spec)
    t0 = time.perf_counter()
    o.crossinterpolate2([[0] * n], opt)
    to = time.perf_counter() - t0
    print(f"{name}: device {tg*1e3:.3f} ms, oracle {to*1e3:.3f} ms, iterations {len(g.history()[0])}, rank {g.rank()}, "
          f"same link dims {g.link_dims() == o.link_dims()}", flush=True)
