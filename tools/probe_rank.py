import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np
import t4a_amd
from t4a_amd.functions import quantics_osc2d
n, chi = 30, 256
for (k1,k2,k3,eps,k4,delta) in [(37,53,211,0.1,97,0.3),(37,53,211,0.1,1021,0.3),(37,53,211,0.1,4099,0.5),(37,53,2111,0.5,16411,0.5),(3001,5003,2111,0.5,16411,1.0)]:
    spec = quantics_osc2d(n, k1=k1,k2=k2,k3=k3,eps=eps,k4=k4,delta=delta)
    t = t4a_amd.TensorCI2([2]*n); t.set_function(spec); t.add_global_pivots([[0]*n]); t.set_max_sample_value(1.0)
    o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=12, ncheck_history=10**6, nsearch=0, max_nglobal_pivot=0)
    t0=time.time(); t.optimize(o, final_sweep1site=False); dt=time.time()-t0
    print((k1,k2,k3,eps,k4,delta), "link", t.link_dims(), "hist", t.history()[0], "err %.2e"%t.history()[1][-1], "%.2fs"%dt, flush=True)
