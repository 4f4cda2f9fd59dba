import os, sys, time
sys.path.insert(0, "tensor4all-rs_amd/python")
import t4a_amd
from t4a_amd.functions import quantics_osc2d
n, chi = 40, 512
for (k1,k2,k3,eps,k4,delta) in [(37,53,2111,0.5,16411,0.5),(37,53,20011,0.5,1048583,0.5),(3001,5003,200003,0.5,1048583,1.0)]:
    spec = quantics_osc2d(n, k1=k1,k2=k2,k3=k3,eps=eps,k4=k4,delta=delta)
    t = t4a_amd.TensorCI2([2]*n); t.set_function(spec); t.add_global_pivots([[0]*n]); t.set_max_sample_value(1.0)
    o = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=12, ncheck_history=10**6, nsearch=0, max_nglobal_pivot=0)
    t0=time.time(); t.optimize(o, final_sweep1site=False); dt=time.time()-t0
    t.profile_enable(True); t.profile_reset()
    t1=time.time(); t.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=2, ncheck_history=10**6, nsearch=0, max_nglobal_pivot=0), final_sweep1site=False); d2=time.time()-t1
    p=t.profile()
    print((k1,k2,k3,eps,k4,delta), "link", t.link_dims(), "hist", t.history()[0], "grow %.2fs"%dt, "full sweep %.3fs"%d2, "rrlu %.1f ms fill %.1f ms pi %.1f"%(p["rrlu_ms"],p["fill_ms"],p["pi_ms"]), flush=True)
