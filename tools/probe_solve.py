import sys, time
sys.path.insert(0,"tensor4all-rs_amd/python"); sys.path.insert(0,"tests")
import numpy as np, t4a_amd, oracle_binding as ob
rng=np.random.default_rng(0)
for (n,nrhs) in [(1,1),(2,3),(31,5),(32,32),(33,70),(100,40),(255,100),(256,512),(300,64),(512,100),(700,33),(1024,16),(1100,8)]:
    a=rng.standard_normal((n,n)); b=rng.standard_normal((n,nrhs))
    t0=time.perf_counter(); x=t4a_amd.solve_matrix(a,b); dt=time.perf_counter()-t0
    xo=ob.solve(a,b)
    print(n,nrhs,"bitwise equal to oracle:",np.array_equal(x,xo),"max diff %.2e"%np.abs(x-xo).max(),"resid %.2e"%np.abs(a@x-b).max(), "%.1f ms"%(dt*1e3), flush=True)
# singular / zero cases
try:
    t4a_amd.solve_matrix(np.zeros((3,3)), np.ones((3,1)))
except t4a_amd.T4aError as e: print("zero matrix ->", e.code)
s=np.array([[1.,2.],[2.,4.]])
try:
    t4a_amd.solve_matrix(s, np.ones((2,1)))
except t4a_amd.T4aError as e: print("singular ->", e.code)
