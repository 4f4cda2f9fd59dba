"""rrLU at cfg4-like sizes: parity against the oracle and time per call (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import t4a_amd
import oracle_binding as ob
rng = np.random.default_rng(2)
for (M, N, r) in [(1024, 1024, 512), (1370, 1376, 512), (1536, 1536, 512), (2048, 1000, 300), (3000, 500, 200)]:
    a = rng.uniform(-1, 1, size=(M, N))
    for left in (True, False):
        t4a_amd.rrlu(a, max_bond_dim=r, left_orthogonal=left)
        t0 = time.perf_counter()
        lu = t4a_amd.rrlu(a, max_bond_dim=r, left_orthogonal=left)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        f, rp, cp, npiv, err = ob.rrlu(a, max_bond_dim=r, left_orthogonal=left)
        dto = time.perf_counter() - t1
        same = (np.array_equal(lu.row_permutation, rp) and np.array_equal(lu.col_permutation, cp) and lu.npivots() == npiv
                and np.array_equal(lu.factored, f) and lu.error == err)
        print(f"M={M} N={N} r={r} left={left}: device {dt*1e3:.1f} ms ({dt*1e6/npiv:.1f} us/step), oracle {dto:.2f} s, bitwise equal: {same}", flush=True)
