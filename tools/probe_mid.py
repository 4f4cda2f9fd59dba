"""Times the mid-chain bond shape of the bench (685 x 688, 256 pivot steps) through the dense C ABI and prints the
per-phase stamps when T4A_RRLU_STAMPS=1 (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np
import t4a_amd
rng = np.random.default_rng(1)
M, N, r = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (685, 688, 256)))
a = rng.uniform(-1, 1, size=(M, N))
for _ in range(3):
    lu = t4a_amd.rrlu(a, max_bond_dim=r)
print("npivots", lu.npivots(), "checksum", float(np.abs(lu.factored).sum()), lu.row_permutation[:5], flush=True)
