"""gemm_kernel (f64 MFMA) at the shapes DESIGN.md quotes; run under `rocprofv3 --kernel-trace --stats` for device times:
1024^3, (256, 256, 512), (512, 256, 1400) and the batched core-core contraction (2 x 256 x 256, 58 batches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np
import t4a_amd
rng = np.random.default_rng(0)
for (m, k, n) in [(1024, 1024, 1024), (256, 512, 256), (512, 1400, 256), (2048, 2048, 2048)]:
    a = rng.uniform(-1, 1, size=(m, k)); b = rng.uniform(-1, 1, size=(k, n))
    for _ in range(3):
        c = t4a_amd.mat_mul(a, b)
    print(m, k, n, "max err", float(np.abs(c - a @ b).max()), flush=True)
