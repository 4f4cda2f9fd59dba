"""Five thin SVDs (or QRs: third argument qr) of one Gaussian 512 x 256 matrix, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/probe_svd_once.py [m n [qr]]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np
import t4a_amd
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
a = np.random.default_rng(0).standard_normal((m, n))
f = t4a_amd.qr_backend if len(sys.argv) > 3 and sys.argv[3] == 'qr' else t4a_amd.svd_backend
f(a)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    f(a)
    ts.append(time.perf_counter() - t0)
print(f"{f.__name__} {m} x {n}: {min(ts) * 1e3:.3f} ms", flush=True)
