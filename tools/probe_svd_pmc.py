"""The workload of the LDS counter passes on jacobi_groups_kernel: thirty svd_backend calls each at 64 x 64, 48 x 32 and 96 x 96 (random
matrices, fixed seed).  Run under `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 tools/probe_svd_pmc.py` and
summarise with tools/svd_pmc_summary.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd  # noqa: E402

rng = np.random.default_rng(0)
for (m, n) in [(64, 64), (48, 32), (96, 96)]:
    a = rng.standard_normal((m, n))
    for _ in range(30):
        t4a_amd.svd_backend(a)
