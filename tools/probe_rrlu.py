"""Ad-hoc timing probe of the persistent rrLU kernel for different (W, T) plans (GPU only)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))


def child(M, N, maxb):
    import numpy as np
    import t4a_amd
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, size=(M, N))
    t4a_amd.rrlu(a, max_bond_dim=maxb)  # warm-up
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        lu = t4a_amd.rrlu(a, max_bond_dim=maxb)
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    print(f"M={M} N={N} steps={lu.npivots()} W={os.environ.get('T4A_RRLU_W','auto')} T={os.environ.get('T4A_RRLU_T','auto')}"
          f" best_call_ms={best*1e3:.3f} us_per_step={best*1e6/max(lu.npivots(),1):.3f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    shapes = [(512, 512, 256), (768, 768, 256)]
    plans = [("auto", "auto"), ("32", "256"), ("64", "256"), ("128", "256"), ("192", "256"), ("64", "512"),
             ("128", "512"), ("128", "128"), ("240", "128"), ("64", "1024")]
    for (M, N, r) in shapes:
        for (W, T) in plans:
            env = dict(os.environ)
            if W != "auto":
                env["T4A_RRLU_W"] = W
                env["T4A_RRLU_T"] = T
            subprocess.call(["timeout", "120", sys.executable, __file__, "child", str(M), str(N), str(r)], env=env)
