"""Single-XCD rrLU kernel against the chip-wide register kernel (GPU only): same inputs through the dense C ABI in two child
processes (T4A_RRLU_IMPL=reg forces the old kernel), digest of every output and time per pivot step.
Usage: python tools/probe_xcd.py [M N r]...   (T4A_RRLU_STAMPS=1 prints the per-phase stamps of rank 0)"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))


def child(M, N, maxb, left):
    import numpy as np
    import t4a_amd
    rng = np.random.default_rng(M * 1000 + N)
    a = rng.uniform(-1, 1, size=(M, N))
    kw = dict(max_bond_dim=maxb, left_orthogonal=bool(left))
    lu = t4a_amd.rrlu(a, **kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        lu = t4a_amd.rrlu(a, **kw)
        ts.append(time.perf_counter() - t0)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(lu.factored).tobytes())
    h.update(np.asarray(lu.row_permutation, dtype=np.int64).tobytes())
    h.update(np.asarray(lu.col_permutation, dtype=np.int64).tobytes())
    h.update(np.float64(lu.error).tobytes())
    best = min(ts)
    print(f"{os.environ.get('T4A_RRLU_IMPL', 'auto'):5s} M={M} N={N} left={left} steps={lu.npivots()} call_ms={best*1e3:.3f} "
          f"us_per_step={best*1e6/max(lu.npivots(),1):.3f} digest={h.hexdigest()[:16]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(*(int(x) for x in sys.argv[2:6]))
        sys.exit(0)
    args = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args) - 2, 3)] or [(685, 688, 256), (512, 512, 256), (768, 768, 256), (256, 256, 128), (100, 700, 100), (700, 100, 100), (130, 130, 130)]
    for (M, N, r) in shapes:
        for left in (1, 0):
            for impl in ("auto", "reg"):
                env = dict(os.environ)
                if impl != "auto":
                    env["T4A_RRLU_IMPL"] = impl
                subprocess.call(["timeout", "120", sys.executable, __file__, "child", str(M), str(N), str(r), str(left)], env=env)
