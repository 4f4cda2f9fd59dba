"""One-wave and one-workgroup rrLU kernels against what would run without them (T4A_NO_W1=1, T4A_NO_WG=1): time per pivot step as
the SLOPE between a factorisation capped at r and one capped at r / 2 steps (the fixed cost of a call drops out), digests of all.
Usage: python tools/probe_wg.py [M N r]...   (T4A_WG_MAXV=96 lifts the plan limit of the one-workgroup kernel)"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))


def child(M, N, r):
    import numpy as np
    import t4a_amd
    rng = np.random.default_rng(M * 1000 + N)
    a = rng.uniform(-1, 1, size=(M, N))
    out = []
    for cap in (r, max(r // 2, 1)):
        kw = dict(max_bond_dim=cap, left_orthogonal=True)
        lu = t4a_amd.rrlu(a, **kw)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            lu = t4a_amd.rrlu(a, **kw)
            ts.append(time.perf_counter() - t0)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(lu.factored).tobytes())
        h.update(np.asarray(lu.row_permutation, dtype=np.int64).tobytes())
        h.update(np.asarray(lu.col_permutation, dtype=np.int64).tobytes())
        out.append((min(ts), lu.npivots(), h.hexdigest()[:12]))
    (t1, n1, d1), (t2, n2, d2) = out
    slope = (t1 - t2) / max(n1 - n2, 1) * 1e6
    print(f"{'old ' if os.environ.get('T4A_NO_WG') else 'wg  ' if os.environ.get('T4A_NO_W1') else 'w1  '} M={M} N={N} steps={n1}/{n2} call_us={t1*1e6:.1f}/{t2*1e6:.1f} us_per_step(slope)={slope:.3f} digest={d1}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(*(int(x) for x in sys.argv[2:5]))
        sys.exit(0)
    args = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args) - 2, 3)] or [(8, 8, 8), (16, 16, 16), (32, 32, 32), (64, 32, 32), (32, 64, 32), (64, 64, 64), (32, 128, 32), (64, 128, 64), (128, 128, 128), (64, 256, 64), (128, 256, 128), (128, 384, 128), (64, 512, 64)]
    for (M, N, r) in shapes:
        for extra in ({}, {"T4A_NO_W1": "1"}, {"T4A_NO_W1": "1", "T4A_NO_WG": "1"}):
            if extra == {} and (M > 64 or N > 64):
                continue
            env = dict(os.environ)
            env.update(extra)
            subprocess.call(["timeout", "120", sys.executable, __file__, "child", str(M), str(N), str(r)], env=env)
