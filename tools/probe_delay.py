"""Kernel time of one rrLU shape for a list of poll delays (child processes: the delay is read once per process)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import t4a_amd
    M, N, r = (int(x) for x in sys.argv[2:5])
    spec = t4a_amd.quantics_osc2d(30)
    a = np.random.default_rng(1).uniform(-1, 1, size=(M, N))
    t4a_amd.rrlu(a, max_bond_dim=r)
    # time through a device-resident loop is not available from the dense ABI: use the stamps total instead
    sys.exit(0)
shapes = [(200, 200, 100), (340, 344, 128), (685, 688, 256)]
for (M, N, r) in shapes:
    for d in (4, 6, 8, 10, 12, 14, 16):
        env = dict(os.environ, T4A_RRLU_POLLDELAY=str(d), T4A_RRLU_STAMPS="1")
        out = subprocess.run([sys.executable, __file__, "child", str(M), str(N), str(r)], env=env, capture_output=True, text=True)
        line = [l for l in out.stderr.splitlines() if "stamps" in l][-1]
        parts = dict(p.split("=") for p in line.split("|")[1].split() if "=" in p and p.split("=")[1].isdigit())
        tot = sum(int(parts[k]) for k in ("s0", "s1", "s2", "s3", "s4", "s7"))
        w = line.split("W=")[1].split()[0]
        print(f"M={M} N={N} r={r} W={w} delay={d}: {tot / r:.0f} cycles/step", flush=True)
