"""Sweep (W, T, CPT, NCOPY) plans of the reg rrLU kernel at the cfg3 mid-chain shape (GPU only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, R = 685, 688, 256
cfgs = []
for T in ("384", "704"):
    for CPT in ("4", "2"):
        for PD in ("0", "8", "16", "24", "32"):
            cfgs.append({"T4A_RRLU_T": T, "T4A_RRLU_CPT": CPT, "T4A_RRLU_POLLDELAY": PD})
for c in cfgs:
    env = dict(os.environ); env.update(c); env["T4A_RRLU_STAMPS"] = "1"
    p = subprocess.run(["timeout", "120", sys.executable, os.path.join(ROOT, "tools", "probe_rrlu.py"), "child", str(M), str(N), str(R)],
                       env=env, capture_output=True, text=True)
    lines = [l for l in (p.stdout + p.stderr).splitlines() if "stamps" in l or "M=" in l]
    st = [l for l in lines if "stamps" in l][-1:] 
    tm = [l for l in lines if l.startswith("M=")][-1:]
    print(c, "|", (st[0].split("|")[1].split("(")[0].strip() if st else "?"), "|", (st[0].split("|")[0].split("]")[1].strip() if st else ""), "|", tm[0].split("best_call_ms=")[1] if tm else "FAIL", flush=True)
