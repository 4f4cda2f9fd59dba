"""debug: cores after crossinterpolate2 (final 1-site sweep) under different switches, against the per-bond path"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)

def child(out):
    import t4a_amd, bench
    n = bench.N_SITES
    chi = int(os.environ.get("DBG_CHI", "24"))
    opt = t4a_amd.TCI2Options(tolerance=1e-9, max_bond_dim=chi, max_iter=4, nsearch=0, max_nglobal_pivot=0)
    t = t4a_amd.TensorCI2([2] * n)
    t.set_function(bench.patch_spec(0, 4))
    t.crossinterpolate2([[0] * n], opt)
    print(os.environ.get("DBG_NAME"), t.chain_stats(), flush=True)
    arrs = [t.site_tensor(s) for s in range(n)]
    arrs += [np.asarray(t.i_set(s), dtype=np.int64) for s in range(n)] + [np.asarray(t.j_set(s), dtype=np.int64) for s in range(n)]
    arrs += [np.asarray(t.pivot_errors()), np.asarray(t.bond_errors())]
    np.savez(out, *arrs)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    variants = {"ref": {"T4A_NO_CHAIN_1SITE": "1"}, "walk": {}, "nowalk": {"T4A_NO_WALK": "1"}}
    res = {}
    for name, env in variants.items():
        e = dict(os.environ); e.update(env); e["DBG_NAME"] = name
        out = f"/tmp/dbg_{name}.npz"
        subprocess.call([sys.executable, __file__, "child", out], env=e)
        res[name] = np.load(out)
    for name in variants:
        if name == "ref": continue
        bad = []
        for k in res["ref"].files:
            a, b = res["ref"][k], res[name][k]
            if a.shape != b.shape: bad.append((k, "shape", a.shape, b.shape)); continue
            if a.size == 0: continue
            d = np.abs(a.astype(float) - b.astype(float)).max()
            if d > 1e-9 * max(1.0, np.abs(a).max()): bad.append((k, float(d), a.shape))
        print(name, "mismatches:", bad[:8], flush=True)
