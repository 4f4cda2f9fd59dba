"""Stress of the group chain (t4a_gpu_tci2_optimize_group: one chain of launches for eight handles, handle i's rrLU on XCD i):
ROUNDS rounds over all 64 config-5 patches in groups of eight (create, optimize_group, fill, checksum of every nested index
set and of the train's sum, destroy); every checksum must equal the one of the same patch optimised alone.  A race in the
group launch (argument blocks, mailboxes, ticket counters shared by mistake) would show up as a differing digest.
Usage: python tools/stress_group_chain.py [rounds=3] [patches=64]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import t4a_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_patches = int(sys.argv[2]) if len(sys.argv) > 2 else 64
OPT = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)


def make(p):
    tp = t4a_amd.TensorCI2([2] * bench.N_SITES)
    tp.set_function(bench.patch_spec(p, 64))
    tp.add_global_pivots([[0] * bench.N_SITES])
    tp.set_max_sample_value(1.0)
    return tp


def digest(tp):
    h = hashlib.sha256()
    for s in range(bench.N_SITES):
        h.update(tp.i_set(s).tobytes())
        h.update(tp.j_set(s).tobytes())
    tp.fill_site_tensors()
    h.update(repr(float(tp.sum())).encode())
    return h.hexdigest()


t0 = time.perf_counter()
ref = {}
for p in range(n_patches):
    tp = make(p)
    tp.optimize(OPT, final_sweep1site=False)
    ref[p] = digest(tp)
    del tp
t_solo = (time.perf_counter() - t0) / n_patches
bad_total, t_grp, grouped = 0, 0.0, 0
for rnd in range(rounds):
    for g0 in range(0, n_patches, 8):
        ps = list(range(g0, min(g0 + 8, n_patches)))
        t1 = time.perf_counter()
        tps = [make(p) for p in ps]
        t4a_amd.optimize_group(tps, OPT, final_sweep1site=False)
        got = {p: digest(tp) for p, tp in zip(ps, tps)}
        t_grp += time.perf_counter() - t1
        grouped += sum(1 for tp in tps if tp.chain_stats()["group_half_sweeps"] > 0)
        bad = [p for p in ps if got[p] != ref[p]]
        bad_total += len(bad)
        if bad:
            print("round", rnd, "mismatch in patches", bad, flush=True)
        del tps
print(f"{rounds} rounds x {n_patches} patches in groups of eight: {bad_total} mismatches ({grouped} handle runs used the group chain); "
      f"{t_grp / (rounds * n_patches) * 1e3:.2f} ms per patch grouped, {t_solo * 1e3:.1f} ms per patch alone (both incl. index-set read-back and fill)")
