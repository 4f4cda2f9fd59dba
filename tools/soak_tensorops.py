"""Soak of the labelled-tensor operations (tensor4all-core/src/defaults: contract.rs:283-444 `contract`, svd.rs:255-338 `svd_with`,
qr.rs:206-328 `qr_with`) against numpy: the bodies of tests/test_gpu_tensor.py::test_random_contractions_and_factorisations over many seeds
and a wider range — operands of rank 1 - 5 with dimensions 1 - 6 (dimension 1 and fully contracted / outer products included), 0 .. all
common labels in a random order; factorisations of rank 2 - 5 tensors over a random bipartition, at unit scale and at 1e+-120 / 1e+-200 (the
scales that exposed the SVD / QR overflow and underflow, profiles/r06_svd_small.txt).
usage: python3 tools/soak_tensorops.py N [seed0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import numpy as np  # noqa: E402
import t4a_amd as t4a  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
letters = "abcdefghijklmnopqrstuvwxyz"
scales = [1.0, 1.0, 1.0, 1e120, 1e-120, 1e200, 1e-200]
fails = 0
worst = {}


def note(what, err, limit, info):
    global fails
    worst[what] = max(worst.get(what, 0.0), err)
    if not (err <= limit):
        fails += 1
        print(f"FAIL {what} {info}: {err:.3e} > {limit:.1e}", flush=True)


t0 = time.perf_counter()
for case in range(N):
    rng = np.random.default_rng(seed0 + case)
    try:
        # ---- contract_pair against einsum
        ra, rb = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        n_common = int(rng.integers(0, min(ra, rb) + 1))
        la = [int(x) for x in rng.choice(10, size=ra, replace=False)]
        lb = [int(x) for x in rng.choice(la, size=n_common, replace=False)] + [int(x) for x in rng.choice(range(10, 20), size=rb - n_common, replace=False)]
        lb = [int(x) for x in rng.permutation(lb)]
        dim_of = {x: int(rng.integers(1, 7)) for x in set(la) | set(lb)}
        a = rng.standard_normal([dim_of[x] for x in la])
        b = rng.standard_normal([dim_of[x] for x in lb])
        info = f"seed {seed0 + case} labels {la} x {lb} dims {dim_of}"
        c, labels = t4a.contract_pair(a, la, b, lb)
        names = {x: letters[i] for i, x in enumerate(sorted(set(la)))}
        names.update({x: letters[len(names) + i] for i, x in enumerate(sorted(set(lb) - set(la)))})
        out = [x for x in la if x not in lb] + [x for x in lb if x not in la]
        if labels != out:
            fails += 1
            print(f"FAIL contract labels {info}: {labels} vs {out}", flush=True)
        else:
            spec = "".join(names[x] for x in la) + "," + "".join(names[x] for x in lb) + "->" + "".join(names[x] for x in out)
            ref = np.einsum(spec, a, b)
            if np.shape(c) != ref.shape:
                fails += 1
                print(f"FAIL contract shape {info}: {np.shape(c)} vs {ref.shape}", flush=True)
            else:
                note("contract |C - einsum|", float(np.abs(c - ref).max()) if ref.size else 0.0, 1e-11, info)
        # ---- svd_with / qr_with over a random bipartition
        rank = int(rng.integers(2, 6))
        dims = [int(rng.integers(1, 7)) for _ in range(rank)]
        labels = [int(x) for x in rng.choice(50, size=rank, replace=False)]
        scale = scales[case % len(scales)]
        t = rng.standard_normal(dims) * scale
        nl = int(rng.integers(1, rank))
        left = [int(x) for x in rng.choice(labels, size=nl, replace=False)]
        info = f"seed {seed0 + case} dims {dims} labels {labels} left {left} scale {scale:g}"
        perm = [labels.index(x) for x in left] + [k for k in range(rank) if labels[k] not in left]
        mat = np.transpose(t, perm).reshape(int(np.prod([dims[p] for p in perm[:nl]])), -1, order="F")
        k = min(mat.shape)
        amax = float(np.abs(mat).max()) or 1.0
        u, s, v = t4a.tensor_svd(t, labels, left, truncate=False)
        um, vm = u.reshape(mat.shape[0], k, order="F"), v.reshape(mat.shape[1], k, order="F")
        sref = np.linalg.svd(mat, compute_uv=False)
        note("svd_with finite", 0.0 if (np.all(np.isfinite(um)) and np.all(np.isfinite(vm)) and np.all(np.isfinite(s))) else 1.0, 0.0, info)
        note("svd_with |U S V^T - A| / |A|max", float(np.abs((um * (s / amax)) @ vm.T - mat / amax).max()), 1e-11, info)
        note("svd_with |s - s_ref| / s_max", float(np.abs(s / sref[0] - sref / sref[0]).max()) if sref[0] > 0 else 0.0, 1e-11, info)
        note("svd_with |U^T U - I|", float(np.abs(um.T @ um - np.eye(k)).max()), 1e-10, info)
        q, r = t4a.tensor_qr(t, labels, left, truncate=False)
        qm, rm = q.reshape(mat.shape[0], k, order="F"), r.reshape(k, mat.shape[1], order="F")
        note("qr_with finite", 0.0 if (np.all(np.isfinite(qm)) and np.all(np.isfinite(rm))) else 1.0, 0.0, info)
        note("qr_with |Q R - A| / |A|max", float(np.abs(qm @ (rm / amax) - mat / amax).max()), 1e-11, info)
        note("qr_with |Q^T Q - I|", float(np.abs(qm.T @ qm - np.eye(k)).max()), 1e-10, info)
    except Exception as exc:  # noqa: BLE001 (a soak reports and goes on)
        fails += 1
        print(f"FAIL seed {seed0 + case}: exception {type(exc).__name__}: {exc}", flush=True)
print(f"{N} cases from seed {seed0}: {fails} failures; " + "; ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())) + f"; {time.perf_counter() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
