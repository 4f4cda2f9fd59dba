#!/usr/bin/env python3
"""Device time of every component that is parity-tested but sits outside the headline, each with the CPU oracle's time for the same
call beside it (one thread; the oracle is the restatement of the reference's algorithm, oracle/): the host-callback path of the
TCI2 sweep (the path a Rust closure takes), the rook pivot search, TreeTCI on a chain and on the 7-site sample tree, the quantics
front end, ACI, TensorCI2::from_tensor_train, compress, evaluate_many, and the dense kernels (GEMM / SVD / QR / solve).

    python tools/bench_components.py [--quick] [--only NAME[,NAME...]] [--no-oracle]

prints ONE JSON object; bench.py embeds it as aux["components"].  Nothing here is the headline metric, nothing here is inside the
headline's timed region.  Wall times include the host round trip of the C ABI call (upload / download where the entry point takes
host buffers), because that is what a caller of the library sees."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def best_of(fn, reps=3):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best * 1e3, out


class NativeCallback:
    """tools/native_callback.c: the built-in integrand as a NATIVE t4a_gpu_batch_eval_fn (one host thread unless T4A_CB_THREADS)."""

    class Ctx(ctypes.Structure):
        _fields_ = [("fid", ctypes.c_int32), ("n_acc", ctypes.c_int32), ("params", ctypes.c_double * 12),
                    ("weights", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("total", ctypes.c_uint64),
                    ("calls", ctypes.c_uint64), ("points", ctypes.c_uint64)]

    def __init__(self, spec):
        path = os.path.join(ROOT, "tools", "libnative_callback.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["gcc", "-O3", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", "-o", path,
                                   os.path.join(ROOT, "tools", "native_callback.c")])
        self.lib = ctypes.CDLL(path)
        self.w = np.ascontiguousarray(spec.weights, dtype=np.uint64)
        self.off = np.ascontiguousarray(np.concatenate([[0], np.cumsum(spec.local_dims)[:-1]]), dtype=np.uint64)
        self.ctx = self.Ctx()
        self.ctx.fid, self.ctx.n_acc = spec.fid, spec.n_acc
        for i in range(12):
            self.ctx.params[i] = float(spec.params[i])
        self.ctx.weights = self.w.ctypes.data
        self.ctx.offset = self.off.ctypes.data
        self.ctx.total = int(sum(spec.local_dims))
        self.fn_addr = ctypes.cast(self.lib.t4a_native_batch_eval, ctypes.c_void_p).value
        self.ctx_addr = ctypes.addressof(self.ctx)

    def attach(self, tci):
        tci.set_callback_raw(self.fn_addr, self.ctx_addr, keepalive=self)


def rand_tt_cores(d, chi, seed, sd=2):
    rng = np.random.default_rng(seed)
    link = [min(sd ** (b + 1), sd ** (d - b - 1), chi) for b in range(d - 1)]
    return [rng.standard_normal((1 if s == 0 else link[s - 1], sd, link[s] if s < d - 1 else 1)) / np.sqrt(float(sd)) for s in range(d)]


class _Args:
    def __init__(self, quick=False, only="", no_oracle=False):
        self.quick, self.only, self.no_oracle = quick, only, no_oracle


def components(quick=False, only="", no_oracle=False):
    """The measurements as a dict (bench.py embeds it as aux["components"])."""
    args = _Args(quick, only, no_oracle)
    only = set(x for x in args.only.split(",") if x)
    import t4a_amd
    from t4a_amd.functions import quantics_osc2d, quantics_trig_exp
    ob = None
    if not args.no_oracle:
        import oracle_binding as ob  # CPU restatement: the checker and the baseline, never the product
    out = {"note": "wall ms per call incl. the host round trip of the C ABI; oracle = CPU restatement of the reference on ONE host thread "
                   "(same inputs, same options); ratio = oracle / device (> 1: the device is faster)"}

    def want(name):
        return not only or name in only

    def ratio(o):
        for k in list(o):
            if k.endswith("_ms") and k.startswith("device"):
                ok = "oracle" + k[len("device"):]
                if ok in o and o[k] > 0:
                    o["ratio" + k[len("device"):-3]] = o[ok] / o[k]
        return o

    # ---- 1. the TCI2 sweep through the host batch callback (BASELINE configs[2] shape, saturated) ------------------------------
    if want("callback_sweep"):
        try:
            d, chi = (30, 256) if not args.quick else (20, 64)
            spec = quantics_osc2d(d, k1=37, k2=53, k3=2111, eps=0.5, k4=16411, delta=0.5)
            o = lambda it: t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=it, ncheck_history=10 ** 6, nsearch=0,
                                               max_nglobal_pivot=0, seed=42)
            # built-in functor: grow to saturation on the fast path, then hand the SAME state to the callback path
            t = t4a_amd.TensorCI2([2] * d)
            t.set_function(spec)
            t.add_global_pivots([[0] * d])
            t.set_max_sample_value(1.0)
            t.optimize(o(10), final_sweep1site=False)
            t0 = time.perf_counter()
            t.optimize(o(2), final_sweep1site=False)
            builtin_ms = (time.perf_counter() - t0) * 1e3
            cb = NativeCallback(spec)
            cb.attach(t)
            t.optimize(o(2), final_sweep1site=False)  # warm-up of the per-bond path
            c0, p0 = cb.ctx.calls, cb.ctx.points
            t0 = time.perf_counter()
            t.optimize(o(2), final_sweep1site=False)
            cb_ms = (time.perf_counter() - t0) * 1e3
            calls, points = cb.ctx.calls - c0, cb.ctx.points - p0
            # opt-in: eight host threads evaluate every candidate matrix concurrently (t4a_gpu_tci2_set_callback_threads)
            t.set_callback_threads(8)
            t.optimize(o(2), final_sweep1site=False)
            t0 = time.perf_counter()
            t.optimize(o(2), final_sweep1site=False)
            cb8_ms = (time.perf_counter() - t0) * 1e3
            t.set_callback_threads(1)
            # the callback alone on the same number of points (what no backend can remove)
            rng = np.random.default_rng(0)
            n_probe = min(int(points), 1 << 20)
            idx = np.ascontiguousarray(rng.integers(0, 2, size=(n_probe, d)), dtype=np.uint32)
            buf = np.zeros(n_probe)
            fn = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p)(cb.fn_addr)
            t0 = time.perf_counter()
            fn(cb.ctx_addr, idx.ctypes.data, d, n_probe, buf.ctypes.data)
            per_point_ns = (time.perf_counter() - t0) / n_probe * 1e9
            out["callback_sweep"] = {
                "workload": f"d={d} chi={chi} interleaved 2-variable integrand, saturated full sweep (2 iterations of optimize incl. fill_site_tensors)",
                "builtin_functor_ms": builtin_ms, "native_callback_ms": cb_ms, "native_callback_ms_8_callback_threads": cb8_ms, "callback_calls": int(calls), "callback_points": int(points),
                "callback_ns_per_point_one_thread": per_point_ns, "callback_alone_ms": per_point_ns * points * 1e-6,
                "threads": int(os.environ.get("T4A_CB_THREADS", "1")), "link_dims_max": int(max(t.link_dims())),
                "note": "native_callback_ms - callback_alone_ms is what the backend adds around a user function: per-bond path (no device chain), "
                        "index decoding, upload of the evaluated candidate matrix",
            }
            if ob is not None and not args.quick:
                ot = ob.OracleTCI2([2] * d)
                ot.set_function(spec)
                ot.add_global_pivots([[0] * d])
                ot.set_max_sample_value(1.0)
                ot.optimize(o(10), final_sweep1site=False)
                t0 = time.perf_counter()
                ot.optimize(o(2), final_sweep1site=False)
                out["callback_sweep"]["oracle_full_sweep_ms"] = (time.perf_counter() - t0) * 1e3
        except Exception as e:  # noqa: BLE001 - a component must never break the others
            out["callback_sweep_error"] = repr(e)

    # ---- 2. rook pivot search (PivotSearchStrategy::Rook) -----------------------------------------------------------------------
    if want("rook"):
        try:
            res = {}
            for name, spec, d, chi, tol in (("cfg2", quantics_trig_exp(20), 20, 64, 1e-8),
                                            ("cfg3_reduced", quantics_osc2d(20, k1=37, k2=53, k3=211, eps=0.1), 20, 64 if not args.quick else 16, 1e-10)):
                o = t4a_amd.TCI2Options(tolerance=tol, max_bond_dim=chi, max_iter=8, nsearch=0, max_nglobal_pivot=0, seed=42,
                                        pivot_search=t4a_amd.TCI2Options.ROOK)
                of = t4a_amd.TCI2Options(tolerance=tol, max_bond_dim=chi, max_iter=8, nsearch=0, max_nglobal_pivot=0, seed=42)

                def run(opt):
                    t = t4a_amd.TensorCI2([2] * d)
                    t.set_function(spec)
                    t.crossinterpolate2([[0] * d], opt)
                    return t
                ms_rook, t = best_of(lambda: run(o), 3)
                ms_full, tf = best_of(lambda: run(of), 2)
                r = {"device_rook_ms": ms_rook, "device_full_ms": ms_full, "rank_rook": int(max(t.link_dims())), "rank_full": int(max(tf.link_dims()))}
                if ob is not None:
                    def orun():
                        ot = ob.OracleTCI2([2] * d)
                        ot.set_function(spec)
                        ot.set_pivot_search(1)
                        ot.crossinterpolate2([[0] * d], o)
                        return ot
                    r["oracle_rook_ms"], _ = best_of(orun, 1)
                res[name] = ratio(r)
            out["rook"] = res
        except Exception as e:  # noqa: BLE001
            out["rook_error"] = repr(e)

    # ---- 3. TreeTCI: a d=30 chain and the 7-site sample tree --------------------------------------------------------------------
    if want("tree"):
        try:
            res = {}
            n = 30 if not args.quick else 16
            chi = 64 if not args.quick else 16
            f = quantics_osc2d(n, eps=0.1)
            chain = [(s, s + 1) for s in range(n - 1)]
            seven = [(0, 1), (1, 2), (1, 3), (3, 4), (4, 5), (4, 6)]  # graph/tests.rs sample tree
            from t4a_amd.functions import lorentz
            for name, dims, edges, fn, cap in (("chain_d30", [2] * n, chain, f, chi), ("seven_site_tree", [4] * 7, seven, lorentz([4] * 7), 16)):
                opt = t4a_amd.TreeTciOptions(tolerance=1e-9, max_iter=6, max_bond_dim=cap, enable_global_pivots=False)

                def run():
                    t = t4a_amd.TreeTCI2(dims, edges)
                    t.set_function(fn)
                    ranks, errors = t.crossinterpolate2([[0] * len(dims)], opt)
                    t.materialize(0)
                    return ranks
                ms, ranks = best_of(run, 2)
                r = {"device_ms": ms, "sweeps": len(ranks), "rank": int(ranks[-1])}
                if ob is not None:
                    oo = ob.TreeOptions(tolerance=1e-9, max_iter=6, max_bond_dim=cap, enable_global_pivots=False)

                    def orun():
                        o_ = ob.OracleTreeTCI2(dims, edges, fn)
                        return o_.crossinterpolate2([[0] * len(dims)], oo)
                    r["oracle_ms"], _ = best_of(orun, 1)
                res[name] = ratio(r)
            out["tree"] = res
        except Exception as e:  # noqa: BLE001
            out["tree_error"] = repr(e)

    # ---- 4. quantics front end, two variables -----------------------------------------------------------------------------------
    if want("quantics"):
        try:
            R = 12 if not args.quick else 6

            def f2(x, y):
                return np.cos(7.0 * x) * np.exp(-y) + 0.1 * np.sin(13.0 * (x + y))
            fq = lambda p: float(f2(p[0], p[1]))
            fq.batched = lambda pts: f2(np.asarray(pts)[:, 0], np.asarray(pts)[:, 1])
            qo = t4a_amd.QtciOptions(tolerance=1e-8, max_bond_dim=64, max_iter=20, n_random_init_pivot=3, seed=5)
            ms, q = best_of(lambda: t4a_amd.quanticscrossinterpolate([R, R], fq, lower=[0.0, 0.0], upper=[1.0, 1.0], options=qo), 2)
            r = {"device_ms": ms, "rank": int(q.rank()), "workload": f"2 variables x {R} bits, interleaved, Python batch callback"}
            if ob is not None:
                oq = ob.QtciOptions(tolerance=1e-8, max_bond_dim=64, max_iter=20, n_random_init_pivot=3, seed=5)
                r["oracle_ms"], _ = best_of(lambda: ob.quanticscrossinterpolate([R, R], fq, lower=[0.0, 0.0], upper=[1.0, 1.0], options=oq), 1)
            out["quantics"] = ratio(r)
        except Exception as e:  # noqa: BLE001
            out["quantics_error"] = repr(e)

    # ---- 5. ACI: Hadamard product of two chi = 64 trains --------------------------------------------------------------------------
    if want("aci"):
        try:
            d, chi = (24, 64) if not args.quick else (12, 8)
            ca, cb_ = rand_tt_cores(d, chi, 1), rand_tt_cores(d, chi, 2)
            a, b = t4a_amd.SimpleTensorTrain(ca), t4a_amd.SimpleTensorTrain(cb_)
            cap = 128 if not args.quick else 32
            o = t4a_amd.AciOptions(tolerance=1e-10, max_bond_dim=cap, enable_global_guard=False, max_iters=6)
            ms, r_ = best_of(lambda: t4a_amd.elementwise_batched(t4a_amd.ACI_PRODUCT, [a, b], o), 2)
            r = {"device_ms": ms, "sweeps": len(r_.ranks), "rank": int(r_.ranks[-1]), "workload": f"Hadamard product of two random chi={chi} trains, d={d}, cap {cap}"}
            if ob is not None:
                oa, ob_ = ob.OracleTT(ca), ob.OracleTT(cb_)
                oo = ob.AciOptions(tolerance=1e-10, max_bond_dim=cap, enable_global_guard=False, max_iters=6)
                r["oracle_ms"], _ = best_of(lambda: ob.aci_elementwise(ob.ACI_PRODUCT, [oa, ob_], oo), 1)
            out["aci"] = ratio(r)
        except Exception as e:  # noqa: BLE001
            out["aci_error"] = repr(e)

    # ---- 6. tensor-train utilities: from_tensor_train, compress, evaluate_many ---------------------------------------------------
    if want("tt"):
        try:
            d, chi = (30, 256) if not args.quick else (16, 32)
            cores = rand_tt_cores(d, chi, 3)
            tt = t4a_amd.SimpleTensorTrain(cores)
            res = {}
            ms, tci = best_of(lambda: t4a_amd.TensorCI2.from_tensor_train(tt, tolerance=1e-10, max_bond_dim=chi, max_iter=2), 2)
            res["from_tensor_train"] = {"device_ms": ms, "rank": int(max(tci.link_dims())), "workload": f"random train d={d} chi={chi}, max_iter 2"}
            big = t4a_amd.SimpleTensorTrain(rand_tt_cores(d, chi // 2, 4)).add(t4a_amd.SimpleTensorTrain(rand_tt_cores(d, chi // 2, 5)))
            for name, method in (("compress_lu", t4a_amd.COMPRESS_LU), ("compress_svd", t4a_amd.COMPRESS_SVD)):
                ms, _ = best_of(lambda: big.compressed(method=method, tolerance=1e-10, max_bond_dim=chi // 2), 2)
                res[name] = {"device_ms": ms, "workload": f"sum of two chi={chi // 2} trains (bond {chi}) back to {chi // 2}"}
            rng = np.random.default_rng(0)
            pts = rng.integers(0, 2, size=(20000, d))
            ms, _ = best_of(lambda: tt.evaluate_many(pts), 3)
            res["evaluate_many"] = {"device_ms": ms, "workload": f"20 000 random points of the d={d} chi={chi} train"}
            if ob is not None:
                ott = ob.OracleTT(cores)
                res["from_tensor_train"]["oracle_ms"], _ = best_of(lambda: ott.to_tci2(tolerance=1e-10, max_bond_dim=chi, max_iter=2), 1)
                res["evaluate_many"]["oracle_ms"], _ = best_of(lambda: ott.evaluate_many(pts), 1)
                if not args.quick:
                    obig_cores = [big.site_tensor(s) for s in range(d)]
                    for name, method in (("compress_lu", 0), ("compress_svd", 2)):
                        def ocomp():
                            ob.OracleTT(obig_cores).compress(method=method, tolerance=1e-10, max_bond_dim=chi // 2)
                        res[name]["oracle_ms"], _ = best_of(ocomp, 1)
            out["tt"] = {k: ratio(v) for k, v in res.items()}
        except Exception as e:  # noqa: BLE001
            out["tt_error"] = repr(e)

    # ---- 7. dense kernels ----------------------------------------------------------------------------------------------------------
    if want("dense"):
        try:
            rng = np.random.default_rng(0)
            res = {}
            for (m, k, n) in ([(1024, 1024, 1024), (512, 256, 1400), (256, 256, 512)] if not args.quick else [(256, 256, 256)]):
                a, b = rng.standard_normal((m, k)), rng.standard_normal((k, n))
                ms, _ = best_of(lambda: t4a_amd.mat_mul(a, b), 3)
                r = {"device_ms": ms, "gflops_incl_transfers": 2.0 * m * n * k / ms / 1e6}
                if ob is not None and m * n * k <= 512 * 256 * 1400:
                    r["oracle_ms"], _ = best_of(lambda: ob.gemm(a, b), 1)
                res[f"mat_mul_{m}x{k}x{n}"] = ratio(r)
            for (m, n) in ([(512, 256), (300, 200), (64, 64)] if not args.quick else [(64, 64)]):
                a = rng.standard_normal((m, n))
                for name, fd, fo in (("svd", t4a_amd.svd_backend, None if ob is None else ob.svd), ("qr", t4a_amd.qr_backend, None if ob is None else ob.qr)):
                    ms, _ = best_of(lambda: fd(a), 3)
                    r = {"device_ms": ms}
                    if fo is not None:
                        r["oracle_ms"], _ = best_of(lambda: fo(a), 1)
                    # a LAPACK-class CPU path beside the oracle's plain restatement: numpy on ONE thread
                    # (threadpoolctl) — the honest number to hold the device against, not part of the ratio the loser list is built from
                    try:
                        from threadpoolctl import threadpool_limits
                        with threadpool_limits(limits=1):
                            r["numpy_lapack_one_thread_ms"], _ = best_of((lambda: np.linalg.svd(a, full_matrices=False)) if name == "svd" else (lambda: np.linalg.qr(a)), 5)
                    except Exception:  # noqa: BLE001
                        pass
                    res[f"{name}_{m}x{n}"] = ratio(r)
            n = 512 if not args.quick else 64
            a, b = rng.standard_normal((n, n)) + n * np.eye(n), rng.standard_normal((n, 2 * n))
            ms, _ = best_of(lambda: t4a_amd.solve_matrix(a, b), 3)
            r = {"device_ms": ms}
            if ob is not None:
                r["oracle_ms"], _ = best_of(lambda: ob.solve(a, b), 1)
            res[f"solve_{n}x{n}_rhs{2 * n}"] = ratio(r)
            for (m, n_, r_) in ([(685, 688, 256), (1024, 1024, 512)] if not args.quick else [(130, 130, 64)]):
                a = rng.uniform(-1, 1, size=(m, n_))
                ms, _ = best_of(lambda: t4a_amd.rrlu(a, max_bond_dim=r_), 3)
                r = {"device_ms": ms, "us_per_pivot_step_incl_transfers": ms * 1e3 / r_}
                if ob is not None:
                    r["oracle_ms"], _ = best_of(lambda: ob.rrlu(a, max_bond_dim=r_), 1)
                res[f"rrlu_{m}x{n_}_rank{r_}"] = ratio(r)
            out["dense"] = res
        except Exception as e:  # noqa: BLE001
            out["dense_error"] = repr(e)

    if want("round5"):
        # the additions of round 5: the edge-local step of TreeACI, the N-ary tensor-network contraction, the randomized SVD
        try:
            res = {}
            rng = np.random.default_rng(9)
            bonds, rows, cols = ([48, 32, 40], 512, 384) if not args.quick else ([5, 3], 60, 40)
            rf = [rng.standard_normal((b, rows)) for b in bonds]
            cf = [rng.standard_normal((b, cols)) for b in bonds]
            ms, u = best_of(lambda: t4a_amd.treeaci_local_update(rf, cf, t4a_amd.ACI_PRODUCT, max_bond_dim=128, tolerance=1e-10), 3)
            r = {"device_ms": ms, "rank": int(u.rank), "workload": f"{len(bonds)} inputs, cut bonds {bonds}, {rows} x {cols} candidates, product operator, cap 128"}
            if ob is not None:
                r["oracle_ms"], _ = best_of(lambda: ob.treeaci_local_update(rf, cf, ob.ACI_PRODUCT, max_bond_dim=128, tolerance=1e-10), 1)
            res["treeaci_local_update"] = ratio(r)
            d = 48 if not args.quick else 8
            shapes, labels = [(d, d, 6), (d, d), (d, d, 5), (d, d)], [[1, 2, 9], [2, 3], [3, 4, 8], [4, 1]]
            arrs = [rng.standard_normal(sh) for sh in shapes]
            handles = [t4a_amd.LabelledTensor(a_, l_) for a_, l_ in zip(arrs, labels)]
            ms, _ = best_of(lambda: t4a_amd.contract(handles).to_numpy(), 3)
            r = {"device_ms": ms, "workload": f"ring of four tensors, bond {d}, two dangling legs (6, 5); operands resident on the device"}
            if ob is not None:
                r["oracle_ms"], _ = best_of(lambda: ob.tensor_contract_many(arrs, labels), 1)
            r["note"] = "the oracle sums directly over every summed label (brute force, written for independence from the device path): not a performance comparison"
            res["contract_network"] = ratio(r)
            m, n, k = (512, 256, 32) if not args.quick else (64, 48, 8)
            q1, _ = np.linalg.qr(rng.standard_normal((m, n)))
            q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
            a = (q1 * 2.0 ** -np.arange(n)) @ q2.T
            ms, _ = best_of(lambda: t4a_amd.randomized_svd(a, k, oversample=8, power_iters=1, seed=1), 3)
            ms_full, _ = best_of(lambda: t4a_amd.svd_backend(a), 2)
            res["randomized_svd"] = {"device_ms": ms, "device_full_svd_ms": ms_full, "workload": f"{m} x {n}, sigma_i = 2^-i, rank {k} + 8, one power iteration"}
            out["round5"] = res
        except Exception as e:  # noqa: BLE001
            out["round5_error"] = repr(e)

    # what loses to one CPU thread (listed in DESIGN.md section 8)
    slower = []

    def walk(prefix, o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k.startswith("ratio") and isinstance(v, float) and v < 1.0:
                    slower.append(f"{prefix}{'' if k == 'ratio' else '.' + k[6:]}: {v:.2f}x")
                else:
                    walk(f"{prefix}.{k}" if prefix else k, v)
    walk("", out)
    out["slower_than_one_cpu_thread"] = slower
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="smaller sizes (smoke run)")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    print(json.dumps(components(a.quick, a.only, a.no_oracle)), flush=True)


if __name__ == "__main__":
    main()
