"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/t4a_gpu.h
declares, validates arguments and FAILS LOUDLY (never falls back to a CPU path) when no GPU is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "t4a_gpu.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t4a_gpu_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    import t4a_amd
    lib = ctypes.CDLL(t4a_amd.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"symbols declared in include/t4a_gpu.h but not exported: {missing}"


def test_version_and_device_count_never_fail():
    import t4a_amd
    assert t4a_amd.version().startswith("t4a-mi355x")
    assert t4a_amd.device_count() >= 0


def test_no_silent_cpu_fallback():
    """Without a GPU every compute entry point must return T4A_GPU_NO_DEVICE with a message."""
    import t4a_amd
    if t4a_amd.device_count() > 0:
        pytest.skip("a GPU is visible: the loud-failure path is covered on the CPU builder")
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.rrlu(np.eye(3))
    assert e.value.code == t4a_amd.NO_DEVICE and "no CPU fallback" in e.value.message
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.TensorCI2([2, 2])
    assert e.value.code == t4a_amd.NO_DEVICE
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.mat_mul(np.eye(2), np.eye(2))
    assert e.value.code == t4a_amd.NO_DEVICE
    # the widened rows fail the same way: tensor-train handle, SVD / QR, rook LUCI, adaptive patching
    for call in (lambda: t4a_amd.SimpleTensorTrain.constant([2, 2], 1.0),
                 lambda: t4a_amd.svd_backend(np.eye(3)),
                 lambda: t4a_amd.qr_backend(np.eye(3)),
                 lambda: t4a_amd.full_piv_lu_matrix(np.eye(3)),
                 lambda: t4a_amd.matrix_luci_factors_rook(np.eye(3)),
                 lambda: t4a_amd.adaptiveinterpolate(lambda i: 1.0, [2, 2], [[0, 0]], t4a_amd.TCI2Options()),
                 lambda: t4a_amd.TreeTCI2([2, 2, 2], [(0, 1), (1, 2)]),
                 lambda: t4a_amd.quanticscrossinterpolate_discrete([4, 4], lambda i: 1.0)):
        with pytest.raises(t4a_amd.T4aError) as e:
            call()
        assert e.value.code == t4a_amd.NO_DEVICE, e.value


def test_argument_validation_happens_before_the_device_is_touched():
    import t4a_amd
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.TensorCI2([3])  # local_dims needs >= 2 entries (tensorci2.rs:381-385)
    assert e.value.code == t4a_amd.INVALID_ARGUMENT
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.TensorCI2([2, 0])
    assert e.value.code == t4a_amd.INVALID_ARGUMENT
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.TCI2Options(max_bond_dim=0).to_c()
    assert e.value.code == t4a_amd.INVALID_ARGUMENT
    with pytest.raises(t4a_amd.T4aError):
        t4a_amd.mat_mul(np.zeros((2, 3)), np.zeros((2, 3)))
    with pytest.raises(t4a_amd.T4aError) as e:  # quanticstci/src/quantics_tci.rs:449-472
        t4a_amd.quanticscrossinterpolate_discrete([5, 5], lambda i: 1.0)
    assert e.value.code == t4a_amd.INVALID_ARGUMENT
    # tree graphs are validated on the host (treetci/src/graph.rs:51-106)
    for dims, edges in (([2, 2, 2], [(0, 1)]), ([2, 2, 2], [(0, 1), (1, 1)]), ([2, 2, 2, 2], [(0, 1), (1, 2), (0, 2)])):
        with pytest.raises(t4a_amd.T4aError) as e:
            t4a_amd.TreeTCI2(dims, edges)
        assert e.value.code == t4a_amd.INVALID_ARGUMENT


def test_dimensions_beyond_the_32_bit_index_range_are_refused():
    """The dense entry points take usize shapes but the kernels index with 32-bit integers: a dimension above INT_MAX is an
    INVALID_ARGUMENT before any narrowing cast (and before the device is touched)."""
    import t4a_amd
    lib = ctypes.CDLL(t4a_amd.LIB_PATH)
    big = ctypes.c_size_t(2 ** 32 + 1)
    one = ctypes.c_size_t(1)
    dummy = (ctypes.c_double * 4)()
    perm = (ctypes.c_size_t * 4)()
    npiv = ctypes.c_size_t(0)
    err = ctypes.c_double(0.0)
    lib.t4a_gpu_rrlu_f64.restype = ctypes.c_int32
    st = lib.t4a_gpu_rrlu_f64(dummy, big, one, ctypes.c_size_t(0), ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int32(1), perm, perm,
                              ctypes.byref(npiv), ctypes.byref(err))
    assert st == t4a_amd.INVALID_ARGUMENT
    lib.t4a_gpu_gemm_batched_f64.restype = ctypes.c_int32
    st = lib.t4a_gpu_gemm_batched_f64(one, big, one, one, dummy, dummy, dummy)
    assert st == t4a_amd.INVALID_ARGUMENT


def test_options_default_matches_reference_defaults():
    import t4a_amd
    o = t4a_amd.TCI2OptionsC()
    assert t4a_amd._lib.t4a_gpu_tci2_options_default(ctypes.byref(o)) == 0
    # TCI2Options::default (tensorci2.rs:152-170)
    assert (o.tolerance, o.max_iter, o.max_bond_dim, o.pivot_search, o.normalize_error) == (1e-8, 20, 0, 0, 1)
    assert (o.max_nglobal_pivot, o.nsearch, o.sweep_strategy, o.ncheck_history, o.strictly_nested) == (5, 5, 2, 3, 0)
    assert o.tol_margin_global_search == 10.0 and o.has_seed == 0
    d = t4a_amd.TCI2Options().to_c()
    for name, _ in t4a_amd.TCI2OptionsC._fields_:
        assert getattr(d, name) == getattr(o, name), name


def test_last_error_message_query_then_fill():
    import t4a_amd
    lib = t4a_amd._lib
    assert lib.t4a_gpu_tci2_options_default(None) == t4a_amd.NULL_POINTER
    need = ctypes.c_size_t(0)
    assert lib.t4a_gpu_last_error_message(None, 0, ctypes.byref(need)) == 0
    assert need.value > 1
    small = ctypes.create_string_buffer(2)
    assert lib.t4a_gpu_last_error_message(small, 2, None) == t4a_amd.BUFFER_TOO_SMALL
    buf = ctypes.create_string_buffer(need.value)
    assert lib.t4a_gpu_last_error_message(buf, need.value, None) == 0
    assert b"null" in buf.value


def test_builtin_function_specs_are_consistent_with_the_oracle_evaluation():
    """host-side weight tables (quantics bit weights, Lorentz squares) against a direct numpy evaluation"""
    import oracle_binding as ob
    from t4a_amd.functions import quantics_trig_exp, quantics_osc2d, lorentz, linear_sum
    rng = np.random.default_rng(0)
    spec = quantics_trig_exp(10, a=7.0, b=0.5, cc=0.3, cs=0.9)
    idx = rng.integers(0, 2, size=(50, 10))
    x = (idx * 2.0 ** -(np.arange(10) + 1)).sum(axis=1)
    ref = (0.3 * np.cos(7 * x) + 0.9 * np.sin(7 * x)) * np.exp(-0.5 * x)
    assert np.abs(ob.fn_eval(spec, idx) - ref).max() < 1e-14
    assert np.array_equal(spec.accumulators(idx)[:, 0], (idx * 2 ** np.arange(9, -1, -1)).sum(axis=1).astype(np.uint64))
    spec = linear_sum([3, 4, 5], scale=2.0, shift=-1.0, site_weights=[1, -2, 3])
    idx = np.stack([rng.integers(0, d, size=40) for d in (3, 4, 5)], axis=1)
    assert np.array_equal(ob.fn_eval(spec, idx), 2.0 * (idx[:, 0] - 2 * idx[:, 1] + 3 * idx[:, 2]) - 1.0)
    spec = lorentz([4, 4, 4], coeff=0.5)
    assert np.allclose(ob.fn_eval(spec, idx % 4), 0.5 / (((idx % 4) ** 2).sum(axis=1) + 1.0), rtol=0, atol=1e-16)
    spec = quantics_osc2d(8, k1=1, k2=2, k3=3, eps=0.25, k4=5, delta=0.5)
    idx = rng.integers(0, 2, size=(60, 8))
    qx = (idx[:, 0::2] * 2 ** np.arange(3, -1, -1)).sum(axis=1)
    qy = (idx[:, 1::2] * 2 ** np.arange(3, -1, -1)).sum(axis=1)
    x, y = qx / 16.0, qy / 16.0
    ref = np.cos(2 * np.pi * x) * np.cos(4 * np.pi * y) + 0.25 * np.sin(6 * np.pi * (x + y)) / (1 + x * x + y * y) \
        + 0.5 * np.cos(2 * np.pi * 5 * x * y)
    assert np.abs(ob.fn_eval(spec, idx) - ref).max() < 1e-13


def test_bench_gpus_flag_is_checked_against_the_launcher():
    """bench.py --gpus N: a launcher-provided WORLD_SIZE that disagrees is an error (no silent world-1 run), and the
    self-launch command line starts N ranks on 127.0.0.1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env,
                         timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        return 7

    real = subprocess.call
    subprocess.call = fake_call
    try:
        assert bench.self_launch(8) == 7
    finally:
        subprocess.call = real
    cmd = seen["cmd"]
    assert "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd and cmd[cmd.index("-m") + 1] == "torch.distributed.run"
