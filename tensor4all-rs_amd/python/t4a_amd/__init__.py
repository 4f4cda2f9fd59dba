"""t4a_amd — thin ctypes host bindings over the C ABI of libt4a_gpu.so (include/t4a_gpu.h).

The names mirror the Rust reference (tensor4all-core / tensor4all-tensorbackend / tensor4all-tensorci):
``rrlu``, ``matrix_luci_factors_from_matrix``, ``mat_mul``, ``solve_matrix``, ``triangular_solve_matrix``,
``TensorCI2``, ``crossinterpolate2``, ``TCI2Options``.  There is NO CPU fallback: every compute call raises
``T4aError`` (status T4A_GPU_NO_DEVICE) when no MI355X is visible, and importing raises if the shared
library has not been built (``python tensor4all-rs_amd/build.py``).
"""
import ctypes
import os

import numpy as np

from .functions import (FnSpec, quantics_trig_exp, quantics_osc2d, lorentz, linear_sum,  # noqa: F401
                        FN_MAX_PARAMS)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T4A_GPU_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libt4a_gpu.so"))  # env: A/B builds

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build it with `python tensor4all-rs_amd/build.py` "
                      "(the MI355X backend has no CPU fallback)")

_lib = ctypes.CDLL(LIB_PATH)

c_size_t = ctypes.c_size_t
c_double = ctypes.c_double
c_int32 = ctypes.c_int32
c_void_p = ctypes.c_void_p

SUCCESS = 0
NULL_POINTER = -1
INVALID_ARGUMENT = -2
BUFFER_TOO_SMALL = -5
INTERNAL_ERROR = -6
NOT_IMPLEMENTED = -7
NAN_ENCOUNTERED = -8
SINGULAR_MATRIX = -9
NO_DEVICE = -10
KERNEL_TIMEOUT = -11
CALLBACK_ERROR = -12

CONVERGED, MAX_BOND_DIMENSION, MAX_ITERATIONS = 0, 1, 2


class T4aError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[t4a_gpu status {code}] {message}")
        self.code = code
        self.message = message


def last_error_message():
    need = c_size_t(0)
    _lib.t4a_gpu_last_error_message(None, 0, ctypes.byref(need))
    buf = ctypes.create_string_buffer(max(need.value, 1))
    _lib.t4a_gpu_last_error_message(buf, len(buf), None)
    return buf.value.decode("utf-8", "replace")


def _check(status):
    if status != SUCCESS:
        raise T4aError(status, last_error_message())


def device_count():
    c = c_int32(0)
    _check(_lib.t4a_gpu_device_count(ctypes.byref(c)))
    return c.value


def set_device(dev):
    _check(_lib.t4a_gpu_set_device(c_int32(dev)))


def version():
    _lib.t4a_gpu_version.restype = ctypes.c_char_p
    return _lib.t4a_gpu_version().decode()


def diag_switches_enabled():
    """True when the loaded library was built with -DT4A_DIAG_SWITCHES (the experiment switches of tools/ are read from the environment);
    on a production build they are no-ops and an A/B arm labelled with one measures the default path."""
    _lib.t4a_gpu_diag_switches_enabled.restype = ctypes.c_int32
    return bool(_lib.t4a_gpu_diag_switches_enabled())


def stdrng_sample(seed, dims):
    """`StdRng::seed_from_u64(seed)` then `random_range(0..d)` for every d of `dims` (the stream of the reference's seeded searches)."""
    dims = np.ascontiguousarray(dims, dtype=np.uintp)
    out = np.zeros(dims.size, dtype=np.uintp)
    _check(_lib.t4a_gpu_stdrng_sample(ctypes.c_uint64(seed), dims.ctypes.data_as(ctypes.POINTER(c_size_t)), c_size_t(dims.size),
                                      out.ctypes.data_as(ctypes.POINTER(c_size_t))))
    return out.astype(np.int64)


def chacha_block(key_words, counter, stream, rounds):
    """One ChaCha block (16 little-endian words) — known-answer hook for the published vectors."""
    key = np.ascontiguousarray(key_words, dtype=np.uint32)
    assert key.size == 8
    out = np.zeros(16, dtype=np.uint32)
    _check(_lib.t4a_gpu_chacha_block(key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_uint64(counter), ctypes.c_uint64(stream),
                                     c_int32(rounds), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    return out


def _vp(a):
    return a.ctypes.data_as(c_void_p)


def siphash(msg, k0=0, k1=0, c_rounds=1, d_rounds=3):
    """SipHash-c-d of `msg` (std `DefaultHasher` = 1-3 with the zero key): known-answer hook (csrc/smallrng.hpp)."""
    m = np.frombuffer(bytes(msg), dtype=np.uint8) if len(msg) else np.zeros(1, dtype=np.uint8)
    out = ctypes.c_uint64(0)
    _check(_lib.t4a_gpu_siphash(_vp(m), c_size_t(len(msg)), ctypes.c_uint64(k0), ctypes.c_uint64(k1), c_int32(c_rounds), c_int32(d_rounds), ctypes.byref(out)))
    return int(out.value)


def smallrng_words(seed, n, state=None):
    """n outputs of rand 0.9 `SmallRng` (xoshiro256++): `seed_from_u64(seed)`, or the four state words given directly."""
    out = np.zeros(max(n, 1), dtype=np.uint64)
    st = None if state is None else np.ascontiguousarray(state, dtype=np.uint64)
    _check(_lib.t4a_gpu_smallrng_words(ctypes.c_uint64(seed), None if st is None else _vp(st), c_size_t(n), _vp(out)))
    return [int(v) for v in out[:n]]


def smallrng_sample(seed, dims):
    dims = np.ascontiguousarray(dims, dtype=np.uintp)
    out = np.zeros(max(dims.size, 1), dtype=np.uintp)
    _check(_lib.t4a_gpu_smallrng_sample(ctypes.c_uint64(seed), _vp(dims), c_size_t(dims.size), _vp(out)))
    return [int(v) for v in out[:dims.size]]


def smallrng_shuffle(seed, n):
    out = np.zeros(max(n, 1), dtype=np.uintp)
    _check(_lib.t4a_gpu_smallrng_shuffle(ctypes.c_uint64(seed), c_size_t(n), _vp(out)))
    return [int(v) for v in out[:n]]


def tree_edge_seed(seed, tag, u, v, history_len, n_pivots_i, n_pivots_j):
    """rng_for_edge's hash (tensor4all-treetci/src/proposer.rs:360-387)."""
    out = ctypes.c_uint64(0)
    _check(_lib.t4a_gpu_tree_edge_seed(ctypes.c_uint64(seed), tag.encode(), c_size_t(u), c_size_t(v), c_size_t(history_len), c_size_t(n_pivots_i),
                                       c_size_t(n_pivots_j), ctypes.byref(out)))
    return int(out.value)


def chacha8_standard_normal(seed, n, n_words=0):
    """(n standard normals, n_words key-stream words) of `ChaCha8Rng::seed_from_u64(seed)` (tensor4all-aci/src/random_tt.rs:31,143-150)."""
    out = np.zeros(max(n, 1))
    w = np.zeros(max(n_words, 1), dtype=np.uint32)
    _check(_lib.t4a_gpu_chacha8_standard_normal(ctypes.c_uint64(seed), c_size_t(n), _vp(out), c_size_t(n_words), _vp(w)))
    return out[:n], [int(x) for x in w[:n_words]]


def _f(a):
    """column-major float64 copy"""
    return np.asfortranarray(np.array(a, dtype=np.float64, copy=True))


def _p(arr):
    return arr.ctypes.data_as(c_void_p)


class TCI2OptionsC(ctypes.Structure):
    _fields_ = [("tolerance", c_double), ("max_iter", c_size_t), ("max_bond_dim", c_size_t),
                ("pivot_search", c_int32), ("normalize_error", c_int32), ("verbosity", c_size_t),
                ("max_nglobal_pivot", c_size_t), ("nsearch", c_size_t), ("sweep_strategy", c_int32),
                ("strictly_nested", c_int32), ("ncheck_history", c_size_t),
                ("tol_margin_global_search", c_double), ("has_seed", c_int32), ("reserved_", c_int32),
                ("seed", ctypes.c_uint64)]


class TCI2Options:
    """TCI2Options (tensorci2.rs:73-170). ``max_bond_dim=None`` <=> Rust ``None``."""
    FULL, ROOK = 0, 1
    FORWARD, BACKWARD, BACK_AND_FORTH = 0, 1, 2

    def __init__(self, tolerance=1e-8, max_iter=20, max_bond_dim=None, pivot_search=0, normalize_error=True,
                 verbosity=0, max_nglobal_pivot=5, nsearch=5, sweep_strategy=2, ncheck_history=3,
                 strictly_nested=False, tol_margin_global_search=10.0, seed=None):
        self.tolerance = tolerance
        self.max_iter = max_iter
        self.max_bond_dim = max_bond_dim
        self.pivot_search = pivot_search
        self.normalize_error = normalize_error
        self.verbosity = verbosity
        self.max_nglobal_pivot = max_nglobal_pivot
        self.nsearch = nsearch
        self.sweep_strategy = sweep_strategy
        self.ncheck_history = ncheck_history
        self.strictly_nested = strictly_nested
        self.tol_margin_global_search = tol_margin_global_search
        self.seed = seed

    def to_c(self):
        if self.max_bond_dim is not None and self.max_bond_dim <= 0:
            raise T4aError(INVALID_ARGUMENT, "max_bond_dim must be positive")
        o = TCI2OptionsC()
        o.tolerance = self.tolerance
        o.max_iter = self.max_iter
        o.max_bond_dim = 0 if self.max_bond_dim is None else int(self.max_bond_dim)
        o.pivot_search = self.pivot_search
        o.normalize_error = 1 if self.normalize_error else 0
        o.verbosity = self.verbosity
        o.max_nglobal_pivot = self.max_nglobal_pivot
        o.nsearch = self.nsearch
        o.sweep_strategy = self.sweep_strategy
        o.strictly_nested = 1 if self.strictly_nested else 0
        o.ncheck_history = self.ncheck_history
        o.tol_margin_global_search = self.tol_margin_global_search
        o.has_seed = 0 if self.seed is None else 1
        o.seed = 0 if self.seed is None else int(self.seed)
        return o


# ---------------------------------------------------------------------------------------- dense ops
class RrLU:
    """Result of rrlu (core/src/matrixlu.rs:69-84)."""

    def __init__(self, factored, row_perm, col_perm, npivots, error, left_orthogonal):
        self.factored = factored
        self.row_permutation = row_perm
        self.col_permutation = col_perm
        self.n_pivot = npivots
        self.error = error
        self.left_orthogonal = left_orthogonal

    def npivots(self):
        return self.n_pivot

    def row_indices(self):
        return self.row_permutation[: self.n_pivot].copy()

    def col_indices(self):
        return self.col_permutation[: self.n_pivot].copy()

    def left(self, permute=False):  # extract_lu_from_factorized (matrixlu.rs:614-668)
        m, n = self.factored.shape
        r = self.n_pivot
        l = np.tril(self.factored[:, :r]).copy()
        if self.left_orthogonal:
            for i in range(r):
                l[i, i] = 1.0
        if permute:
            out = np.zeros_like(l)
            out[self.row_permutation, :] = l
            return out
        return l

    def right(self, permute=False):
        r = self.n_pivot
        u = np.triu(self.factored[:r, :]).copy()
        if not self.left_orthogonal:
            for i in range(r):
                u[i, i] = 1.0
        if permute:
            out = np.zeros_like(u)
            out[:, self.col_permutation] = u
            return out
        return u

    def pivot_errors(self):
        d = np.array([self.factored[i, i] for i in range(self.n_pivot)])
        return np.concatenate([np.sqrt(d * d), [self.error]])

    def last_pivot_error(self):
        return self.error


def rrlu(a, max_bond_dim=None, rel_tol=1e-14, abs_tol=0.0, left_orthogonal=True):
    a = _f(a)
    m, n = a.shape
    rp = np.zeros(m, dtype=np.uintp)
    cp = np.zeros(n, dtype=np.uintp)
    npiv = c_size_t(0)
    err = c_double(0.0)
    _check(_lib.t4a_gpu_rrlu_f64(_p(a), c_size_t(m), c_size_t(n), c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                 c_double(rel_tol), c_double(abs_tol), c_int32(1 if left_orthogonal else 0), _p(rp),
                                 _p(cp), ctypes.byref(npiv), ctypes.byref(err)))
    return RrLU(a, rp.astype(np.int64), cp.astype(np.int64), npiv.value, err.value, left_orthogonal)


class MatrixLuciFactors:
    def __init__(self, row_indices, col_indices, pivot_errors, rank, left, right):
        self.row_indices = row_indices
        self.col_indices = col_indices
        self.pivot_errors = pivot_errors
        self.rank = rank
        self.left = left
        self.right = right


def matrix_luci_factors_from_matrix(a, max_bond_dim=None, rel_tol=1e-14, abs_tol=0.0, left_orthogonal=True):
    a = _f(a)
    m, n = a.shape
    k = min(m, n)
    rows = np.zeros(max(k, 1), dtype=np.uintp)
    cols = np.zeros(max(k, 1), dtype=np.uintp)
    pe = np.zeros(k + 1, dtype=np.float64)
    left = np.zeros(max(m * k, 1), dtype=np.float64)
    right = np.zeros(max(k * n, 1), dtype=np.float64)
    rank = c_size_t(0)
    _check(_lib.t4a_gpu_luci_f64(_p(a), c_size_t(m), c_size_t(n), c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                 c_double(rel_tol), c_double(abs_tol), c_int32(1 if left_orthogonal else 0),
                                 ctypes.byref(rank), _p(rows), _p(cols), _p(pe), _p(left), _p(right)))
    r = rank.value
    return MatrixLuciFactors(rows[:r].astype(np.int64), cols[:r].astype(np.int64), pe[: r + 1].copy(), r,
                             left[: m * r].reshape((m, r), order="F").copy(),
                             right[: r * n].reshape((r, n), order="F").copy())


def _luci_outputs(m, n):
    k = min(m, n)
    return (np.zeros(max(k, 1), dtype=np.uintp), np.zeros(max(k, 1), dtype=np.uintp), np.zeros(k + 1),
            np.zeros(max(m * k, 1)), np.zeros(max(k * n, 1)), c_size_t(0))


def _luci_result(m, n, rows, cols, pe, left, right, rank):
    r = rank.value
    return MatrixLuciFactors(rows[:r].astype(np.int64), cols[:r].astype(np.int64), pe[: r + 1].copy(), r,
                             left[: m * r].reshape((m, r), order="F").copy(),
                             right[: r * n].reshape((r, n), order="F").copy())


def matrix_luci_factors_rook(a, max_bond_dim=None, rel_tol=1e-14, abs_tol=0.0, left_orthogonal=True):
    """LazyBlockRookKernel + CrossFactors on a dense matrix (core/src/matrixluci/block_rook.rs)."""
    a = _f(a)
    m, n = a.shape
    rows, cols, pe, left, right, rank = _luci_outputs(m, n)
    _check(_lib.t4a_gpu_luci_rook_f64(_p(a), c_size_t(m), c_size_t(n),
                                      c_size_t(0 if max_bond_dim is None else max_bond_dim), c_double(rel_tol),
                                      c_double(abs_tol), c_int32(1 if left_orthogonal else 0), ctypes.byref(rank),
                                      _p(rows), _p(cols), _p(pe), _p(left), _p(right)))
    return _luci_result(m, n, rows, cols, pe, left, right, rank)


_FILL_BLOCK_CB = ctypes.CFUNCTYPE(None, c_void_p, ctypes.POINTER(c_size_t), c_size_t, ctypes.POINTER(c_size_t), c_size_t,
                                  ctypes.POINTER(c_double))


def matrix_luci_factors_from_blocks(nrows, ncols, fill_block, max_bond_dim=None, rel_tol=1e-14, abs_tol=0.0,
                                    left_orthogonal=True):
    """matrix_luci_factors_from_blocks (core/src/matrix_luci.rs:440): ``fill_block(rows, cols)`` returns the
    len(rows) x len(cols) block of the candidate matrix."""
    def _cb(ctx, rows, nr, cols, nc, out):
        r = [rows[i] for i in range(nr)]
        c = [cols[j] for j in range(nc)]
        blk = np.asarray(fill_block(r, c), dtype=np.float64).reshape(nr, nc)
        flat = np.asfortranarray(blk).reshape(-1, order="F")
        ctypes.memmove(out, flat.ctypes.data, flat.size * 8)

    cb = _FILL_BLOCK_CB(_cb)
    rows, cols, pe, left, right, rank = _luci_outputs(nrows, ncols)
    _check(_lib.t4a_gpu_luci_blocks_f64(c_size_t(nrows), c_size_t(ncols), cb, None,
                                        c_size_t(0 if max_bond_dim is None else max_bond_dim), c_double(rel_tol),
                                        c_double(abs_tol), c_int32(1 if left_orthogonal else 0), ctypes.byref(rank),
                                        _p(rows), _p(cols), _p(pe), _p(left), _p(right)))
    return _luci_result(nrows, ncols, rows, cols, pe, left, right, rank)


def mat_mul(a, b):
    a = _f(a)
    b = _f(b)
    m, k = a.shape
    k2, n = b.shape
    if k != k2:
        raise T4aError(INVALID_ARGUMENT, "mat_mul: inner dimension mismatch")
    c = np.zeros((m, n), dtype=np.float64, order="F")
    _check(_lib.t4a_gpu_gemm_f64(_p(a), _p(b), c_size_t(m), c_size_t(k), c_size_t(n), _p(c)))
    return c


def batched_mat_mul_same_shape(batch, m, k, n, a, b):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    c = np.zeros(batch * m * n, dtype=np.float64)
    _check(_lib.t4a_gpu_gemm_batched_f64(c_size_t(batch), c_size_t(m), c_size_t(k), c_size_t(n), _p(a), _p(b), _p(c)))
    return c


def triangular_solve_matrix(a, b, left_side, lower, transpose_a, unit_diagonal):
    a = _f(a)
    b = _f(b)
    x = np.zeros(b.shape, dtype=np.float64, order="F")
    _check(_lib.t4a_gpu_trsm_f64(_p(a), c_size_t(a.shape[0]), _p(b), c_size_t(b.shape[0]), c_size_t(b.shape[1]),
                                 c_int32(int(left_side)), c_int32(int(lower)), c_int32(int(transpose_a)),
                                 c_int32(int(unit_diagonal)), _p(x)))
    return x


def solve_matrix(a, b):
    a = _f(a)
    b = _f(b)
    x = np.zeros(b.shape, dtype=np.float64, order="F")
    _check(_lib.t4a_gpu_solve_f64(_p(a), c_size_t(a.shape[0]), _p(b), c_size_t(b.shape[1]), _p(x)))
    return x


def fn_eval(spec, local_dims, idx):
    """Evaluate a built-in function on the device. idx: (n_pts, n_sites) integer array."""
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uintp))
    n_pts, n_sites = idx.shape
    ld = np.asarray(local_dims, dtype=np.uintp)
    out = np.zeros(n_pts, dtype=np.float64)
    params = np.asarray(spec.params, dtype=np.float64)
    w = np.ascontiguousarray(spec.weights, dtype=np.uint64)
    _check(_lib.t4a_gpu_fn_eval(c_int32(spec.fid), c_int32(spec.n_acc), _p(params), _p(w), _p(ld), c_size_t(n_sites),
                                _p(idx), c_size_t(n_pts), _p(out)))
    return out


# ---------------------------------------------------------------------------------------- TCI2
_BATCH_CB = ctypes.CFUNCTYPE(ctypes.c_int64, c_void_p, ctypes.POINTER(ctypes.c_uint32), c_size_t, c_size_t,
                             ctypes.POINTER(c_double))


_ALLGATHER_CB = ctypes.CFUNCTYPE(c_int32, c_void_p, ctypes.POINTER(c_double), c_size_t, ctypes.POINTER(c_double))


def pi_shard_eval(rank, world, f, a_digits, a0, b_digits, b0, n_sites, gather=None):
    """The host part of the column-block shard alone (t4a_gpu_pi_shard_eval; no device needed): the len(a) x len(b) matrix of f on
    the row halves `a_digits` (placed at site a0) x column halves `b_digits` (at site b0), evaluated by THIS rank's column block and
    all-gathered.  Returns the row-major matrix."""
    a = np.ascontiguousarray(a_digits, dtype=np.uint32)
    b = np.ascontiguousarray(b_digits, dtype=np.uint32)
    na, wa = a.shape
    nb, wb = b.shape
    out = np.zeros((na, nb))
    cb = _batch_callback(f)

    def _g(ctx, send_ptr, count, recv_ptr):
        try:
            send = np.ctypeslib.as_array(send_ptr, shape=(count,))
            got = np.asarray(gather(send), dtype=np.float64).ravel()
            if got.size != world * count:
                return 2
            np.ctypeslib.as_array(recv_ptr, shape=(world * count,))[:] = got
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    g = _ALLGATHER_CB(_g) if world > 1 else ctypes.cast(None, _ALLGATHER_CB)
    _check(_lib.t4a_gpu_pi_shard_eval(c_size_t(rank), c_size_t(world), cb, None, g, None, c_size_t(n_sites),
                                      a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), c_size_t(wa), c_size_t(a0), c_size_t(na),
                                      b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), c_size_t(wb), c_size_t(b0), c_size_t(nb), _p(out)))
    return out


def _batch_callback(f):
    """ctypes batch callback (t4a_gpu_batch_eval_fn) around a Python function: f(list_of_int) -> float, or an object with a
    `.batched(idx_array[n_pts, n_sites]) -> values` attribute."""
    scalar = f
    batched = getattr(f, "batched", None)

    def _cb(ctx, idx_ptr, n_sites, n_pts, out_ptr):
        try:
            idx = np.ctypeslib.as_array(idx_ptr, shape=(n_pts, n_sites))
            if batched is not None:
                vals = np.asarray(batched(idx), dtype=np.float64).ravel()
            else:
                vals = np.array([scalar([int(v) for v in row]) for row in idx], dtype=np.float64)
            k = min(len(vals), n_pts)
            out = np.ctypeslib.as_array(out_ptr, shape=(n_pts,))
            out[:k] = vals[:k]
            return len(vals)
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -1

    return _BATCH_CB(_cb)


def opt_first_pivot(f, local_dims, first_pivot, max_sweep=1000):
    """opt_first_pivot (tensorci/src/optfirstpivot.rs:40): greedy coordinate search for a large |f|."""
    ld = np.asarray(list(local_dims), dtype=np.uintp)
    fp = np.asarray(list(first_pivot), dtype=np.uintp)
    if len(fp) != len(ld):
        raise T4aError(INVALID_ARGUMENT, "first pivot does not fit local_dims")
    out = np.zeros(max(len(ld), 1), dtype=np.uintp)
    cb = _batch_callback(f)
    _check(_lib.t4a_gpu_opt_first_pivot(cb, None, _p(ld), c_size_t(len(ld)), _p(fp), c_size_t(max_sweep), _p(out)))
    return [int(v) for v in out[:len(ld)]]


class TensorCI2:
    """TensorCI2<f64> (tensorci/src/tensorci2.rs:349) — state lives in a device-side handle."""

    def __init__(self, local_dims):
        self.local_dims = [int(d) for d in local_dims]
        ld = np.asarray(self.local_dims, dtype=np.uintp)
        self._h = c_void_p()
        _check(_lib.t4a_gpu_tci2_new(_p(ld), c_size_t(len(ld)), ctypes.byref(self._h)))
        self._cb_keepalive = None
        self.n_callback_calls = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_tci2_release(h)
            self._h = None

    # --- function source
    def set_function(self, f):
        """f: FnSpec (built-in device functor) or a Python callable f(list[int]) -> float, optionally with a
        ``batched`` callable taking an (n_pts, n_sites) array (wrapped into the host batch callback)."""
        if isinstance(f, FnSpec):
            params = np.asarray(f.params, dtype=np.float64)
            w = np.ascontiguousarray(f.weights, dtype=np.uint64)
            _check(_lib.t4a_gpu_tci2_set_builtin_function(self._h, c_int32(f.fid), c_int32(f.n_acc), _p(params), _p(w)))
            self._cb_keepalive = None
            return
        scalar = f
        batched = getattr(f, "batched", None)
        owner = self

        def _cb(ctx, idx_ptr, n_sites, n_pts, out_ptr):
            try:
                owner.n_callback_calls += 1
                idx = np.ctypeslib.as_array(idx_ptr, shape=(n_pts, n_sites))
                if batched is not None:
                    vals = np.asarray(batched(idx), dtype=np.float64).ravel()
                else:
                    vals = np.array([scalar([int(v) for v in row]) for row in idx], dtype=np.float64)
                k = min(len(vals), n_pts)
                out = np.ctypeslib.as_array(out_ptr, shape=(n_pts,))
                out[:k] = vals[:k]
                return len(vals)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1

        cb = _BATCH_CB(_cb)
        self._cb_keepalive = cb
        _check(_lib.t4a_gpu_tci2_set_callback(self._h, cb, None))

    def set_callback_raw(self, fn_ptr, ctx_ptr, keepalive=None):
        """A NATIVE batch callback (t4a_gpu_batch_eval_fn as an address or ctypes function, ctx as an address): the path a Rust closure
        takes, without the Python interpreter between the library and the function (tools/native_callback.c)."""
        self._cb_keepalive = keepalive
        cb = ctypes.cast(fn_ptr, _BATCH_CB) if not isinstance(fn_ptr, _BATCH_CB) else fn_ptr
        _check(_lib.t4a_gpu_tci2_set_callback(self._h, cb, ctypes.c_void_p(ctx_ptr)))

    def set_pi_shard(self, rank, world, gather=None):
        """Column-block shard of every callback-evaluated candidate matrix over a process group.
        gather(send: np.ndarray[count]) -> np.ndarray[world * count] (rank-major all-gather on host buffers), e.g.
        parallel.PiShardGather(dist, torch); world == 1 switches the shard off."""
        if world > 1 and gather is None:
            raise T4aError(NULL_POINTER, "a column-block shard over more than one rank needs an all-gather callable")

        def _g(ctx, send_ptr, count, recv_ptr):
            try:
                send = np.ctypeslib.as_array(send_ptr, shape=(count,))
                got = np.asarray(gather(send), dtype=np.float64).ravel()
                if got.size != world * count:
                    return 2
                np.ctypeslib.as_array(recv_ptr, shape=(world * count,))[:] = got
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        cb = _ALLGATHER_CB(_g) if world > 1 else ctypes.cast(None, _ALLGATHER_CB)
        self._gather_keepalive = cb
        _check(_lib.t4a_gpu_tci2_set_pi_shard(self._h, c_size_t(rank), c_size_t(world), cb, None))

    def pi_shard_stats(self):
        out = (c_size_t * 2)()
        _check(_lib.t4a_gpu_tci2_pi_shard_stats(self._h, out))
        return {"gathers": int(out[0]), "bytes_sent": int(out[1])}

    # --- reference API
    def __len__(self):
        return len(self.local_dims)

    def add_global_pivots(self, pivots):
        piv = np.ascontiguousarray(np.asarray(pivots, dtype=np.uintp).reshape(len(pivots), len(self.local_dims)))
        if piv.size and piv.shape[1] != len(self.local_dims):
            raise T4aError(INVALID_ARGUMENT, "Pivot length must match number of sites")
        _check(_lib.t4a_gpu_tci2_add_global_pivots(self._h, _p(piv), c_size_t(len(pivots))))

    def crossinterpolate2(self, initial_pivots, options):
        o = options.to_c()
        piv = np.ascontiguousarray(np.asarray(initial_pivots, dtype=np.uintp).reshape(len(initial_pivots), len(self.local_dims)))
        _check(_lib.t4a_gpu_tci2_crossinterpolate2(self._h, _p(piv), c_size_t(len(initial_pivots)), ctypes.byref(o)))

    def optimize(self, options, final_sweep1site=True):
        o = options.to_c()
        _check(_lib.t4a_gpu_tci2_optimize(self._h, ctypes.byref(o), c_int32(1 if final_sweep1site else 0)))

    def sweep2site(self, forward, options):
        o = options.to_c()
        _check(_lib.t4a_gpu_tci2_sweep2site(self._h, c_int32(1 if forward else 0), ctypes.byref(o)))

    def sweep1site(self, forward, rel_tol, abs_tol, max_bond_dim=None, update_tensors=True):
        if max_bond_dim is not None and max_bond_dim <= 0:
            raise T4aError(INVALID_ARGUMENT, "max_bond_dim must be positive")
        _check(_lib.t4a_gpu_tci2_sweep1site(self._h, c_int32(1 if forward else 0), c_double(rel_tol), c_double(abs_tol),
                                            c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                            c_int32(1 if update_tensors else 0)))

    def fill_site_tensors(self):
        _check(_lib.t4a_gpu_tci2_fill_site_tensors(self._h))

    def make_canonical(self, rel_tol, abs_tol, max_bond_dim=None):
        _check(_lib.t4a_gpu_tci2_make_canonical(self._h, c_double(rel_tol), c_double(abs_tol),
                                                c_size_t(0 if max_bond_dim is None else max_bond_dim)))

    def rank(self):
        v = c_size_t(0)
        _check(_lib.t4a_gpu_tci2_rank(self._h, ctypes.byref(v)))
        return v.value

    def link_dims(self):
        out = np.zeros(len(self.local_dims) - 1, dtype=np.uintp)
        _check(_lib.t4a_gpu_tci2_link_dims(self._h, _p(out)))
        return [int(x) for x in out]

    def max_sample_value(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_tci2_max_sample_value(self._h, ctypes.byref(v)))
        return v.value

    def set_max_sample_value(self, value):
        _check(_lib.t4a_gpu_tci2_set_max_sample_value(self._h, c_double(value)))

    def max_bond_error(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_tci2_max_bond_error(self._h, ctypes.byref(v)))
        return v.value

    def bond_errors(self):
        out = np.zeros(len(self.local_dims) - 1, dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_bond_errors(self._h, _p(out)))
        return out

    def pivot_errors(self):
        n = c_size_t(0)
        _check(_lib.t4a_gpu_tci2_pivot_errors(self._h, ctypes.byref(n), None))
        out = np.zeros(max(n.value, 1), dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_pivot_errors(self._h, ctypes.byref(n), _p(out)))
        return out[: n.value]

    def _index_set(self, which, site):
        cnt = c_size_t(0)
        wid = c_size_t(0)
        _check(_lib.t4a_gpu_tci2_index_set(self._h, c_int32(which), c_size_t(site), ctypes.byref(cnt), ctypes.byref(wid), None))
        out = np.zeros(max(cnt.value * wid.value, 1), dtype=np.uintp)
        _check(_lib.t4a_gpu_tci2_index_set(self._h, c_int32(which), c_size_t(site), ctypes.byref(cnt), ctypes.byref(wid),
                                           _p(out)))
        return out[: cnt.value * wid.value].reshape(cnt.value, wid.value).astype(np.int64)

    def i_set(self, site):
        return self._index_set(0, site)

    def j_set(self, site):
        return self._index_set(1, site)

    def set_index_set(self, which, site, entries):
        e = np.ascontiguousarray(np.asarray(entries, dtype=np.uintp))
        count = e.shape[0] if e.ndim == 2 else len(entries)
        _check(_lib.t4a_gpu_tci2_set_index_set(self._h, c_int32(which), c_size_t(site), c_size_t(count), _p(e)))

    def clear_history(self):
        _check(_lib.t4a_gpu_tci2_clear_history(self._h))

    def site_tensor(self, site):
        d = (c_size_t * 3)()
        _check(_lib.t4a_gpu_tci2_site_tensor(self._h, c_size_t(site), d, None))
        shape = (d[0], d[1], d[2])
        out = np.zeros(max(shape[0] * shape[1] * shape[2], 1), dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_site_tensor(self._h, c_size_t(site), d, _p(out)))
        return out[: shape[0] * shape[1] * shape[2]].reshape(shape, order="F").copy()

    def site_tensor_dims(self, site):
        d = (c_size_t * 3)()
        _check(_lib.t4a_gpu_tci2_site_tensor(self._h, c_size_t(site), d, None))
        return (d[0], d[1], d[2])

    def site_tensor_to_device(self, site, device_ptr):
        _check(_lib.t4a_gpu_tci2_site_tensor_device(self._h, c_size_t(site), c_void_p(device_ptr)))

    def set_site_tensor_from_device(self, site, dims3, device_ptr):
        d = (c_size_t * 3)(*[int(x) for x in dims3])
        _check(_lib.t4a_gpu_tci2_set_site_tensor_device(self._h, c_size_t(site), d, c_void_p(device_ptr)))

    def export_site_tensors_async(self, device_ptr, stride, consumer_stream):
        """Enqueue device-to-device copies of all cores to device_ptr + site*stride*8 and make `consumer_stream`
        (an integer hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream) wait for them."""
        _check(_lib.t4a_gpu_tci2_export_site_tensors_async(self._h, c_void_p(device_ptr), c_size_t(stride),
                                                           c_void_p(consumer_stream)))

    def export_site_shard_async(self, device_ptr, stride, consumer_stream):
        """Local sites (set_site_shard) -> device buffer [per_rank][stride], ordered after the fill in flight."""
        _check(_lib.t4a_gpu_tci2_export_site_shard_async(self._h, c_void_p(device_ptr), c_size_t(stride), c_void_p(consumer_stream)))

    def import_site_shard_async(self, device_ptr, stride, per_rank, producer_stream):
        """Remote sites out of the gathered device buffer [world][per_rank][stride] into this handle."""
        _check(_lib.t4a_gpu_tci2_import_site_shard_async(self._h, c_void_p(device_ptr), c_size_t(stride), c_size_t(per_rank),
                                                         c_void_p(producer_stream)))

    def set_keep_site_tensors(self, keep=True):
        _check(_lib.t4a_gpu_tci2_set_keep_site_tensors(self._h, c_int32(1 if keep else 0)))

    def set_site_shard(self, rank, world):
        _check(_lib.t4a_gpu_tci2_set_site_shard(self._h, c_size_t(rank), c_size_t(world)))
        self._site_shard = (int(rank), int(world))

    def site_shard(self):
        """(rank, world) of the site-sharded fill set by set_site_shard ((0, 1): not sharded)."""
        return getattr(self, "_site_shard", (0, 1))

    def history(self):
        n = c_size_t(0)
        _check(_lib.t4a_gpu_tci2_n_iterations(self._h, ctypes.byref(n)))
        ranks = np.zeros(max(n.value, 1), dtype=np.uintp)
        errors = np.zeros(max(n.value, 1), dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_history(self._h, _p(ranks), _p(errors)))
        return [int(r) for r in ranks[: n.value]], errors[: n.value].copy()

    def termination(self):
        v = c_int32(0)
        _check(_lib.t4a_gpu_tci2_termination(self._h, ctypes.byref(v)))
        return v.value

    def evaluate(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uintp).reshape(-1, len(self.local_dims)))
        out = np.zeros(idx.shape[0], dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_evaluate(self._h, _p(idx), c_size_t(idx.shape[0]), _p(out)))
        return out

    def sum(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_tci2_sum(self._h, ctypes.byref(v)))
        return v.value

    def to_tensor_train(self):
        """TensorCI2::to_tensor_train — a device-resident SimpleTensorTrain."""
        h = c_void_p()
        _check(_lib.t4a_gpu_tci2_to_tensor_train(self._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    @classmethod
    def from_tensor_train(cls, tt, tolerance=1e-12, max_bond_dim=None, max_iter=3):
        """TensorCI2::from_tensor_train (tensorci/src/conversion.rs:66-121)."""
        h = c_void_p()
        _check(_lib.t4a_gpu_tci2_from_tensor_train(tt._h, c_double(tolerance),
                                                   c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                                   c_size_t(max_iter), ctypes.byref(h)))
        self = cls.__new__(cls)
        self.local_dims = [int(d) for d in tt.site_dims()]
        self._h = h
        self._cb_keepalive = None
        self.n_callback_calls = 0
        return self

    def last_sweep_shapes(self):
        out = np.zeros(3 * (len(self.local_dims) - 1), dtype=np.uintp)
        _check(_lib.t4a_gpu_tci2_last_sweep_shapes(self._h, _p(out)))
        return out.reshape(-1, 3).astype(np.int64)

    def profile_enable(self, on=True):
        _check(_lib.t4a_gpu_tci2_profile_enable(self._h, c_int32(1 if on else 0)))

    def profile_reset(self):
        _check(_lib.t4a_gpu_tci2_profile_reset(self._h))

    def profile(self):
        out = np.zeros(16, dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_profile_get(self._h, _p(out)))
        keys = ["rrlu_ms", "rrlu_launches", "pi_ms", "pi_launches", "fill_ms", "fill_calls", "factor_ms",
                "factor_calls", "pivot_steps", "rrlu_bytes", "flops", "evals"]
        d = {k: float(out[i]) for i, k in enumerate(keys)}
        d.update(dom_ms=float(out[12]), dom_launches=float(out[13]), dom_bytes=float(out[14]), dom_code=int(out[15]))
        return d

    def profile_variants(self):
        """Rows {code, ms, launches, bytes, steps} per rrLU kernel instantiation since the last profile_reset."""
        n = c_size_t(0)
        _check(_lib.t4a_gpu_tci2_profile_variants(self._h, None, c_size_t(0), ctypes.byref(n)))
        out = np.zeros((max(n.value, 1), 5), dtype=np.float64)
        _check(_lib.t4a_gpu_tci2_profile_variants(self._h, _p(out), c_size_t(out.shape[0]), ctypes.byref(n)))
        return [dict(code=int(r[0]), ms=float(r[1]), launches=float(r[2]), bytes=float(r[3]), steps=float(r[4])) for r in out[: n.value]]


    def set_chain(self, enable=True, verify=False, event_timing=False, small_engine=True, small_stamps=False, fill_graph_relaxed=False, small_tile32=False):
        """Device-side bond chain on / off for this handle; verify: read the device tables back after every chain; event_timing:
        while profiling, time the rrLU launches with HIP events instead of the kernels' own time stamps; small_engine=False: the
        one-launch engine for small problems is not offered the handle's optimize calls."""
        _check(_lib.t4a_gpu_tci2_set_chain(self._h, c_int32(1 if enable else 0),
                                           c_int32((1 if verify else 0) | (2 if event_timing else 0) | (0 if small_engine else 4) | (8 if small_stamps else 0)
                                                   | (16 if fill_graph_relaxed else 0) | (32 if small_tile32 else 0))))

    def set_callback_threads(self, n):
        """Opt-in: evaluate one candidate matrix through the host callback on n host threads at once (thread-safe NATIVE callbacks only;
        a Python callable serialises on the interpreter lock).  Results are bitwise those of one thread."""
        _check(_lib.t4a_gpu_tci2_set_callback_threads(self._h, c_size_t(n)))

    def small_stats(self):
        """Small-problem engine (one launch per optimize call): dict(completed, iterations, handed_back, not_eligible,
        device_us=(input, iterations, final sweep + results) of the last launch, reason)."""
        out = (ctypes.c_uint64 * 16)()
        _check(_lib.t4a_gpu_tci2_small_stats(self._h, out))
        return {"completed": int(out[0]), "iterations": int(out[1]), "handed_back": int(out[2]), "not_eligible": int(out[3]),
                "device_us": (out[4] / 100.0, out[5] / 100.0, out[6] / 100.0), "reason": int(out[7]),
                "phase_cycles": dict(zip(("lists", "evaluation", "pivot_steps", "gather", "factors", "fill", "snapshots", "rest"), (int(out[8 + k]) for k in range(8))))}

    def rook_stats(self):
        out = (ctypes.c_uint64 * 4)()
        _check(_lib.t4a_gpu_tci2_rook_stats(self._h, out))
        return {"device_searches": int(out[0]), "device_visits": int(out[1]), "host_searches": int(out[2]), "host_syncs": int(out[3])}

    def fill_stats(self):
        out = (ctypes.c_uint64 * 3)()
        _check(_lib.t4a_gpu_tci2_fill_stats(self._h, out))
        return {"fills": int(out[0]), "graph_replays": int(out[1]), "graph_captures": int(out[2])}

    def chain_stats(self):
        """Device-side bond chain: dict(half_sweeps, bonds, fell_back, not_eligible, group_half_sweeps) of the 2-site half-sweeps since
        the handle was created, plus walked_sweeps (sweeps that ran as one persistent workgroup) and one_site_sweeps /
        one_site_not_eligible / one_site_fell_back for sweep1site."""
        out = np.zeros(5, dtype=np.uint64)
        _check(_lib.t4a_gpu_tci2_chain_stats(self._h, _p(out)))
        ext = np.zeros(4, dtype=np.uint64)
        _check(_lib.t4a_gpu_tci2_chain_stats_ext(self._h, _p(ext)))
        return dict(half_sweeps=int(out[0]), bonds=int(out[1]), fell_back=int(out[2]), not_eligible=int(out[3]),
                    group_half_sweeps=int(out[4]), walked_sweeps=int(ext[0]), one_site_sweeps=int(ext[1]),
                    one_site_not_eligible=int(ext[2]), one_site_fell_back=int(ext[3]))


def optimize_group(tcis, options, final_sweep1site=True):
    """optimize() on up to eight TensorCI2 handles at once, in lock-step from this thread (one XCD per handle): the per-GPU form
    of the patch farm.  Results on every handle are exactly those of handle.optimize(options, final_sweep1site)."""
    o = options.to_c()
    arr = (c_void_p * len(tcis))(*[t._h for t in tcis])
    _check(_lib.t4a_gpu_tci2_optimize_group(arr, c_size_t(len(tcis)), ctypes.byref(o), c_int32(1 if final_sweep1site else 0)))


def fill_site_tensors_group(tcis):
    """fill_site_tensors() on several TensorCI2 handles at once: all fills issued (each on its handle's own stream), then all
    completed.  Results are exactly those of handle.fill_site_tensors()."""
    arr = (c_void_p * len(tcis))(*[t._h for t in tcis])
    _check(_lib.t4a_gpu_tci2_fill_site_tensors_group(arr, c_size_t(len(tcis))))


def crossinterpolate2(f, local_dims, initial_pivots, options):
    """crossinterpolate2 (tensorci2.rs:1513). Returns the optimised TensorCI2; histories via .history()."""
    options.to_c()  # validate before anything else
    tci = TensorCI2(local_dims)
    tci.set_function(f)
    tci.crossinterpolate2(initial_pivots, options)
    return tci


# ---------------------------------------------------------------------------------------- svd / qr / full-piv LU
def svd_backend(a):
    """svd_backend (tensorbackend/src/backend.rs:709): thin (u, s, vt)."""
    a = _f(a)
    m, n = a.shape
    k = min(m, n)
    u = np.zeros((m, k), order="F")
    s = np.zeros(k)
    vt = np.zeros((k, n), order="F")
    _check(_lib.t4a_gpu_svd_f64(_p(a), c_size_t(m), c_size_t(n), _p(u), _p(s), _p(vt)))
    return u, s, vt


def randomized_svd(a, rank, oversample=8, power_iters=1, seed=0):
    """t4a_gpu_rsvd_f64: randomized rank-`rank` SVD (range finder + Jacobi SVD of the small projected matrix): (u, s, vt) with u
    m x rank, vt rank x n."""
    a = _f(a)
    m, n = a.shape
    u = np.zeros((m, rank), order="F")
    s = np.zeros(rank)
    vt = np.zeros((rank, n), order="F")
    _check(_lib.t4a_gpu_rsvd_f64(_p(a), c_size_t(m), c_size_t(n), c_size_t(rank), c_size_t(oversample), c_size_t(power_iters),
                                 ctypes.c_uint64(seed), _p(u), _p(s), _p(vt)))
    return u, s, vt


def qr_backend(a):
    """qr_backend (tensorbackend/src/backend.rs:742): thin (q, r)."""
    a = _f(a)
    m, n = a.shape
    k = min(m, n)
    q = np.zeros((m, k), order="F")
    r = np.zeros((k, n), order="F")
    _check(_lib.t4a_gpu_qr_f64(_p(a), c_size_t(m), c_size_t(n), _p(q), _p(r)))
    return q, r


def full_piv_lu_matrix(a):
    """full_piv_lu_matrix (tensorbackend/src/backend.rs:1022): (p, l, u, q) with p @ a @ q.T == l @ u."""
    a = _f(a)
    n = a.shape[0]
    if a.shape[1] != n:
        raise T4aError(INVALID_ARGUMENT, "full_piv_lu expects a square matrix")
    out = [np.zeros((n, n), order="F") for _ in range(4)]
    _check(_lib.t4a_gpu_full_piv_lu_f64(_p(a), c_size_t(n), *[_p(x) for x in out]))
    return tuple(out)


# ---------------------------------------------------------------------------------------- SimpleTensorTrain
COMPRESS_LU, COMPRESS_CI, COMPRESS_SVD = 0, 1, 2


class SimpleTensorTrain:
    """SimpleTensorTrain<f64> (simplett/src/tensortrain.rs:97) — site tensors live on the device."""

    def __init__(self, cores):
        cores = [np.asarray(c, dtype=np.float64) for c in cores]
        for c in cores:
            if c.ndim != 3:
                raise T4aError(INVALID_ARGUMENT, "site tensors must have three legs (left, site, right)")
        dims = np.array([c.shape for c in cores], dtype=np.uintp).reshape(-1)
        flat = np.ascontiguousarray(np.concatenate([c.reshape(-1, order="F") for c in cores])
                                    if cores else np.zeros(1))
        self._h = c_void_p()
        _check(_lib.t4a_gpu_tt_new(_p(dims) if len(cores) else None, c_size_t(len(cores)), _p(flat),
                                   ctypes.byref(self._h)))

    @classmethod
    def _adopt(cls, handle):
        self = cls.__new__(cls)
        self._h = handle
        return self

    @classmethod
    def constant(cls, site_dims, value):
        """SimpleTensorTrain::constant (tensortrain.rs:166-211)."""
        cores = [np.ones((1, int(d), 1)) for d in site_dims]
        if cores:
            cores[-1] = cores[-1] * value
        return cls(cores)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_tt_release(h)
            self._h = None

    def clone(self):
        h = c_void_p()
        _check(_lib.t4a_gpu_tt_clone(self._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    # arithmetic (simplett/src/arithmetic.rs:34-180, tensortrain.rs:264-345, :449-583)
    def add(self, other):
        h = c_void_p()
        _check(_lib.t4a_gpu_tt_add(self._h, other._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def sub(self, other):
        h = c_void_p()
        _check(_lib.t4a_gpu_tt_sub(self._h, other._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    __add__, __sub__ = add, sub

    def scale_mut(self, factor):
        _check(_lib.t4a_gpu_tt_scale(self._h, c_double(factor)))

    def scale(self, factor):
        r = self.clone()
        r.scale_mut(factor)
        return r

    def negate(self):
        return self.scale(-1.0)

    def inner_product(self, other):
        """inner_product (contraction.rs:82-186)"""
        v = c_double(0)
        _check(_lib.t4a_gpu_tt_inner_product(self._h, other._h, ctypes.byref(v)))
        return v.value

    def reverse(self):
        h = c_void_p()
        _check(_lib.t4a_gpu_tt_reverse(self._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def partial_sum(self, dims):
        d = np.asarray(list(dims), dtype=np.uintp)
        h = c_void_p()
        _check(_lib.t4a_gpu_tt_partial_sum(self._h, _p(d) if len(d) else None, c_size_t(len(d)), ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def __len__(self):
        v = c_size_t(0)
        _check(_lib.t4a_gpu_tt_len(self._h, ctypes.byref(v)))
        return v.value

    def dims(self):
        d = np.zeros(max(3 * len(self), 1), dtype=np.uintp)
        _check(_lib.t4a_gpu_tt_dims(self._h, _p(d)))
        return d[:3 * len(self)].reshape(-1, 3).astype(np.int64)

    def site_dims(self):
        return [int(x) for x in self.dims()[:, 1]]

    def link_dims(self):
        return [int(x) for x in self.dims()[1:, 0]]

    def rank(self):
        ld = self.link_dims()
        return max(ld) if ld else 1

    def site_tensor(self, site):
        l, s, r = (int(x) for x in self.dims()[site])
        buf = np.zeros(max(l * s * r, 1))
        _check(_lib.t4a_gpu_tt_site_tensor(self._h, c_size_t(site), _p(buf)))
        return buf[:l * s * r].reshape((l, s, r), order="F")

    def site_tensors(self):
        return [self.site_tensor(s) for s in range(len(self))]

    def evaluate(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uintp).reshape(-1, len(self)))
        out = np.zeros(idx.shape[0])
        _check(_lib.t4a_gpu_tt_evaluate(self._h, _p(idx), c_size_t(idx.shape[0]), _p(out)))
        return out

    def sum(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_tt_sum(self._h, ctypes.byref(v)))
        return v.value

    def norm2(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_tt_norm2(self._h, ctypes.byref(v)))
        return v.value

    def norm(self):
        return float(np.sqrt(self.norm2()))

    def compress(self, method=COMPRESS_LU, tolerance=1e-12, max_bond_dim=None, normalize_error=True):
        """compress(&CompressionOptions) (compression.rs:375); defaults = CompressionOptions::default()."""
        _check(_lib.t4a_gpu_tt_compress(self._h, c_int32(method), c_double(tolerance),
                                        c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                        c_int32(1 if normalize_error else 0)))

    def compressed(self, **kw):
        t = self.clone()
        t.compress(**kw)
        return t

    def evaluate_many(self, idx, split=None, return_split=False):
        """TTCache::evaluate_many (cache.rs:558)."""
        n = len(self)
        idx = np.asarray(idx, dtype=np.uintp)
        if idx.size == 0:
            return (np.zeros(0), split) if return_split else np.zeros(0)
        if idx.ndim != 2 or idx.shape[1] != n:
            raise T4aError(INVALID_ARGUMENT, f"index length mismatch: expected {n}")
        idx = np.ascontiguousarray(idx)
        out = np.zeros(idx.shape[0])
        used = c_size_t(0)
        if split is not None and split <= 0:
            raise T4aError(INVALID_ARGUMENT, f"Invalid split position: {split} (n_sites={n})")
        _check(_lib.t4a_gpu_tt_evaluate_many(self._h, _p(idx), c_size_t(idx.shape[0]),
                                             c_size_t(0 if split is None else split), _p(out), ctypes.byref(used)))
        return (out, used.value) if return_split else out

    def floating_zone(self, f, local_dims=None, init_p=None, early_stop_tol=float(np.finfo(np.float64).max), seed=0):
        """floating_zone (tensorci/src/globalsearch.rs:163): (pivot, max |f - tt|) of a greedy coordinate search."""
        ld = np.asarray(self.site_dims() if local_dims is None else list(local_dims), dtype=np.uintp)
        out = np.zeros(max(len(ld), 1), dtype=np.uintp)
        err = c_double(0.0)
        ip = None if init_p is None else np.asarray(list(init_p), dtype=np.uintp)
        if ip is not None and len(ip) != len(ld):
            raise T4aError(INVALID_ARGUMENT, "initial pivot does not fit local_dims")
        cb = _batch_callback(f)
        _check(_lib.t4a_gpu_tt_floating_zone(self._h, cb, None, _p(ld), c_size_t(len(ld)), None if ip is None else _p(ip),
                                             ctypes.c_uint64(seed), c_double(early_stop_tol), _p(out), ctypes.byref(err)))
        return [int(v) for v in out[:len(ld)]], err.value

    def estimate_true_error(self, f, nsearch=100, initial_points=None, seed=0):
        """estimate_true_error (globalsearch.rs:70): [(pivot, error)] sorted by descending error, duplicates removed."""
        n = len(self)
        pts = None if initial_points is None else np.ascontiguousarray(np.asarray(initial_points, dtype=np.uintp).reshape(-1, n))
        cap = nsearch if pts is None else pts.shape[0]
        piv = np.zeros((max(cap, 1), n), dtype=np.uintp)
        errs = np.zeros(max(cap, 1))
        n_out = c_size_t(0)
        cb = _batch_callback(f)
        _check(_lib.t4a_gpu_tt_estimate_true_error(self._h, cb, None, c_size_t(nsearch), None if pts is None else _p(pts),
                                                   c_size_t(0 if pts is None else pts.shape[0]), ctypes.c_uint64(seed), _p(piv), _p(errs),
                                                   c_size_t(cap), ctypes.byref(n_out)))
        return [([int(v) for v in piv[k]], float(errs[k])) for k in range(n_out.value)]

    def full_tensor(self):
        """full_tensor (tensortrain.rs:374): all values, leftmost site fastest."""
        sd = self.site_dims()
        total = int(np.prod(sd))
        grids = np.indices(sd[::-1]).reshape(len(sd), -1)[::-1].T  # leftmost fastest
        return self.evaluate(grids.reshape(total, len(sd)))


# ---------------------------------------------------------------------------------------- adaptive patching
class PartitionedTT:
    """Result of adaptiveinterpolate (partitionedtt/src/adaptive_interpolation.rs): patches in FIFO acceptance order."""

    def __init__(self, handle, local_dims):
        self._h = handle
        self.local_dims = [int(d) for d in local_dims]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_ptt_release(h)
            self._h = None

    def __len__(self):
        v = c_size_t(0)
        _check(_lib.t4a_gpu_ptt_len(self._h, ctypes.byref(v)))
        return v.value

    def projector(self, k):
        cnt = c_size_t(0)
        _check(_lib.t4a_gpu_ptt_projector(self._h, c_size_t(k), ctypes.byref(cnt), None, None))
        pos = np.zeros(max(cnt.value, 1), dtype=np.uintp)
        val = np.zeros(max(cnt.value, 1), dtype=np.uintp)
        _check(_lib.t4a_gpu_ptt_projector(self._h, c_size_t(k), ctypes.byref(cnt), _p(pos), _p(val)))
        return {int(pos[i]): int(val[i]) for i in range(cnt.value)}

    def projectors(self):
        return [self.projector(k) for k in range(len(self))]

    def patch(self, k):
        """SubDomainTT of patch k as a SimpleTensorTrain over all sites."""
        h = c_void_p()
        _check(_lib.t4a_gpu_ptt_patch_tt(self._h, c_size_t(k), ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def evaluate(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uintp).reshape(-1, len(self.local_dims)))
        out = np.zeros(idx.shape[0])
        _check(_lib.t4a_gpu_ptt_evaluate(self._h, _p(idx), c_size_t(idx.shape[0]), _p(out)))
        return out

    def dense(self):
        """all values, site 0 fastest (the order of the reference's dense test helper)"""
        sd = self.local_dims
        grids = np.indices(sd[::-1]).reshape(len(sd), -1)[::-1].T
        return self.evaluate(grids)


def adaptiveinterpolate(f, local_dims, initial_pivots, options, patch_order=None, n_initial_pivots=5,
                        recycle_pivots=False):
    """partitionedtt::adaptiveinterpolate (adaptive_interpolation.rs:58).  `f` as for TensorCI2.set_function."""
    o = options.to_c()
    ld = np.asarray([int(d) for d in local_dims], dtype=np.uintp)
    n = len(ld)
    piv = np.ascontiguousarray(np.asarray(initial_pivots, dtype=np.uintp).reshape(len(initial_pivots), n))
    po = None if patch_order is None else np.ascontiguousarray(np.asarray(patch_order, dtype=np.uintp))
    h = c_void_p()
    common = (_p(piv) if len(initial_pivots) else None, c_size_t(len(initial_pivots)), ctypes.byref(o),
              None if po is None else _p(po), c_size_t(n_initial_pivots), c_int32(1 if recycle_pivots else 0),
              ctypes.byref(h))
    if isinstance(f, FnSpec):
        params = np.asarray(f.params, dtype=np.float64)
        w = np.ascontiguousarray(f.weights, dtype=np.uint64)
        _check(_lib.t4a_gpu_adaptive_interpolate_builtin(_p(ld), c_size_t(n), c_int32(f.fid), c_int32(f.n_acc), _p(params),
                                                         _p(w), *common))
    else:
        scalar = f
        batched = getattr(f, "batched", None)

        def _cb(ctx, idx_ptr, n_sites, n_pts, out_ptr):
            try:
                idx = np.ctypeslib.as_array(idx_ptr, shape=(n_pts, n_sites))
                if batched is not None:
                    vals = np.asarray(batched(idx), dtype=np.float64).ravel()
                else:
                    vals = np.array([scalar([int(v) for v in row]) for row in idx], dtype=np.float64)
                k = min(len(vals), n_pts)
                out = np.ctypeslib.as_array(out_ptr, shape=(n_pts,))
                out[:k] = vals[:k]
                return len(vals)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1

        cb = _BATCH_CB(_cb)
        _check(_lib.t4a_gpu_adaptive_interpolate_callback(_p(ld), c_size_t(n), cb, None, *common))
    return PartitionedTT(h, local_dims)


# ---------------------------------------------------------------------------------------- tree TCI (tensor4all-treetci)
class TreeTciOptionsC(ctypes.Structure):
    _fields_ = [("tolerance", c_double), ("max_iter", c_size_t), ("max_bond_dim", c_size_t),
                ("normalize_error", c_int32), ("enable_global_pivots", c_int32), ("nsearch", c_size_t),
                ("max_nglobal_pivot", c_size_t), ("tol_margin_global_search", c_double), ("has_seed", c_int32),
                ("seed", ctypes.c_uint64)]


class TreeTciOptions:
    """TreeTciOptions (treetci/src/optimize.rs:13-76)."""

    def __init__(self, tolerance=1e-8, max_iter=20, max_bond_dim=None, normalize_error=True, enable_global_pivots=True,
                 nsearch=5, max_nglobal_pivot=5, tol_margin_global_search=10.0, seed=None):
        self.tolerance = tolerance
        self.max_iter = max_iter
        self.max_bond_dim = max_bond_dim
        self.normalize_error = normalize_error
        self.enable_global_pivots = enable_global_pivots
        self.nsearch = nsearch
        self.max_nglobal_pivot = max_nglobal_pivot
        self.tol_margin_global_search = tol_margin_global_search
        self.seed = seed

    def to_c(self):
        o = TreeTciOptionsC()
        _check(_lib.t4a_gpu_treetci_options_default(ctypes.byref(o)))
        o.tolerance = self.tolerance
        o.max_iter = self.max_iter
        o.max_bond_dim = 0 if self.max_bond_dim is None else self.max_bond_dim
        o.normalize_error = int(self.normalize_error)
        o.enable_global_pivots = int(self.enable_global_pivots)
        o.nsearch = self.nsearch
        o.max_nglobal_pivot = self.max_nglobal_pivot
        o.tol_margin_global_search = self.tol_margin_global_search
        o.has_seed = 0 if self.seed is None else 1
        o.seed = 0 if self.seed is None else self.seed
        return o


class TreeTCI2:
    """TreeTCI2<f64> on a TreeTciGraph (treetci/src/state.rs:41, graph.rs:44); edges = [(u, v), ...]."""

    def __init__(self, local_dims, edges):
        self.local_dims = [int(d) for d in local_dims]
        self.n = len(self.local_dims)
        ld = np.asarray(self.local_dims, dtype=np.uintp)
        e = np.ascontiguousarray(np.asarray(edges, dtype=np.uintp).reshape(-1, 2))
        self.n_edges = len(e)
        self._h = c_void_p()
        self._cb_keepalive = None
        self.n_callback_calls = 0
        _check(_lib.t4a_gpu_treetci_new(_p(ld), c_size_t(self.n), _p(e), c_size_t(len(e)), ctypes.byref(self._h)))
        self._edges = sorted((min(int(a), int(b)), max(int(a), int(b))) for a, b in e)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_treetci_release(h)
            self._h = None

    def set_function(self, f):
        """f as for TensorCI2.set_function; the batch callback receives the GlobalIndexBatch layout."""
        if isinstance(f, FnSpec):
            params = np.asarray(f.params, dtype=np.float64)
            w = np.ascontiguousarray(f.weights, dtype=np.uint64)
            _check(_lib.t4a_gpu_treetci_set_builtin_function(self._h, c_int32(f.fid), c_int32(f.n_acc), _p(params), _p(w)))
            self._cb_keepalive = None
            return
        scalar = f
        batched = getattr(f, "batched", None)
        owner = self

        def _cb(ctx, idx_ptr, n_sites, n_pts, out_ptr):
            try:
                owner.n_callback_calls += 1
                idx = np.ctypeslib.as_array(idx_ptr, shape=(n_pts, n_sites))
                if batched is not None:
                    vals = np.asarray(batched(idx), dtype=np.float64).ravel()
                else:
                    vals = np.array([scalar([int(v) for v in row]) for row in idx], dtype=np.float64)
                k = min(len(vals), n_pts)
                out = np.ctypeslib.as_array(out_ptr, shape=(n_pts,))
                out[:k] = vals[:k]
                return len(vals)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1

        cb = _BATCH_CB(_cb)
        self._cb_keepalive = cb
        _check(_lib.t4a_gpu_treetci_set_callback(self._h, cb, None))

    def edges(self):
        return list(self._edges)

    def set_proposer(self, kind, seed=0):
        """0 DefaultProposer, 1 SimpleProposer::seeded(seed), 2 TruncatedDefaultProposer::seeded(seed)"""
        _check(_lib.t4a_gpu_treetci_set_proposer(self._h, c_int32(kind), ctypes.c_uint64(seed)))

    def add_global_pivots(self, pivots):
        piv = np.ascontiguousarray(np.asarray(pivots, dtype=np.uintp).reshape(len(pivots), self.n))
        _check(_lib.t4a_gpu_treetci_add_global_pivots(self._h, _p(piv), c_size_t(len(pivots))))

    def subregion_vertices(self, u, v):
        nl, nr = c_size_t(0), c_size_t(0)
        _check(_lib.t4a_gpu_treetci_subregion_vertices(self._h, c_size_t(u), c_size_t(v), ctypes.byref(nl), None,
                                                       ctypes.byref(nr), None))
        l, r = np.zeros(nl.value, dtype=np.uintp), np.zeros(nr.value, dtype=np.uintp)
        _check(_lib.t4a_gpu_treetci_subregion_vertices(self._h, c_size_t(u), c_size_t(v), ctypes.byref(nl), _p(l),
                                                       ctypes.byref(nr), _p(r)))
        return [int(x) for x in l], [int(x) for x in r]

    def candidates(self, u, v):
        lk, rk = self.subregion_vertices(u, v)
        nl, nr = c_size_t(0), c_size_t(0)
        _check(_lib.t4a_gpu_treetci_candidates(self._h, c_size_t(u), c_size_t(v), ctypes.byref(nl), None, ctypes.byref(nr), None))
        l = np.zeros((nl.value, len(lk)), dtype=np.uintp)
        r = np.zeros((nr.value, len(rk)), dtype=np.uintp)
        _check(_lib.t4a_gpu_treetci_candidates(self._h, c_size_t(u), c_size_t(v), ctypes.byref(nl), _p(l), ctypes.byref(nr), _p(r)))
        return l.astype(np.int64), r.astype(np.int64)

    def update_edge(self, u, v, max_bond_dim=None, rel_tol=1e-14, abs_tol=0.0):
        nl, nr = c_size_t(0), c_size_t(0)
        _check(_lib.t4a_gpu_treetci_candidates(self._h, c_size_t(u), c_size_t(v), ctypes.byref(nl), None, ctypes.byref(nr), None))
        cap = min(nl.value, nr.value)
        rows, cols = np.zeros(cap, dtype=np.uintp), np.zeros(cap, dtype=np.uintp)
        errs = np.zeros(cap + 1)
        rank = c_size_t(0)
        _check(_lib.t4a_gpu_treetci_update_edge(self._h, c_size_t(u), c_size_t(v),
                                                c_size_t(0 if max_bond_dim is None else max_bond_dim), c_double(rel_tol),
                                                c_double(abs_tol), ctypes.byref(rank), _p(rows), _p(cols), _p(errs)))
        r = rank.value
        return {"rank": r, "row_indices": rows[:r].astype(np.int64), "col_indices": cols[:r].astype(np.int64),
                "pivot_errors": errs[:r + 1].copy()}

    def _history(self, fn, *args):
        o = args[-1]
        ranks = np.zeros(max(o.max_iter, 1), dtype=np.uintp)
        errors = np.zeros(max(o.max_iter, 1))
        n_iter = c_size_t(0)
        _check(fn(self._h, *args[:-1], ctypes.byref(o), ctypes.byref(n_iter), _p(ranks), _p(errors)))
        k = n_iter.value
        return [int(x) for x in ranks[:k]], [float(x) for x in errors[:k]]

    def optimize(self, options):
        """optimize_default: returns (ranks, errors) per sweep."""
        return self._history(_lib.t4a_gpu_treetci_optimize, options.to_c())

    def crossinterpolate2(self, initial_pivots, options):
        piv = np.ascontiguousarray(np.asarray(initial_pivots, dtype=np.uintp).reshape(len(initial_pivots), self.n))
        return self._history(_lib.t4a_gpu_treetci_crossinterpolate2, _p(piv), c_size_t(len(initial_pivots)), options.to_c())

    def find_global_pivots(self, nsearch, max_nglobal_pivot, tol_margin, abs_tol, seed):
        c = c_size_t(0)
        out = np.zeros((max(max_nglobal_pivot, 1), self.n), dtype=np.uintp)
        _check(_lib.t4a_gpu_treetci_find_global_pivots(self._h, c_size_t(nsearch), c_size_t(max_nglobal_pivot),
                                                       c_double(tol_margin), c_double(abs_tol), ctypes.c_uint64(seed),
                                                       ctypes.byref(c), _p(out)))
        return out[:c.value].astype(np.int64)

    def pivots(self, key):
        k = np.ascontiguousarray(np.asarray(sorted(key), dtype=np.uintp))
        c = c_size_t(0)
        _check(_lib.t4a_gpu_treetci_pivots(self._h, _p(k), c_size_t(len(k)), ctypes.byref(c), None))
        out = np.zeros((c.value, len(k)), dtype=np.uintp)
        _check(_lib.t4a_gpu_treetci_pivots(self._h, _p(k), c_size_t(len(k)), ctypes.byref(c), _p(out)))
        return out.astype(np.int64)

    def bond_errors(self):
        out = np.zeros(self.n_edges)
        _check(_lib.t4a_gpu_treetci_bond_errors(self._h, _p(out)))
        return out

    def pivot_errors(self):
        c = c_size_t(0)
        _check(_lib.t4a_gpu_treetci_pivot_errors(self._h, ctypes.byref(c), None))
        out = np.zeros(c.value)
        _check(_lib.t4a_gpu_treetci_pivot_errors(self._h, ctypes.byref(c), _p(out)))
        return out

    def flush_pivot_errors(self):
        _check(_lib.t4a_gpu_treetci_flush_pivot_errors(self._h))

    def max_sample_value(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_treetci_max_sample_value(self._h, ctypes.byref(v)))
        return v.value

    def set_max_sample_value(self, value):
        _check(_lib.t4a_gpu_treetci_set_max_sample_value(self._h, c_double(value)))

    def max_bond_error(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_treetci_max_bond_error(self._h, ctypes.byref(v)))
        return v.value

    def max_bond_dim(self):
        v = c_size_t(0)
        _check(_lib.t4a_gpu_treetci_max_bond_dim(self._h, ctypes.byref(v)))
        return v.value

    def materialize(self, center_site=0):
        """to_treetn: per-site dense tensors [site, incoming bonds..., bond to the parent] on the device."""
        _check(_lib.t4a_gpu_treetci_materialize(self._h, c_size_t(center_site)))

    def site_tensor(self, site):
        nd = c_size_t(0)
        dims = np.zeros(self.n + 1, dtype=np.uintp)
        _check(_lib.t4a_gpu_treetci_site_tensor(self._h, c_size_t(site), ctypes.byref(nd), _p(dims), None))
        shape = [int(x) for x in dims[:nd.value]]
        out = np.zeros(int(np.prod(shape)))
        _check(_lib.t4a_gpu_treetci_site_tensor(self._h, c_size_t(site), ctypes.byref(nd), _p(dims), _p(out)))
        return out.reshape(shape, order="F")

    def evaluate(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.uintp).reshape(-1, self.n))
        out = np.zeros(idx.shape[0], dtype=np.float64)
        _check(_lib.t4a_gpu_treetci_evaluate(self._h, _p(idx), c_size_t(idx.shape[0]), _p(out)))
        return out


def tree_crossinterpolate2(f, local_dims, edges, initial_pivots, options, center_site=None):
    """treetci::crossinterpolate2 (api.rs:21-96) with the default proposer: returns (tree, ranks, errors) with the
    network materialised around `center_site` (default 0)."""
    t = TreeTCI2(local_dims, edges)
    t.set_function(f)
    ranks, errors = t.crossinterpolate2(initial_pivots, options)
    t.materialize(0 if center_site is None else center_site)
    return t, ranks, errors


# ---------------------------------------------------------------------------------------- quantics front end (tensor4all-quanticstci)
INTERLEAVED, FUSED = 0, 1
_COORD_CB = ctypes.CFUNCTYPE(ctypes.c_int64, c_void_p, ctypes.POINTER(c_double), c_size_t, c_size_t, ctypes.POINTER(c_double))
_GRIDIDX_CB = ctypes.CFUNCTYPE(ctypes.c_int64, c_void_p, ctypes.POINTER(c_size_t), c_size_t, c_size_t, ctypes.POINTER(c_double))


class QtciOptionsC(ctypes.Structure):
    _fields_ = [("tolerance", c_double), ("max_bond_dim", c_size_t), ("max_iter", c_size_t), ("n_random_init_pivot", c_size_t),
                ("unfolding_scheme", c_int32), ("normalize_error", c_int32), ("has_seed", c_int32), ("seed", ctypes.c_uint64)]


class QtciOptions:
    """QtciOptions (quanticstci/src/options.rs:9-45); `seed` fixes the random initial pivots."""

    def __init__(self, tolerance=1e-8, max_bond_dim=None, max_iter=200, n_random_init_pivot=5, unfolding_scheme=INTERLEAVED,
                 normalize_error=True, seed=None):
        self.tolerance = tolerance
        self.max_bond_dim = max_bond_dim
        self.max_iter = max_iter
        self.n_random_init_pivot = n_random_init_pivot
        self.unfolding_scheme = unfolding_scheme
        self.normalize_error = normalize_error
        self.seed = seed

    def to_c(self):
        o = QtciOptionsC()
        _check(_lib.t4a_gpu_qtci_options_default(ctypes.byref(o)))
        o.tolerance = self.tolerance
        o.max_bond_dim = 0 if self.max_bond_dim is None else self.max_bond_dim
        o.max_iter = self.max_iter
        o.n_random_init_pivot = self.n_random_init_pivot
        o.unfolding_scheme = self.unfolding_scheme
        o.normalize_error = int(self.normalize_error)
        o.has_seed = 0 if self.seed is None else 1
        o.seed = 0 if self.seed is None else self.seed
        return o


def _batch_wrapper(f, cb_type, dtype, as_int):
    """f(point) -> float, optionally with f.batched((n_pts, n_vars) array) -> values."""
    batched = getattr(f, "batched", None)

    def _cb(ctx, ptr, n_vars, n_pts, out_ptr):
        try:
            pts = np.ctypeslib.as_array(ptr, shape=(n_pts, n_vars))
            if batched is not None:
                vals = np.asarray(batched(pts), dtype=np.float64).ravel()
            elif as_int:
                vals = np.array([f([int(v) for v in row]) for row in pts], dtype=np.float64)
            else:
                vals = np.array([f([float(v) for v in row]) for row in pts], dtype=np.float64)
            k = min(len(vals), n_pts)
            np.ctypeslib.as_array(out_ptr, shape=(n_pts,))[:k] = vals[:k]
            return len(vals)
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -1

    return cb_type(_cb)


def _qtci_pivots(pivots, n_vars):
    if pivots is None:
        return [c_int32(0), None, c_size_t(0)], None
    piv = np.ascontiguousarray(np.asarray(pivots, dtype=np.uintp).reshape(len(pivots), n_vars))
    return [c_int32(1), _p(piv), c_size_t(len(pivots))], piv


class QuanticsTensorCI2:
    """QuanticsTensorCI2<f64> (quanticstci/src/quantics_tci.rs:53-173): tensor train + grid + evaluation cache."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep
        ns, nv, disc = c_size_t(0), c_size_t(0), c_int32(0)
        _check(_lib.t4a_gpu_qtci_n_sites(self._h, ctypes.byref(ns), ctypes.byref(nv), ctypes.byref(disc)))
        self.n_sites, self.n_vars, self._disc = ns.value, nv.value, bool(disc.value)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_qtci_release(h)
            self._h = None

    def is_discretized(self):
        return self._disc

    def evaluate(self, grididx):
        g = np.ascontiguousarray(np.asarray(grididx, dtype=np.uintp).reshape(-1, self.n_vars))
        out = np.zeros(g.shape[0])
        _check(_lib.t4a_gpu_qtci_evaluate(self._h, _p(g), c_size_t(g.shape[0]), _p(out)))
        return out

    def sum(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_qtci_sum(self._h, ctypes.byref(v)))
        return v.value

    def integral(self):
        v = c_double(0)
        _check(_lib.t4a_gpu_qtci_integral(self._h, ctypes.byref(v)))
        return v.value

    def link_dims(self):
        out = np.zeros(max(self.n_sites - 1, 1), dtype=np.uintp)
        _check(_lib.t4a_gpu_qtci_link_dims(self._h, _p(out)))
        return [int(x) for x in out[:self.n_sites - 1]]

    def rank(self):
        ld = self.link_dims()
        return max(ld) if ld else 1

    def history(self):
        n = c_size_t(0)
        _check(_lib.t4a_gpu_qtci_history(self._h, ctypes.byref(n), None, None))
        ranks, errors = np.zeros(max(n.value, 1), dtype=np.uintp), np.zeros(max(n.value, 1))
        _check(_lib.t4a_gpu_qtci_history(self._h, ctypes.byref(n), _p(ranks), _p(errors)))
        return [int(x) for x in ranks[:n.value]], [float(x) for x in errors[:n.value]]

    def tensor_train(self):
        h = c_void_p()
        _check(_lib.t4a_gpu_qtci_tensor_train(self._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def tree_pivots(self, key):
        k = np.ascontiguousarray(np.asarray(sorted(key), dtype=np.uintp))
        c = c_size_t(0)
        _check(_lib.t4a_gpu_qtci_tree_pivots(self._h, _p(k), c_size_t(len(k)), ctypes.byref(c), None))
        out = np.zeros((c.value, len(k)), dtype=np.uintp)
        _check(_lib.t4a_gpu_qtci_tree_pivots(self._h, _p(k), c_size_t(len(k)), ctypes.byref(c), _p(out)))
        return out.astype(np.int64)

    def cachedata(self):
        c = c_size_t(0)
        _check(_lib.t4a_gpu_qtci_cachedata(self._h, ctypes.byref(c), None, None, None, None))
        q, v = np.zeros((c.value, self.n_sites), dtype=np.uintp), np.zeros(c.value)
        _check(_lib.t4a_gpu_qtci_cachedata(self._h, ctypes.byref(c), _p(q), _p(v), None, None))
        return {tuple(int(x) for x in row): float(val) for row, val in zip(q, v)}

    def user_call_stats(self):
        c, calls, pts = c_size_t(0), c_size_t(0), c_size_t(0)
        _check(_lib.t4a_gpu_qtci_cachedata(self._h, ctypes.byref(c), None, None, ctypes.byref(calls), ctypes.byref(pts)))
        return calls.value, pts.value

    def _grid(self, which, arr, n_out, floats=False):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.uintp))
        ou, od = np.zeros(max(n_out, 1), dtype=np.uintp), np.zeros(max(n_out, 1))
        _check(_lib.t4a_gpu_qtci_grid(self._h, c_int32(which), _p(a), _p(ou), _p(od)))
        return [float(x) for x in od[:n_out]] if floats else [int(x) for x in ou[:n_out]]

    def grididx_to_quantics(self, g):
        return self._grid(0, g, self.n_sites)

    def quantics_to_grididx(self, q):
        return self._grid(1, q, self.n_vars)

    def quantics_to_origcoord(self, q):
        return self._grid(2, q, self.n_vars, floats=True)

    def local_dimensions(self):
        return self._grid(3, [0], self.n_sites)

    def grid_step(self):
        return self._grid(4, [0], self.n_vars, floats=True)


def quanticscrossinterpolate(rs, f, lower=None, upper=None, include_endpoint=False, grid_unfolding=INTERLEAVED,
                             initial_pivots=None, options=None):
    """quanticscrossinterpolate(&DiscretizedGrid::builder(rs)..., f(coords), initial_pivots, options)."""
    options = options or QtciOptions()
    rs_a = np.asarray(rs, dtype=np.uintp)
    nv = len(rs_a)
    lo = None if lower is None else np.asarray(lower, dtype=np.float64)
    up = None if upper is None else np.asarray(upper, dtype=np.float64)
    cb = _batch_wrapper(f, _COORD_CB, np.float64, False)
    pa, keep = _qtci_pivots(initial_pivots, nv)
    o = options.to_c()
    h = c_void_p()
    _check(_lib.t4a_gpu_quanticscrossinterpolate(_p(rs_a), c_size_t(nv), None if lo is None else _p(lo),
                                                 None if up is None else _p(up), c_int32(int(include_endpoint)),
                                                 c_int32(grid_unfolding), cb, None, *pa, ctypes.byref(o), ctypes.byref(h)))
    return QuanticsTensorCI2(h, (cb, keep))


def quanticscrossinterpolate_discrete(sizes, f, initial_pivots=None, options=None):
    options = options or QtciOptions()
    sz = np.asarray(sizes, dtype=np.uintp)
    cb = _batch_wrapper(f, _GRIDIDX_CB, np.uintp, True)
    pa, keep = _qtci_pivots(initial_pivots, len(sz))
    o = options.to_c()
    h = c_void_p()
    _check(_lib.t4a_gpu_quanticscrossinterpolate_discrete(_p(sz), c_size_t(len(sz)), cb, None, *pa, ctypes.byref(o),
                                                          ctypes.byref(h)))
    return QuanticsTensorCI2(h, (cb, keep))


def quanticscrossinterpolate_from_arrays(xvals, f, initial_pivots=None, options=None):
    options = options or QtciOptions()
    sz = np.asarray([len(x) for x in xvals], dtype=np.uintp)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in xvals]) if len(xvals) else np.zeros(1))
    cb = _batch_wrapper(f, _COORD_CB, np.float64, False)
    pa, keep = _qtci_pivots(initial_pivots, len(sz))
    o = options.to_c()
    h = c_void_p()
    _check(_lib.t4a_gpu_quanticscrossinterpolate_from_arrays(_p(flat), _p(sz), c_size_t(len(sz)), cb, None, *pa,
                                                             ctypes.byref(o), ctypes.byref(h)))
    return QuanticsTensorCI2(h, (cb, keep))


_COORD_VEC_CB = ctypes.CFUNCTYPE(ctypes.c_int64, c_void_p, ctypes.POINTER(c_double), c_size_t, c_size_t, c_size_t,
                                 ctypes.POINTER(c_double))


def quanticscrossinterpolate_batched(rs, f, output_dims, lower=None, upper=None, include_endpoint=False,
                                     grid_unfolding=INTERLEAVED, initial_pivots=None, options=None):
    """quanticscrossinterpolate_batched (quanticstci/src/batched/mod.rs:50): f(coords) -> prod(output_dims) components.
    Returns (SimpleTensorTrain with one extra component site, ranks, errors, number of points the user function saw)."""
    options = options or QtciOptions()
    rs_a = np.asarray(rs, dtype=np.uintp)
    nv = len(rs_a)
    lo = None if lower is None else np.asarray(lower, dtype=np.float64)
    up = None if upper is None else np.asarray(upper, dtype=np.float64)
    od = np.asarray(output_dims, dtype=np.uintp)

    def _cb(ctx, ptr, n_vars, n_pts, n_comp, out_ptr):
        try:
            pts = np.ctypeslib.as_array(ptr, shape=(n_pts, n_vars))
            out = np.ctypeslib.as_array(out_ptr, shape=(n_pts, n_comp))
            total = 0
            for p, row in enumerate(pts):
                vals = np.asarray(f([float(v) for v in row]), dtype=np.float64).ravel()
                k = min(len(vals), n_comp)
                out[p, :k] = vals[:k]
                total += len(vals)
            return total
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return -1

    cb = _COORD_VEC_CB(_cb)
    pa, keep = _qtci_pivots(initial_pivots, nv)
    o = options.to_c()
    h = c_void_p()
    n_iter, upts = c_size_t(0), c_size_t(0)
    ranks, errors = np.zeros(max(options.max_iter, 1), dtype=np.uintp), np.zeros(max(options.max_iter, 1))
    _check(_lib.t4a_gpu_quanticscrossinterpolate_batched(
        _p(rs_a), c_size_t(nv), None if lo is None else _p(lo), None if up is None else _p(up), c_int32(int(include_endpoint)),
        c_int32(grid_unfolding), cb, None, _p(od) if len(od) else None, c_size_t(len(od)), *pa, ctypes.byref(o), ctypes.byref(h),
        ctypes.byref(n_iter), _p(ranks), _p(errors), ctypes.byref(upts)))
    k = n_iter.value
    return SimpleTensorTrain._adopt(h), [int(x) for x in ranks[:k]], [float(x) for x in errors[:k]], upts.value


# ---------------------------------------------------------------------------------------- dense labelled tensors (tensor4all-core defaults)
class SvdPolicyC(ctypes.Structure):
    _fields_ = [("threshold", c_double), ("scale", c_int32), ("measure", c_int32), ("rule", c_int32)]


RELATIVE, ABSOLUTE = 0, 1
VALUE, SQUARED_VALUE = 0, 1
PER_VALUE, DISCARDED_TAIL_SUM = 0, 1


class SvdTruncationPolicy:
    """SvdTruncationPolicy (core/src/truncation.rs:137): threshold x scale x measure x rule."""

    def __init__(self, threshold=1e-12, scale=RELATIVE, measure=VALUE, rule=PER_VALUE):
        self.threshold, self.scale, self.measure, self.rule = threshold, scale, measure, rule

    def to_c(self):
        return SvdPolicyC(self.threshold, self.scale, self.measure, self.rule)


def svd_retained_rank(s, policy=None):
    """compute_retained_rank (defaults/svd.rs:150): host-side rule, no device needed."""
    s = np.ascontiguousarray(np.asarray(s, dtype=np.float64))
    out = c_size_t(0)
    pc = None if policy is None else policy.to_c()
    _check(_lib.t4a_gpu_svd_retained_rank(_p(s) if len(s) else None, c_size_t(len(s)), None if pc is None else ctypes.byref(pc),
                                          ctypes.byref(out)))
    return out.value


def qr_retained_rank(r, k, n, rtol):
    """compute_retained_rank_qr_from_dense (defaults/qr.rs:74): r is k x n column-major."""
    r = np.ascontiguousarray(np.asarray(r, dtype=np.float64))
    out = c_size_t(0)
    _check(_lib.t4a_gpu_qr_retained_rank(_p(r) if len(r) else None, c_size_t(k), c_size_t(n), c_double(rtol), ctypes.byref(out)))
    return out.value


def _labelled(t, labels):
    a = np.asfortranarray(np.array(t, dtype=np.float64))
    lab = np.asarray(labels, dtype=np.int64)
    if len(lab) != a.ndim:
        raise T4aError(INVALID_ARGUMENT, "one label per tensor axis is required")
    return a, np.asarray(a.shape, dtype=np.uintp), lab


def contract_pair(a, a_labels, b, b_labels):
    """contract_pair (defaults/contract.rs:334): returns (array, labels) with lhs free axes then rhs free axes."""
    a, ad, al = _labelled(a, a_labels)
    b, bd, bl = _labelled(b, b_labels)
    cap = max(a.ndim + b.ndim, 1)
    od, ol, orank = np.zeros(cap, dtype=np.uintp), np.zeros(cap, dtype=np.int64), c_size_t(0)
    args = [_p(a), _p(ad), _p(al), c_size_t(a.ndim), _p(b), _p(bd), _p(bl), c_size_t(b.ndim)]
    _check(_lib.t4a_gpu_tensor_contract_f64(*args, None, _p(od), _p(ol), ctypes.byref(orank)))
    shape = [int(x) for x in od[:orank.value]]
    out = np.zeros(int(np.prod(shape)) if shape else 1)
    _check(_lib.t4a_gpu_tensor_contract_f64(*args, _p(out), _p(od), _p(ol), ctypes.byref(orank)))
    return out.reshape(shape, order="F"), [int(x) for x in ol[:orank.value]]


def _split(shape, labels, left):
    labels = list(labels)
    if not all(x in labels for x in left):
        return [1], [1]
    return [shape[labels.index(x)] for x in left], [d for d, x in zip(shape, labels) if x not in left]


def tensor_svd(t, labels, left, truncate=True, policy=None, max_bond_dim=None):
    """svd_with (defaults/svd.rs:347): returns (U [left.., r], S (r), V [right.., r])."""
    a, dims, lab = _labelled(t, labels)
    lf = np.asarray(left, dtype=np.int64)
    ld, rd = _split(list(a.shape), labels, list(left))
    m, n = int(np.prod(ld)), int(np.prod(rd))
    k = max(min(m, n), 1)
    u, s, v = np.zeros(m * k), np.zeros(k), np.zeros(n * k)
    r = c_size_t(0)
    pc = None if policy is None else policy.to_c()
    _check(_lib.t4a_gpu_tensor_svd_f64(_p(a), _p(dims), _p(lab), c_size_t(a.ndim), _p(lf) if len(lf) else None, c_size_t(len(lf)),
                                       c_int32(int(truncate)), None if pc is None else ctypes.byref(pc),
                                       c_int32(0 if max_bond_dim is None else 1), c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                       ctypes.byref(r), _p(u), _p(s), _p(v)))
    rr = r.value
    return u[:m * rr].reshape(ld + [rr], order="F"), s[:rr].copy(), v[:n * rr].reshape(rd + [rr], order="F")


def tensor_qr(t, labels, left, truncate=True, rtol=None):
    """qr_with (defaults/qr.rs:206): returns (Q [left.., r], R [r, right..])."""
    a, dims, lab = _labelled(t, labels)
    lf = np.asarray(left, dtype=np.int64)
    ld, rd = _split(list(a.shape), labels, list(left))
    m, n = int(np.prod(ld)), int(np.prod(rd))
    k = max(min(m, n), 1)
    q, rf = np.zeros(m * k), np.zeros(k * n)
    r = c_size_t(0)
    _check(_lib.t4a_gpu_tensor_qr_f64(_p(a), _p(dims), _p(lab), c_size_t(a.ndim), _p(lf) if len(lf) else None, c_size_t(len(lf)),
                                      c_int32(int(truncate)), c_int32(0 if rtol is None else 1), c_double(0.0 if rtol is None else rtol),
                                      ctypes.byref(r), _p(q), _p(rf)))
    rr = r.value
    return q[:m * rr].reshape(ld + [rr], order="F"), rf[:rr * n].reshape([rr] + rd, order="F")


def contract(tensors, retain=()):
    """contract / contract_with_options of tensor4all-core (defaults/contract.rs:283-298): ONE connected network of LabelledTensor
    operands; labels shared by several operands are summed unless listed in `retain` (ContractionOptions::retain_indices)."""
    arr = (c_void_p * max(len(tensors), 1))(*[t._h for t in tensors])
    rt = np.asarray(list(retain) or [0], dtype=np.int64)
    h = c_void_p()
    _check(_lib.t4a_gpu_tensor_contract_many(arr, c_size_t(len(tensors)), _p(rt), c_size_t(len(retain)), ctypes.byref(h)))
    return LabelledTensor(_handle=h)


class LabelledTensor:
    """Device-resident dense tensor with one integer label per axis (the dense payload of an IdxTensor): contractions and
    factorisations chain on the GPU without host round trips."""

    def __init__(self, data=None, labels=None, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        a, dims, lab = _labelled(data, labels)
        self._h = c_void_p()
        _check(_lib.t4a_gpu_tensor_new(_p(a), _p(dims) if a.ndim else None, _p(lab) if a.ndim else None, c_size_t(a.ndim),
                                       ctypes.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_tensor_release(h)
            self._h = None

    def _meta(self):
        r = c_size_t(0)
        _check(_lib.t4a_gpu_tensor_rank(self._h, ctypes.byref(r)))
        dims, labels = np.zeros(max(r.value, 1), dtype=np.uintp), np.zeros(max(r.value, 1), dtype=np.int64)
        _check(_lib.t4a_gpu_tensor_dims(self._h, _p(dims), _p(labels)))
        return [int(x) for x in dims[:r.value]], [int(x) for x in labels[:r.value]]

    @property
    def dims(self):
        return self._meta()[0]

    @property
    def labels(self):
        return self._meta()[1]

    def to_numpy(self):
        dims, _ = self._meta()
        out = np.zeros(int(np.prod(dims)) if dims else 1)
        _check(_lib.t4a_gpu_tensor_to_host(self._h, _p(out)))
        return out.reshape(dims, order="F")

    def permute(self, labels):
        lab = np.asarray(labels, dtype=np.int64)
        h = c_void_p()
        _check(_lib.t4a_gpu_tensor_permute(self._h, _p(lab) if len(lab) else None, ctypes.byref(h)))
        return LabelledTensor(_handle=h)

    def relabel(self, old, new):
        _check(_lib.t4a_gpu_tensor_relabel(self._h, ctypes.c_int64(old), ctypes.c_int64(new)))
        return self

    def contract(self, other):
        h = c_void_p()
        _check(_lib.t4a_gpu_tensor_contract(self._h, other._h, ctypes.byref(h)))
        return LabelledTensor(_handle=h)

    __mul__ = contract

    def outer_product(self, other):
        """outer_product (defaults/contract.rs:440): operands without a common label"""
        h = c_void_p()
        _check(_lib.t4a_gpu_tensor_outer_product(self._h, other._h, ctypes.byref(h)))
        return LabelledTensor(_handle=h)

    def tensordot(self, other, pairs):
        """tensordot (defaults/contract.rs:420): pairs = [(label of self, label of other), ...]"""
        la = np.asarray([p[0] for p in pairs] or [0], dtype=np.int64)
        lb = np.asarray([p[1] for p in pairs] or [0], dtype=np.int64)
        h = c_void_p()
        _check(_lib.t4a_gpu_tensor_tensordot(self._h, other._h, _p(la), _p(lb), c_size_t(len(pairs)), ctypes.byref(h)))
        return LabelledTensor(_handle=h)

    def svd(self, left, bond_label, bond_label_v=None, truncate=True, policy=None, max_bond_dim=None):
        lf = np.asarray(left, dtype=np.int64)
        pc = None if policy is None else policy.to_c()
        u, s, v = c_void_p(), c_void_p(), c_void_p()
        _check(_lib.t4a_gpu_tensor_svd(self._h, _p(lf) if len(lf) else None, c_size_t(len(lf)), c_int32(int(truncate)),
                                       None if pc is None else ctypes.byref(pc), c_int32(0 if max_bond_dim is None else 1),
                                       c_size_t(0 if max_bond_dim is None else max_bond_dim), ctypes.c_int64(bond_label),
                                       ctypes.c_int64(bond_label if bond_label_v is None else bond_label_v),
                                       ctypes.byref(u), ctypes.byref(s), ctypes.byref(v)))
        return LabelledTensor(_handle=u), LabelledTensor(_handle=s), LabelledTensor(_handle=v)

    def qr(self, left, bond_label, truncate=True, rtol=None):
        lf = np.asarray(left, dtype=np.int64)
        q, r = c_void_p(), c_void_p()
        _check(_lib.t4a_gpu_tensor_qr(self._h, _p(lf) if len(lf) else None, c_size_t(len(lf)), c_int32(int(truncate)),
                                      c_int32(0 if rtol is None else 1), c_double(0.0 if rtol is None else rtol),
                                      ctypes.c_int64(bond_label), ctypes.byref(q), ctypes.byref(r)))
        return LabelledTensor(_handle=q), LabelledTensor(_handle=r)


FACTORIZE_SVD, FACTORIZE_QR, FACTORIZE_LU, FACTORIZE_CI = 0, 1, 2, 3
CANONICAL_LEFT, CANONICAL_RIGHT = 0, 1


def _factorize(self, left, bond_label, alg=FACTORIZE_SVD, canonical=CANONICAL_LEFT, full_rank=False, policy=None,
               max_bond_dim=None, qr_rtol=None):
    """factorize / factorize_full_rank (core/src/defaults/factorize.rs:86): returns (left, right, rank, singular values or None)"""
    lf = np.asarray(left, dtype=np.int64)
    pc = None if policy is None else policy.to_c()
    hl, hr, rank = c_void_p(), c_void_p(), c_size_t(0)
    dims = self.dims
    sv = np.zeros(max(int(np.prod(dims)), 1))
    _check(_lib.t4a_gpu_tensor_factorize(self._h, _p(lf) if len(lf) else None, c_size_t(len(lf)), c_int32(alg), c_int32(canonical),
                                         c_int32(int(full_rank)), None if pc is None else ctypes.byref(pc),
                                         c_int32(0 if max_bond_dim is None else 1), c_size_t(0 if max_bond_dim is None else max_bond_dim),
                                         c_int32(0 if qr_rtol is None else 1), c_double(0.0 if qr_rtol is None else qr_rtol),
                                         ctypes.c_int64(bond_label), ctypes.byref(hl), ctypes.byref(hr), ctypes.byref(rank), _p(sv)))
    return (LabelledTensor(_handle=hl), LabelledTensor(_handle=hr), rank.value,
            sv[:rank.value].copy() if alg == FACTORIZE_SVD else None)


LabelledTensor.factorize = _factorize


# ---- tensor4all-aci: elementwise operations on tensor trains by alternating cross interpolation ----
class AciOptionsC(ctypes.Structure):
    _fields_ = [("max_iters", c_size_t), ("min_iters", c_size_t), ("has_max_bond_dim", c_int32), ("max_bond_dim", c_size_t),
                ("tolerance", c_double), ("scale_tolerance", c_int32), ("rng_seed", ctypes.c_uint64),
                ("enable_global_guard", c_int32), ("nsearch_global_pivots", c_size_t), ("max_nglobal_pivot", c_size_t),
                ("nsweeps_global_search", c_size_t), ("tol_margin_global_search", c_double)]


class AciOptions:
    """AciOptions (crates/tensor4all-aci/src/options.rs:37-168), same defaults.  initial_guess: SimpleTensorTrain, list of
    cores or None (random guess; the library's stream is not the reference's ChaCha8 one)."""

    def __init__(self, max_iters=20, min_iters=2, max_bond_dim=None, tolerance=1e-12, scale_tolerance=True, initial_guess=None,
                 rng_seed=0, enable_global_guard=True, nsearch_global_pivots=5, max_nglobal_pivot=5, nsweeps_global_search=100,
                 tol_margin_global_search=10.0):
        self.max_iters, self.min_iters, self.max_bond_dim, self.tolerance = max_iters, min_iters, max_bond_dim, tolerance
        self.scale_tolerance, self.initial_guess, self.rng_seed = scale_tolerance, initial_guess, rng_seed
        self.enable_global_guard, self.nsearch_global_pivots = enable_global_guard, nsearch_global_pivots
        self.max_nglobal_pivot, self.nsweeps_global_search = max_nglobal_pivot, nsweeps_global_search
        self.tol_margin_global_search = tol_margin_global_search

    def to_c(self):
        return AciOptionsC(self.max_iters, self.min_iters, 0 if self.max_bond_dim is None else 1, self.max_bond_dim or 0,
                           self.tolerance, int(self.scale_tolerance), self.rng_seed, int(self.enable_global_guard),
                           self.nsearch_global_pivots, self.max_nglobal_pivot, self.nsweeps_global_search,
                           self.tol_margin_global_search)


ACI_OP_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, ctypes.POINTER(c_double), c_size_t, c_size_t, ctypes.POINTER(c_double))
ACI_CALLBACK, ACI_PRODUCT, ACI_SUM = 0, 1, 2
ACI_CONVERGED, ACI_RANK_LIMITED, ACI_MAX_ITERATIONS = 0, 1, 2


def _aci_op(op):
    """ACI_PRODUCT / ACI_SUM (fused on the device) or a callable values (n_inputs, n_points) -> n_points values"""
    if not callable(op):
        return int(op), ACI_OP_FN(), None
    err = []

    def tramp(user, values, n_inputs, n_points, out):
        try:
            v = np.ctypeslib.as_array(values, shape=(n_points * n_inputs,)).reshape((n_inputs, n_points), order="F")
            np.ctypeslib.as_array(out, shape=(n_points,))[:] = op(v)
            return 0
        except Exception as e:  # noqa: BLE001
            err.append(e)
            return 1
    cb = ACI_OP_FN(tramp)
    return ACI_CALLBACK, cb, (cb, err)


class TreeAciLocalUpdate:
    """LocalUpdateResult of tensor4all-treeaci (local_update.rs:21-33): selected candidates as indices into the candidate lists"""


def treeaci_local_update(row_frames, col_frames, op=None, max_bond_dim=None, tolerance=1e-12, scale_tolerance=True,
                         left_orthogonal=True):
    """materialize_and_factor_edge (tensor4all-treeaci/src/local_update.rs:35) from the candidate frames on.  row_frames / col_frames:
    one array per input, shape (bond_k, row_count) / (bond_k, col_count) — column k = the frame vector of candidate k; op: ACI_PRODUCT
    (default, hadamard_many), ACI_SUM, or a callable values (n_inputs, n_points) -> n_points values."""
    if op is None:
        op = ACI_PRODUCT
    K = len(row_frames)
    if K != len(col_frames):
        raise ValueError("one row-frame and one column-frame block per input")
    rfs = [np.asfortranarray(np.asarray(f, dtype=np.float64)) for f in row_frames]
    cfs = [np.asfortranarray(np.asarray(f, dtype=np.float64)) for f in col_frames]
    row_count = rfs[0].shape[1] if K else 0
    col_count = cfs[0].shape[1] if K else 0
    for r, c in zip(rfs, cfs):
        if r.ndim != 2 or c.ndim != 2 or r.shape[0] != c.shape[0] or r.shape[1] != row_count or c.shape[1] != col_count:
            raise ValueError("opposite input frames have different cut bond dimensions or candidate counts")
    bd = (c_size_t * max(K, 1))(*[r.shape[0] for r in rfs])
    rp = (c_void_p * max(K, 1))(*[r.ctypes.data for r in rfs])
    cp = (c_void_p * max(K, 1))(*[c.ctypes.data for c in cfs])
    kind, cb, keep = _aci_op(op)
    cap = max(min(row_count, col_count), 1)
    rank, npe = c_size_t(0), c_size_t(0)
    rows, cols = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint64)
    perr = np.zeros(cap + 1)
    left, right = np.zeros(max(row_count * cap, 1)), np.zeros(max(cap * col_count, 1))
    scale = c_double(0.0)
    local = np.zeros(max(row_count * col_count, 1))
    st = _lib.t4a_gpu_treeaci_local_update_f64(c_size_t(K), bd, rp, cp, c_size_t(row_count), c_size_t(col_count), c_int32(kind), cb, None,
                                               c_size_t(0 if max_bond_dim is None else int(max_bond_dim)), c_double(tolerance),
                                               c_int32(1 if scale_tolerance else 0), c_int32(1 if left_orthogonal else 0),
                                               ctypes.byref(rank), _p(rows), _p(cols), _p(perr), ctypes.byref(npe), _p(left), _p(right),
                                               ctypes.byref(scale), _p(local))
    if keep is not None and keep[1]:
        raise keep[1][0]
    _check(st)
    r = TreeAciLocalUpdate()
    k = int(rank.value)
    r.rank = k
    r.row_indices = [int(v) for v in rows[:k]]
    r.col_indices = [int(v) for v in cols[:k]]
    r.pivot_errors = perr[:int(npe.value)].copy()
    r.left = left[:row_count * k].reshape((row_count, k), order="F").copy()
    r.right = right[:k * col_count].reshape((k, col_count), order="F").copy()
    r.sampled_scale = float(scale.value)
    r.row_count, r.col_count = row_count, col_count
    r.local_values = local[:row_count * col_count].copy()
    return r


def _aci_inputs(inputs, options):
    tts = [t if isinstance(t, SimpleTensorTrain) else SimpleTensorTrain(t) for t in inputs]
    arr = (c_void_p * max(len(tts), 1))(*[t._h for t in tts])
    g = options.initial_guess
    guess = None if g is None else (g if isinstance(g, SimpleTensorTrain) else SimpleTensorTrain(g))
    return tts, arr, guess


class AciResult:
    """AciResult (result.rs): tensor_train, ranks, errors, nglobal_pivots, termination"""


def elementwise_batched(op, inputs, options=None):
    """elementwise_batched (crates/tensor4all-aci/src/elementwise.rs:107)"""
    o = options or AciOptions()
    tts, arr, guess = _aci_inputs(inputs, o)
    kind, cb, keep = _aci_op(op)
    oc = o.to_c()
    cap = max(int(o.max_iters), 1)
    ranks, errs, ng = np.zeros(cap, dtype=np.uintp), np.zeros(cap), np.zeros(cap, dtype=np.uintp)
    h, n, term = c_void_p(), c_size_t(0), c_int32(0)
    _check(_lib.t4a_gpu_aci_elementwise(arr if tts else None, c_size_t(len(tts)), c_int32(kind), cb, None, ctypes.byref(oc),
                                        guess._h if guess else None, ctypes.byref(h), ctypes.byref(n), _p(ranks), _p(errs), _p(ng),
                                        ctypes.byref(term)))
    r = AciResult()
    r.tensor_train = SimpleTensorTrain._adopt(h)
    r.ranks, r.errors = [int(x) for x in ranks[:n.value]], [float(x) for x in errs[:n.value]]
    r.nglobal_pivots, r.termination = [int(x) for x in ng[:n.value]], term.value
    return r


def elementwise(op, inputs, options=None):
    """elementwise (elementwise.rs:311): scalar operator values[n_inputs] -> value"""
    if not callable(op):
        return elementwise_batched(op, inputs, options)
    return elementwise_batched(lambda v: np.array([op(v[:, p]) for p in range(v.shape[1])]), inputs, options)


class ElementwiseProblem:
    """ElementwiseProblem (state.rs:24-109) stepped bond by bond"""

    def __init__(self, op, inputs, options=None):
        o = options or AciOptions()
        self._tts, arr, self._guess = _aci_inputs(inputs, o)
        kind, cb, self._keep = _aci_op(op)
        oc = o.to_c()
        self._h = c_void_p()
        _check(_lib.t4a_gpu_aci_problem_new(arr if self._tts else None, c_size_t(len(self._tts)), c_int32(kind), cb, None,
                                            ctypes.byref(oc), self._guess._h if self._guess else None, ctypes.byref(self._h)))
        self.n_sites = len(self._tts[0])
        self.max_nglobal_pivot = o.max_nglobal_pivot

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.t4a_gpu_aci_problem_release(h)
            self._h = None

    def local_update(self, bond, left_orthogonal):
        _check(_lib.t4a_gpu_aci_problem_local_update(self._h, c_size_t(bond), c_int32(int(left_orthogonal))))

    def add_global_pivots(self, pivots):
        pv = np.asfortranarray(np.asarray(pivots, dtype=np.uintp).reshape(-1, self.n_sites).T)
        added = c_size_t(0)
        _check(_lib.t4a_gpu_aci_problem_add_global_pivots(self._h, _p(pv), c_size_t(pv.shape[1]), ctypes.byref(added)))
        return added.value

    def find_global_pivots(self, seed):
        out = np.zeros((self.n_sites, max(self.max_nglobal_pivot, 1)), dtype=np.uintp, order="F")
        count = c_size_t(0)
        _check(_lib.t4a_gpu_aci_problem_find_global_pivots(self._h, ctypes.c_uint64(seed), ctypes.byref(count), _p(out)))
        return [[int(x) for x in out[:, p]] for p in range(count.value)]

    def solution(self):
        h = c_void_p()
        _check(_lib.t4a_gpu_aci_problem_solution(self._h, ctypes.byref(h)))
        return SimpleTensorTrain._adopt(h)

    def frame(self, right, input, site):
        r, c = c_size_t(0), c_size_t(0)
        _check(_lib.t4a_gpu_aci_problem_frame(self._h, c_int32(int(right)), c_size_t(input), c_size_t(site), ctypes.byref(r),
                                              ctypes.byref(c), None))
        if r.value == 0:
            return None
        out = np.zeros(r.value * c.value)
        _check(_lib.t4a_gpu_aci_problem_frame(self._h, c_int32(int(right)), c_size_t(input), c_size_t(site), ctypes.byref(r),
                                              ctypes.byref(c), _p(out)))
        return out.reshape((r.value, c.value), order="F")

    def errors(self):
        e, sc = np.zeros(max(self.n_sites - 1, 1)), np.zeros(max(self.n_sites - 1, 1))
        _check(_lib.t4a_gpu_aci_problem_errors(self._h, _p(e), _p(sc)))
        return e[:self.n_sites - 1], sc[:self.n_sites - 1]


def tensor_train_to_tensors(tt, site_labels, bond_labels):
    """tensor_train_to_treetn_with_names_and_site_indices on a chain (treetn/src/simplett_bridge.rs:118, :706-794): one
    device-resident LabelledTensor per site with legs [bond s-1, site s, bond s] (boundary legs dropped)"""
    n = len(tt)
    sl, bl = np.asarray(site_labels, dtype=np.int64), np.asarray(bond_labels, dtype=np.int64)
    if len(sl) != n or len(bl) != max(n - 1, 0):
        raise T4aError(INVALID_ARGUMENT, "tensor_train_to_treetn: site_indices / bond count must match the tensor-train length")
    out = (c_void_p * max(n, 1))()
    _check(_lib.t4a_gpu_tt_to_tensors(tt._h, _p(sl) if n else None, _p(bl) if n > 1 else None, out))
    return [LabelledTensor(_handle=c_void_p(out[s])) for s in range(n)]


def tensors_to_tensor_train(tensors):
    """treetn_to_tensor_train for a chain of LabelledTensors (simplett_bridge.rs:172-277)"""
    arr = (c_void_p * max(len(tensors), 1))(*[t._h for t in tensors])
    h = c_void_p()
    _check(_lib.t4a_gpu_tensors_to_tt(arr if tensors else None, c_size_t(len(tensors)), ctypes.byref(h)))
    return SimpleTensorTrain._adopt(h)
