"""Host-side builders for the built-in function family of include/t4a_testfunctions.h.

A function is (fid, params[12], integer weight table [n_acc, sum(local_dims)]).  Pure numpy; used by both
the product bindings and the oracle bindings in tests/ (it only describes the workload)."""
import numpy as np

FN_MAX_PARAMS = 12
FN_QUANTICS_TRIG_EXP, FN_QUANTICS_OSC2D, FN_LORENTZ, FN_LINEAR = 0, 1, 2, 3


class FnSpec:
    def __init__(self, fid, params, weights, local_dims):
        self.fid = int(fid)
        p = np.zeros(FN_MAX_PARAMS, dtype=np.float64)
        p[: len(params)] = params
        self.params = p
        self.weights = np.ascontiguousarray(weights, dtype=np.uint64)
        self.n_acc = int(self.weights.shape[0])
        self.local_dims = [int(d) for d in local_dims]
        assert self.weights.shape[1] == sum(self.local_dims)

    def accumulators(self, idx):
        """idx: (n_pts, n_sites) -> (n_pts, n_acc) uint64 (wrap-around)."""
        idx = np.asarray(idx, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(self.local_dims)[:-1]])
        acc = np.zeros((idx.shape[0], self.n_acc), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for k in range(self.n_acc):
                acc[:, k] = self.weights[k][off[None, :] + idx].sum(axis=1, dtype=np.uint64)
        return acc


def _quantics_weights(n_sites, bit_of_site):
    """bit_of_site[s] = (acc index, bit weight exponent) for binary site s"""
    n_acc = 1 + max(a for a, _ in bit_of_site)
    w = np.zeros((n_acc, 2 * n_sites), dtype=np.uint64)
    for s, (a, e) in enumerate(bit_of_site):
        w[a, 2 * s + 1] = np.uint64(1) << np.uint64(e)
    return w


def quantics_trig_exp(n_bits, a=10.0, b=1.0, cc=1.0, cs=0.0):
    """(cc cos(a x) + cs sin(a x)) exp(-b x) on R = n_bits binary sites, MSB first
    (cfg2: cos(10x) exp(-x); regression test tensorci2/tests/mod.rs:728-768: sin(10x))."""
    w = _quantics_weights(n_bits, [(0, n_bits - 1 - s) for s in range(n_bits)])
    return FnSpec(FN_QUANTICS_TRIG_EXP, [a, b, cc, cs, n_bits], w, [2] * n_bits)


def quantics_osc2d(n_sites, k1=37, k2=53, k3=211, eps=0.1, k4=0, delta=0.0):
    """2-variable oscillatory integrand, interleaved bits (site 2k -> x, site 2k+1 -> y), MSB first."""
    assert n_sites % 2 == 0
    nb = n_sites // 2
    bits = []
    for s in range(n_sites):
        bits.append((s % 2, nb - 1 - s // 2))
    w = _quantics_weights(n_sites, bits)
    return FnSpec(FN_QUANTICS_OSC2D, [k1, k2, k3, eps, k4, delta, nb, nb], w, [2] * n_sites)


def lorentz(local_dims, coeff=1.0):
    """coeff / (sum_i v_i^2 + 1)  (tensorci2/tests/mod.rs:945-1002)."""
    tot = sum(local_dims)
    w = np.zeros((1, tot), dtype=np.uint64)
    o = 0
    for d in local_dims:
        for v in range(d):
            w[0, o + v] = v * v
        o += d
    return FnSpec(FN_LORENTZ, [coeff], w, local_dims)


def linear_sum(local_dims, scale=1.0, shift=0.0, site_weights=None):
    """scale * sum_i c_i * idx_i + shift (e.g. i + j, tensorci2/tests/mod.rs:397)."""
    tot = sum(local_dims)
    w = np.zeros((1, tot), dtype=np.uint64)
    o = 0
    for s, d in enumerate(local_dims):
        c = 1 if site_weights is None else int(site_weights[s])
        for v in range(d):
            w[0, o + v] = np.uint64(np.int64(c * v))
        o += d
    return FnSpec(FN_LINEAR, [scale, shift], w, local_dims)
