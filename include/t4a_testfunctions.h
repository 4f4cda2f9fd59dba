/*
 * t4a_testfunctions.h — the synthetic TCI2 workload family (BASELINE.json configs 2-5).
 *
 * This header DEFINES the user function `f` that the TCI2 sweep interpolates in the built-in
 * ("device functor") mode.  It is workload definition, not algorithm: the reference
 * (`crossinterpolate2`, tensor4all-tensorci/src/tensorci2.rs:1513) takes `f` as an opaque
 * closure.  The same inline code is compiled for the host (CPU oracle, gcc) and for gfx950
 * (hipcc), so that the candidate matrix Π is BIT-IDENTICAL on both sides — the precondition
 * for the bit-exact pivot contract (SURVEY.md §7 hard part (ii)).
 *
 * Determinism rules used below:
 *   - only IEEE-754 binary64 +, -, *, / and integer ops; no libm, no fused multiply-add
 *     (both compilers are invoked with -ffp-contract=off);
 *   - every multi-index is first folded into K (<= T4A_FN_MAX_ACC) 64-bit INTEGER accumulators
 *         acc[k] = sum_site  W[k][offset[site] + idx[site]]      (wrap-around uint64 arithmetic)
 *     which is associative, so a row-part + column-part split of the index (any bond) gives
 *     exactly the same integers as a left-to-right scan of the full index;
 *   - the scalar value is g_fid(acc, params) evaluated with the fixed operation order below.
 *
 * Quantics convention (reference test tensorci2/tests/mod.rs:735-742): site i of R binary
 * sites carries weight 2^(R-1-i), x = q / 2^R.
 */
#ifndef T4A_TESTFUNCTIONS_H
#define T4A_TESTFUNCTIONS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define T4A_HD __host__ __device__ inline
#else
#define T4A_HD static inline
#endif

#define T4A_FN_MAX_ACC 4
#define T4A_FN_MAX_PARAMS 12

/* function ids */
enum t4a_fn_id {
    /* (cc*cos(a x) + cs*sin(a x)) * exp(-b x),  x = acc0 / 2^nbits.
       params: [0]=a [1]=b [2]=cc [3]=cs [4]=nbits.       cfg2: a=10,b=1,cc=1,cs=0, nbits=20. */
    T4A_FN_QUANTICS_TRIG_EXP = 0,
    /* 2-variable oscillatory integrand (cfg3/4/5), x = acc0/2^nbx, y = acc1/2^nby:
         cos(2pi k1 x) cos(2pi k2 y) + eps * sin(2pi k3 (x+y)) / (1 + x^2 + y^2)
           + delta * cos(2pi k4 x y)
       params: [0]=k1 [1]=k2 [2]=k3 [3]=eps [4]=k4 [5]=delta [6]=nbx [7]=nby (k's integer valued) */
    T4A_FN_QUANTICS_OSC2D = 1,
    /* Lorentzian coeff / (acc0 + 1), acc0 = sum_i v_i^2 (reference test
       tensorci2/tests/mod.rs:945-1002). params: [0]=coeff */
    T4A_FN_LORENTZ = 2,
    /* acc0 interpreted as a signed integer, value = scale * acc0 + shift (linear functions
       such as i+j, tensorci2/tests/mod.rs:397). params: [0]=scale [1]=shift */
    T4A_FN_LINEAR = 3,
    T4A_FN_COUNT = 4
};

/* ---------- deterministic elementary functions (no libm, no FMA) ---------- */

/* round-to-nearest-even for |v| < 2^51 via the 2^52 trick: deterministic IEEE add/sub. */
T4A_HD double t4a_det_rint(double v)
{
    const double big = 4503599627370496.0; /* 2^52 */
    /* (no -ffast-math: neither compiler may re-associate these) */
    if (v >= 0.0) {
        double t = v + big;
        return t - big;
    } else {
        double t = v - big;
        return t + big;
    }
}

/* 2^n for -1022 <= n <= 1023 built from the exponent bits. */
T4A_HD double t4a_det_pow2(int n)
{
    union { uint64_t u; double d; } c;
    c.u = (uint64_t)(n + 1023) << 52;
    return c.d;
}

/* exp(x) for |x| <= 700: x = n ln2 + r, degree-13 Taylor in r (|r| <= 0.3466), Horner. */
T4A_HD double t4a_det_exp(double x)
{
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01; /* 32 significant bits */
    const double ln2_lo = 1.90821492927058770002e-10;
    double nd = t4a_det_rint(x * inv_ln2);
    int n = (int)nd;
    double r = (x - nd * ln2_hi) - nd * ln2_lo;
    double p = 1.0 / 6227020800.0; /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (n < -1000) return 0.0;
    return p * t4a_det_pow2(n);
}

/* sin and cos of theta for |theta| < ~1e6: Cody-Waite reduction by pi/2 in three parts,
   then the classic degree-13/14 minimax kernels on |r| <= pi/4. */
T4A_HD void t4a_det_sincos(double theta, double* s_out, double* c_out)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_2 = 6.07710050630396597660e-11; /* next 33 bits */
    const double pio2_3 = 2.02226624871116645580e-21; /* next 33 bits */
    double nd = t4a_det_rint(theta * two_over_pi);
    double r = ((theta - nd * pio2_1) - nd * pio2_2) - nd * pio2_3;
    long long n = (long long)nd;
    double z = r * r;
    /* sin kernel */
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    double sr = r + (r * z) * ps;
    /* cos kernel */
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    switch ((int)(n & 3)) {
    case 0: *s_out = sr;  *c_out = cr;  break;
    case 1: *s_out = cr;  *c_out = -sr; break;
    case 2: *s_out = -sr; *c_out = -cr; break;
    default: *s_out = -cr; *c_out = sr; break;
    }
}

/* sin/cos of 2*pi*(num / 2^nbits) with num already reduced mod 2^nbits (exact phase). */
T4A_HD void t4a_det_sincos_2pi_frac(uint64_t num, int nbits, double* s_out, double* c_out)
{
    const double two_pi = 6.28318530717958623200e+00;
    double u = (double)num * t4a_det_pow2(-nbits); /* exact for nbits <= 52 */
    t4a_det_sincos(u * two_pi, s_out, c_out);
}

/* ---------- the scalar stage g_fid(acc, params) ---------- */
T4A_HD double t4a_fn_value(int fid, const uint64_t* acc, const double* p)
{
    switch (fid) {
    case T4A_FN_QUANTICS_TRIG_EXP: {
        int nbits = (int)p[4];
        double x = (double)acc[0] * t4a_det_pow2(-nbits);
        double s, c;
        t4a_det_sincos(p[0] * x, &s, &c);
        double osc = p[2] * c + p[3] * s;
        double e = t4a_det_exp(-(p[1] * x));
        return osc * e;
    }
    case T4A_FN_QUANTICS_OSC2D: {
        int nbx = (int)p[6], nby = (int)p[7];
        uint64_t qx = acc[0], qy = acc[1];
        uint64_t mx = (nbx >= 64) ? ~0ull : ((1ull << nbx) - 1ull);
        uint64_t my = (nby >= 64) ? ~0ull : ((1ull << nby) - 1ull);
        uint64_t k1 = (uint64_t)p[0], k2 = (uint64_t)p[1], k3 = (uint64_t)p[2], k4 = (uint64_t)p[4];
        double x = (double)qx * t4a_det_pow2(-nbx);
        double y = (double)qy * t4a_det_pow2(-nby);
        double s1, c1, s2, c2;
        t4a_det_sincos_2pi_frac((k1 * qx) & mx, nbx, &s1, &c1);
        t4a_det_sincos_2pi_frac((k2 * qy) & my, nby, &s2, &c2);
        double v = c1 * c2;
        if (p[3] != 0.0) {
            /* sin(2pi k3 (x+y)) = sin(a)cos(b) + cos(a)sin(b) with exact phases a, b */
            double sa, ca, sb, cb;
            t4a_det_sincos_2pi_frac((k3 * qx) & mx, nbx, &sa, &ca);
            t4a_det_sincos_2pi_frac((k3 * qy) & my, nby, &sb, &cb);
            double num = sa * cb + ca * sb;
            double den = (1.0 + x * x) + y * y;
            v = v + p[3] * (num / den);
        }
        if (p[5] != 0.0) {
            /* cos(2pi k4 x y): phase k4*qx*qy / 2^(nbx+nby), reduced mod 2^(nbx+nby) */
            int nb = nbx + nby; /* callers keep nb <= 52 */
            uint64_t m = (nb >= 64) ? ~0ull : ((1ull << nb) - 1ull);
            double s4, c4;
            t4a_det_sincos_2pi_frac((k4 * qx * qy) & m, nb, &s4, &c4);
            v = v + p[5] * c4;
        }
        return v;
    }
    case T4A_FN_LORENTZ: {
        double sum_sq = (double)acc[0];
        return p[0] / (sum_sq + 1.0);
    }
    case T4A_FN_LINEAR: {
        double a = (double)(int64_t)acc[0];
        return p[0] * a + p[1];
    }
    default:
        return 0.0;
    }
}

/*
 * Host-side description of a built-in function: the integer weight table and parameters.
 *   n_sites, local_dims[site], offset[site] = sum_{s<site} local_dims[s]
 *   weights[k * total + offset[site] + v]  for v < local_dims[site], total = sum local_dims
 */
typedef struct t4a_fn_spec {
    int32_t fid;
    int32_t n_acc;                       /* K */
    double params[T4A_FN_MAX_PARAMS];
    /* weights are passed separately (length n_acc * sum(local_dims)) */
} t4a_fn_spec;

#endif /* T4A_TESTFUNCTIONS_H */
