// kernels_linalg.hip — thin SVD and thin QR for gfx950:
//   svd_backend (tensor4all-tensorbackend/src/backend.rs:709-731)  -> one-sided Jacobi (Hestenes), round-robin
//                                                                     pair ordering, one workgroup per column pair
//   qr_backend  (tensor4all-tensorbackend/src/backend.rs:742-760)  -> blocked Householder (compact WY): panel kernel + GEMMs
// The reference forwards both to tenferro-rs (faer); values are tolerance-level there (reconstruction 1e-10,
// backend/tests/mod.rs:58-110), so the contract here is: singular values non-increasing, U/V orthonormal,
// U diag(S) Vt == A and Q R == A to rounding.
#include "kernels.hpp"

#include <cstdlib>
#include <mutex>
#include <type_traits>

#include <atomic>
#include <cstdlib>

namespace t4a {

namespace {

// Wave sum through DPP row reductions (no LDS crossbar traffic): every lane of the wave ends with the total.
template <int CTRL> __device__ inline double dpp_mov_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ inline double wave_sum_dpp(double v)
{
    v += dpp_mov_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v); // row_half_mirror
    v += dpp_mov_f64<0x140>(v); // row_mirror: every lane of a row of 16 holds the row sum
    // rows -> wave: lanes 15, 31, 47, 63 hold the four row sums
    const long long b = __double_as_longlong(v);
    double t = 0.0;
#pragma unroll
    for (int l = 15; l < 64; l += 16) {
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        t += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return t;
}

// (round 5: the shuffle form — six __shfl_down = twelve ds_bpermute round trips + a broadcast — is gone: block_sum sits in front of
// every Householder column of the QR panel and of every rank count of the SVD's sort)
__device__ inline double wave_sum(double v) { return wave_sum_dpp(v); }

// Sum over the whole workgroup, identical on every thread.  `red` holds one slot per wave.
__device__ inline double block_sum(double v, double* red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];
    __syncthreads();
    return t;
}

// circle-method tournament: pair `k` of round `r` among np (even) players
__device__ inline void rr_pair(int np, int r, int k, int* a, int* b)
{
    const int q = np - 1;
    int x, y;
    if (k == 0) {
        x = q;
        y = r % q;
    } else {
        x = (r + k) % q;
        y = (r - k + q) % q;
    }
    *a = x < y ? x : y;
    *b = x < y ? y : x;
}

struct Rot {
    double c, s;
    int apply;
};
// A pair rotates when the cosine of the angle between its columns exceeds tol = sqrt(m) * eps — the stopping rule of LAPACK's one-sided
// Jacobi (dgesvj: "TOL = SQRT(M) * EPS").  Round 5: with tol = eps the last two sweeps of every decomposition found 1 - 200 of the 32 640
// pairs of a 256-column matrix to rotate (angles of a few eps); singular values and vectors agree to the same 1e-14 either way.
__device__ inline Rot jacobi_rotation(double alpha, double beta, double gamma, double tol)
{
    Rot r;
    r.c = 1.0;
    r.s = 0.0;
    r.apply = 0;
    if (gamma == 0.0 || !(fabs(gamma) > tol * sqrt(alpha * beta))) return r;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    // c = (1 + t^2)^(-1/2), |t| <= 1: the hardware estimate and two Newton steps (y <- y (3 - x y^2) / 2) instead of a square root and a
    // division in the longest dependent chain of a local round
    const double x = 1.0 + t * t;
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    r.c = y;
    r.s = r.c * t;
    r.apply = 1;
    return r;
}
__device__ inline double jacobi_tol(int m) { return 2.220446049250313e-16 * sqrt((double)m); }

// One tournament round, one workgroup per pair (large problems).
__global__ void __launch_bounds__(256) jacobi_round_kernel(double* W, int m, double* V, int n, int np, int round,
                                                           int* rotated)
{
    __shared__ double red[4];
    if (rotated[3]) return; // (an earlier sweep of this batch found nothing to rotate: jacobi_sweep_end_kernel)
    int i, j;
    rr_pair(np, round, blockIdx.x, &i, &j);
    if (j >= n) return;
    double* wi = W + (size_t)m * i;
    double* wj = W + (size_t)m * j;
    double a = 0.0, b = 0.0, g = 0.0;
    for (int r = threadIdx.x; r < m; r += blockDim.x) {
        const double x = wi[r], y = wj[r];
        a += x * x;
        b += y * y;
        g += x * y;
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    g = block_sum(g, red);
    const Rot rot = jacobi_rotation(a, b, g, jacobi_tol(m));
    if (!rot.apply) return;
    if (threadIdx.x == 0) *rotated = 1;
    for (int r = threadIdx.x; r < m; r += blockDim.x) {
        const double x = wi[r], y = wj[r];
        wi[r] = rot.c * x - rot.s * y;
        wj[r] = rot.s * x + rot.c * y;
    }
    double* vi = V + (size_t)n * i;
    double* vj = V + (size_t)n * j;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const double x = vi[r], y = vj[r];
        vi[r] = rot.c * x - rot.s * y;
        vj[r] = rot.s * x + rot.c * y;
    }
}

// Whole Jacobi iteration inside one workgroup (small problems): one wave per pair, all sweeps in-kernel,
// W and V staged through LDS when they fit.
__global__ void __launch_bounds__(1024) jacobi_small_kernel(double* Wg, int m, double* Vg, int n, int np,
                                                            int max_sweeps, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_rot;
    double* W = Wg;
    double* V = Vg;
    const int tid = threadIdx.x, T = blockDim.x;
    if (use_lds) {
        W = (double*)smem_raw;
        V = W + (size_t)m * n;
        for (int e = tid; e < m * n; e += T) W[e] = Wg[e];
        for (int e = tid; e < n * n; e += T) V[e] = Vg[e];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; ++round) {
            for (int k = wave; k < np / 2; k += nw) {
                int i, j;
                rr_pair(np, round, k, &i, &j);
                if (j >= n) continue;
                double* wi = W + (size_t)m * i;
                double* wj = W + (size_t)m * j;
                double a = 0.0, b = 0.0, g = 0.0;
                for (int r = lane; r < m; r += 64) {
                    const double x = wi[r], y = wj[r];
                    a += x * x;
                    b += y * y;
                    g += x * y;
                }
                a = wave_sum(a);
                b = wave_sum(b);
                g = wave_sum(g);
                const Rot rot = jacobi_rotation(a, b, g, jacobi_tol(m));
                if (!rot.apply) continue;
                if (lane == 0) s_rot = 1;
                for (int r = lane; r < m; r += 64) {
                    const double x = wi[r], y = wj[r];
                    wi[r] = rot.c * x - rot.s * y;
                    wj[r] = rot.s * x + rot.c * y;
                }
                double* vi = V + (size_t)n * i;
                double* vj = V + (size_t)n * j;
                for (int r = lane; r < n; r += 64) {
                    const double x = vi[r], y = vj[r];
                    vi[r] = rot.c * x - rot.s * y;
                    vj[r] = rot.s * x + rot.c * y;
                }
            }
            __syncthreads();
        }
        const int any = s_rot;
        __syncthreads();
        if (!any) break;
    }
    if (use_lds) {
        for (int e = tid; e < m * n; e += T) Wg[e] = W[e];
        for (int e = tid; e < n * n; e += T) Vg[e] = V[e];
    }
}

// Sum over an aligned group of G = 8 or 16 lanes (DPP inside a row of 16, no readlane): every lane of the group ends with bitwise the same
// total (the two operands of every addition are the same pair of partial sums in every lane, in either order).
template <int G> __device__ inline double group_sum(double v)
{
    v += dpp_mov_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v); // row_half_mirror: the sum of 8 lanes
    if (G == 16) v += dpp_mov_f64<0x140>(v); // row_mirror
    return v;
}

// jacobi_rotation with the divisions and square roots replaced by the hardware reciprocal / reciprocal-square-root estimates and Newton
// steps (the rotation arithmetic is the longest dependent chain of a round of jacobi_groups_kernel).  The ANGLE (zeta, t) is computed to
// ~2^-46: an error there leaves a residual inner product of that relative size, which the next sweep removes; the COSINE keeps both
// Newton steps (c^2 + s^2 = 1 to rounding: every rotation stays orthogonal).  Out-of-range intermediates (|zeta| > 1e154, subnormal
// gamma) surface as a NaN and mean "no rotation", as the threshold test does for an overflowing alpha * beta.
__device__ inline Rot jacobi_rotation_fast(double alpha, double beta, double gamma, double tol)
{
    Rot r;
    r.c = 1.0;
    r.s = 0.0;
    r.apply = 0;
    if (gamma == 0.0) return r;
    const double ab = alpha * beta;
    const double sq = ab == 0.0 ? 0.0 : ab * __builtin_amdgcn_rsq(ab); // sqrt(alpha beta) to 2^-23: a threshold
    if (!(fabs(gamma) > tol * sq)) return r;
    const double den = 2.0 * gamma;
    double rd = __builtin_amdgcn_rcp(den);
    rd = rd * (2.0 - den * rd);
    const double zeta = (beta - alpha) * rd;
    const double x = 1.0 + zeta * zeta;
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    const double d2 = fabs(zeta) + x * y; // |zeta| + sqrt(1 + zeta^2) >= 1
    double rt = __builtin_amdgcn_rcp(d2);
    rt = rt * (2.0 - d2 * rt);
    const double t = zeta >= 0.0 ? rt : -rt;
    const double x2 = 1.0 + t * t;
    double c = __builtin_amdgcn_rsq(x2);
    c = c * (1.5 - 0.5 * x2 * c * c);
    c = c * (1.5 - 0.5 * x2 * c * c);
    const double sn = c * t;
    if (!(c == c) || !(sn == sn)) return r;
    r.c = c;
    r.s = sn;
    r.apply = 1;
    return r;
}

// Whole Jacobi iteration inside one workgroup, a GROUP OF G = 16 LANES per column pair (round 6; matrices up to 96 columns whose W and V
// fit the LDS together).  jacobi_small_kernel gives a pair a whole wave: three 64-lane sums with readlanes and the scalar rotation
// arithmetic replicated over 64 lanes made a round of a 64 x 64 matrix 3.3 us (2.2 ms per decomposition, slower than the blocked
// tournament's 1.2 ms).  Here a wave carries four pairs: a lane owns the rows sub, sub + 16, ... of its pair's two columns (MR of W, VR
// of V, in registers from the dot products to the rotation: one LDS round trip per round), the three dot products are 16-lane DPP sums and
// the rotation arithmetic is executed once per wave.  Measured (profiles/r06_svd_small.txt sections 3 and 9): 0.85 us = 2 043 cycles per round
// at 64 x 64 (10 sweeps of 63 rounds in 536 us).  Hardware counters over that kernel: VALU issue 55 % of a SIMD's time plus 20 % for LDS
// instructions, the LDS array busy 43 % (880 cycles per round: all of W and V cross it twice), SQ_LDS_BANK_CONFLICT 0, the rest dependency
// stalls and barriers — bound by instruction issue of two waves per SIMD (three at 96 columns: 59 % + 27 %, LDS array 60 %; one at 32
// columns: 38 % + 10 %, half the time stalled).  Eight lanes per pair (half the waves, 1.7 x fewer instructions per round) measured
// SLOWER at every size (64 x 64: 776 against 727 us per call: with one wave per SIMD the stalls no longer overlap): G stays a parameter,
// only 16 is instantiated.
// V starts as the identity in the LDS (never read from memory); a non-finite input sets *nonfinite and leaves W untouched.
// LDS: double W[np][ldw], double V[np][ldv]: the launcher picks the instantiation whose column lengths cover m and n, the padding rows
// are zero and stay zero under rotations, so no loop carries a bound.
template <int G, int MR, int VR>
__global__ void __launch_bounds__(768) jacobi_groups_kernel(double* Wg, int m, double* Vg, int n, int np, int max_sweeps, int* nonfinite)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_rot[3];
    __shared__ int s_bad;
    __shared__ unsigned long long s_amax;
    // a column's stride in the LDS: G MR (G VR) rows, m <= G MR, n <= G VR, the padding rows are zero.  (No bank staggering: a group reads
    // 16 consecutive doubles and a ds_read_b64 only conflicts inside a group of 16 lanes — SQ_LDS_BANK_CONFLICT is 0 for every instantiation;
    // an earlier version padded the strides for nothing.)
    constexpr int ldw = G * MR, ldv = G * VR;
    double* W = (double*)smem_raw;
    double* V = W + (size_t)np * ldw;
    const int tid = threadIdx.x, T = blockDim.x;
    if (tid == 0) {
        s_rot[0] = s_rot[1] = s_rot[2] = 0;
        s_bad = 0;
        s_amax = 0ull;
    }
    __syncthreads();
    int bad = 0;
    double mx = 0.0;
    for (int e = tid; e < np * ldw; e += T) {
        const int c = e / ldw, r = e - c * ldw;
        double x = 0.0;
        if (c < n && r < m) {
            x = Wg[(size_t)c * m + r];
            const double ax = fabs(x);
            if (!(ax <= 1.79769313486231570e308)) bad = 1;
            else if (ax > mx) mx = ax;
        }
        W[e] = x;
    }
    for (int e = tid; e < np * ldv; e += T) {
        const int c = e / ldv, r = e - c * ldv;
        V[e] = (c == r && c < n) ? 1.0 : 0.0;
    }
    if (bad) s_bad = 1;
    if (mx > 0.0) atomicMax(&s_amax, (unsigned long long)__double_as_longlong(mx));
    __syncthreads();
    if (s_bad) {
        if (tid == 0) *nonfinite = 1;
        return;
    }
    // a matrix whose largest entry is far from 1 is scaled by a power of two (exact) for the iteration: alpha * beta in the pair test overflows
    // from |a| ~ 1e77 on (see nonfinite_absmax_kernel).  W goes back SCALED (the column norms of the finalisation would overflow the same
    // way) and nonfinite[2] tells the caller the exponent to put back on the singular values; matrices in the usual range are not touched
    int scale_e = 0;
    {
        const double amax = __longlong_as_double((long long)s_amax);
        scale_e = pow2_scale_exponent(amax);
    }
    if (tid == 0) nonfinite[2] = scale_e;
    if (scale_e != 0) {
        for (int e = tid; e < np * ldw; e += T) W[e] = ldexp(W[e], -scale_e);
        __syncthreads();
    }
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int PW = 64 / G;              // pairs per wave
    const int sub = lane & (G - 1);
    const int k = wave * PW + lane / G;      // the pair slot of this group of lanes
    const int q = np - 1;
    const bool slot = k < np / 2;
    const double tol = jacobi_tol(m);
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // three flags in rotation: sweep s sets [s % 3], reads it behind its last barrier, and clears [(s + 1) % 3] for the next sweep
        // (a thread still reading [s % 3] can be overtaken by the next sweep's first round, never by the sweep after that)
        if (tid == 0) s_rot[(sweep + 1) % 3] = 0;
        int* rotated = &s_rot[sweep % 3];
        for (int round = 0; round < q; ++round) {
            // circle method (rr_pair) without the divisions: round < q and k < q
            int x, y;
            if (k == 0) {
                x = q;
                y = round;
            } else {
                x = round + k;
                if (x >= q) x -= q;
                y = round - k;
                if (y < 0) y += q;
            }
            const int i = x < y ? x : y, j = x < y ? y : x;
            if (slot && j < n) {
                double* wi = W + (size_t)i * ldw + sub;
                double* wj = W + (size_t)j * ldw + sub;
                double* vi = V + (size_t)i * ldv + sub;
                double* vj = V + (size_t)j * ldv + sub;
                double xv[MR], yv[MR], vx[VR], vy[VR];
#pragma unroll
                for (int t = 0; t < MR; ++t) {
                    xv[t] = wi[G * t];
                    yv[t] = wj[G * t];
                }
#pragma unroll
                for (int t = 0; t < VR; ++t) {
                    vx[t] = vi[G * t];
                    vy[t] = vj[G * t];
                }
                double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
                for (int t = 0; t < MR; ++t) {
                    a += xv[t] * xv[t];
                    b += yv[t] * yv[t];
                    g += xv[t] * yv[t];
                }
                a = group_sum<G>(a);
                b = group_sum<G>(b);
                g = group_sum<G>(g);
                const Rot rot = jacobi_rotation_fast(a, b, g, tol);
                if (rot.apply) {
                    if (sub == 0) *rotated = 1;
#pragma unroll
                    for (int t = 0; t < MR; ++t) {
                        wi[G * t] = rot.c * xv[t] - rot.s * yv[t];
                        wj[G * t] = rot.s * xv[t] + rot.c * yv[t];
                    }
#pragma unroll
                    for (int t = 0; t < VR; ++t) {
                        vi[G * t] = rot.c * vx[t] - rot.s * vy[t];
                        vj[G * t] = rot.s * vx[t] + rot.c * vy[t];
                    }
                }
            }
            __syncthreads();
        }
        if (!*rotated) {
            if (tid == 0) nonfinite[1] = sweep + 1; // (flags[3]: the sweep count, for T4A_SVD_DEBUG)
            break;
        }
    }
    for (int e = tid; e < n * m; e += T) {
        const int c = e / m, r = e - c * m;
        Wg[e] = W[(size_t)c * ldw + r]; // (still scaled by 2^-scale_e: the caller scales the singular values back)
    }
    for (int e = tid; e < n * n; e += T) {
        const int c = e / n, r = e - c * n;
        Vg[e] = V[(size_t)c * ldv + r];
    }
}

// Block round of the blocked one-sided Jacobi (round 4): the n columns are cut into blocks of `w`; a sweep is a tournament over
// the BLOCKS (nbp - 1 launches instead of n - 1), and the workgroup of a block pair (I, J) brings its 2 w columns into the LDS
// and runs a whole local tournament over them there: 2 w - 1 local rounds of w pairs, one WAVE per pair, dot products by
// wavefront shuffles, one barrier per local round.  The rotations of the block are accumulated in a 2w x 2w matrix Q (LDS) and
// applied to the rows of V in one pass at the end (V never enters the LDS).
// LDS: double cols[2 w][m], double Q[2 w][2 w], int idx[2 w].
typedef double jb_double2 __attribute__((ext_vector_type(2)));
typedef double jb_double4 __attribute__((ext_vector_type(4)));

template <int BW, int JB_RMAX = 16>
__global__ void __launch_bounds__(BW * 64 < 256 ? 256 : BW * 64) jacobi_block_kernel(double* __restrict__ W, int m, double* __restrict__ V, int n, int nbp, int round,
                                                                                     int* rotated)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int w = BW, w2 = 2 * BW;
    static_assert(JB_RMAX % 2 == 0, "the register window holds pairs of rows");
    // a column in the LDS: ms doubles (m rounded up to even: a lane owns PAIRS of consecutive rows — rows 2 lane + e + 128 qq — and moves
    // them with 16-byte LDS operations; the padding row of an odd m is zero and stays zero under every rotation)
    const int ms = (m + 1) & ~1;
    double* const cols = reinterpret_cast<double*>(smem_raw);
    double* const Q = cols + (size_t)w2 * ms;
    int* const idx = reinterpret_cast<int*>(Q + (size_t)w2 * w2);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
#ifdef T4A_JB_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define JSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define JSTAMP(k) do {} while (0)
#endif
    if (rotated[3]) return; // (an earlier sweep of this batch found nothing to rotate: jacobi_sweep_end_kernel)
    int bi, bj;
    rr_pair(nbp, round, blockIdx.x, &bi, &bj);
    // global column of local column c (a block beyond the matrix, the tail of the last block: no column)
    auto gcol = [&](int c) {
        const int g = c < w ? bi * w + c : bj * w + (c - w);
        return g < n ? g : -1;
    };
    if (tid < w2) idx[tid] = gcol(tid);
    for (int e = tid; e < w2 * w2; e += T) Q[e] = (e / w2 == e % w2) ? 1.0 : 0.0;
    {
        // the block's columns into the LDS: every wave requests all elements of its columns that fit the register window BEFORE it stores
        // the first one (as a load -> store loop the compiler waits for every element in turn: 2 w m / T dependent memory round trips
        // at the head of every block round)
        constexpr int NWV = (BW * 64 < 256 ? 256 : BW * 64) / 64, CPW = (w2 + NWV - 1) / NWV;
        double tmp[CPW][JB_RMAX];
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int c = wave + NWV * k;
            const int gc = c < w2 ? gcol(c) : -1;
#pragma unroll
            for (int q = 0; q < JB_RMAX; ++q) {
                const int r = 2 * lane + (q & 1) + 128 * (q >> 1);
                tmp[k][q] = (gc >= 0 && r < m) ? W[(size_t)m * gc + r] : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int c = wave + NWV * k;
            if (c < w2) {
#pragma unroll
                for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
                    const int r = 2 * lane + 128 * qq;
                    if (r < ms) *reinterpret_cast<jb_double2*>(cols + (size_t)c * ms + r) = (jb_double2){tmp[k][2 * qq], tmp[k][2 * qq + 1]};
                }
                const int gc = gcol(c);
                for (int r = lane + 64 * JB_RMAX; r < ms; r += 64) cols[(size_t)c * ms + r] = (gc >= 0 && r < m) ? W[(size_t)m * gc + r] : 0.0;
            }
        }
    }
    __syncthreads();
    JSTAMP(0);
    bool any = false;
    const double tol = jacobi_tol(m);
    // every column pair once per sweep: the first block round runs the full local tournament (pairs inside the two blocks and
    // across them), the others only the w^2 pairs ACROSS the blocks (w local rounds: column k of I with column (k + lr) mod w of J)
    // — a pair inside a block that was met in every block round kept rotating on rounding noise and the sweeps never ended.
    // In those rounds wave k keeps column k of I in its registers from the first local round to the last (nobody else touches it): one
    // column read and, when the pair rotates, one column written per local round — the LDS moved 64 KB per local round at m = 256, a
    // quarter of the round's time, with both columns going through it.
    const bool resident = round != 0;
    const int n_local = resident ? w : w2 - 1;
    jb_double2 xr[JB_RMAX / 2];
    const bool a_valid = wave < w && gcol(wave) >= 0;
    if (resident && a_valid) {
#pragma unroll
        for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
            const int r = 2 * lane + 128 * qq;
            xr[qq] = r < ms ? *reinterpret_cast<const jb_double2*>(cols + (size_t)wave * ms + r) : (jb_double2){0.0, 0.0};
        }
    }
    for (int lr = 0; lr < n_local; ++lr) {
        if (wave < w) {
            int a, b;
            if (!resident) {
                rr_pair(w2, lr, wave, &a, &b);
            } else {
                a = wave;
                b = w + (wave + lr) % w;
            }
            if (gcol(a) >= 0 && gcol(b) >= 0) {
                double* const ca = cols + (size_t)a * ms;
                double* const cb = cols + (size_t)b * ms;
                // the two columns stay in registers between the dot products and the rotation; columns longer than 64 * JB_RMAX rows
                // re-read their tail.  JB_RMAX is sized for the column length (round 5: with a fixed 16 a 256-row column — every matrix
                // behind the QR preconditioner at chi = 256 — executed four times the loads, selects and multiply-adds it needed)
                jb_double2 yr[JB_RMAX / 2];
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
                    const int r = 2 * lane + 128 * qq;
                    const bool in = r < ms;
                    if (!resident) xr[qq] = in ? *reinterpret_cast<const jb_double2*>(ca + r) : (jb_double2){0.0, 0.0};
                    yr[qq] = in ? *reinterpret_cast<const jb_double2*>(cb + r) : (jb_double2){0.0, 0.0};
                }
#pragma unroll
                for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
                    al += xr[qq].x * xr[qq].x;
                    be += yr[qq].x * yr[qq].x;
                    ga += xr[qq].x * yr[qq].x;
                    al += xr[qq].y * xr[qq].y;
                    be += yr[qq].y * yr[qq].y;
                    ga += xr[qq].y * yr[qq].y;
                }
                for (int r = lane + 64 * JB_RMAX; r < m; r += 64) {
                    const double x = ca[r], y = cb[r];
                    al += x * x;
                    be += y * y;
                    ga += x * y;
                }
                JSTAMP(1);
                al = wave_sum_dpp(al);
                be = wave_sum_dpp(be);
                ga = wave_sum_dpp(ga);
                JSTAMP(2);
                const Rot rot = jacobi_rotation(al, be, ga, tol);
                JSTAMP(3);
                if (rot.apply) {
                    any = true;
#pragma unroll
                    for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
                        const int r = 2 * lane + 128 * qq;
                        const jb_double2 xn = (jb_double2){rot.c * xr[qq].x - rot.s * yr[qq].x, rot.c * xr[qq].y - rot.s * yr[qq].y};
                        const jb_double2 yn = (jb_double2){rot.s * xr[qq].x + rot.c * yr[qq].x, rot.s * xr[qq].y + rot.c * yr[qq].y};
                        if (r < ms) {
                            if (!resident) *reinterpret_cast<jb_double2*>(ca + r) = xn;
                            *reinterpret_cast<jb_double2*>(cb + r) = yn;
                        }
                        xr[qq] = xn;
                    }
                    for (int r = lane + 64 * JB_RMAX; r < m; r += 64) {
                        const double x = ca[r], y = cb[r];
                        ca[r] = rot.c * x - rot.s * y;
                        cb[r] = rot.s * x + rot.c * y;
                    }
                    JSTAMP(4);
                    if (lane < w2) { // column a / b of Q (Q[row][col] at row * w2 + col)
                        const double x = Q[lane * w2 + a], y = Q[lane * w2 + b];
                        Q[lane * w2 + a] = rot.c * x - rot.s * y;
                        Q[lane * w2 + b] = rot.s * x + rot.c * y;
                    }
                    JSTAMP(5);
                }
            }
        }
        __syncthreads();
        JSTAMP(6);
    }
    if (resident && a_valid) { // (the last barrier of the loop is behind every read of this column's tail; the write-out below reads it)
#pragma unroll
        for (int qq = 0; qq < JB_RMAX / 2; ++qq) {
            const int r = 2 * lane + 128 * qq;
            if (r < ms) *reinterpret_cast<jb_double2*>(cols + (size_t)wave * ms + r) = xr[qq];
        }
    }
    if (resident) __syncthreads();
    if (any && lane == 0) *rotated = 1;
    // a block pair in which no pair rotated (most of them in the last two sweeps of every decomposition) leaves W and V as they are:
    // nothing to write back, Q is the identity
    if (!__syncthreads_or(any ? 1 : 0)) return;
    for (int c = wave; c < w2; c += (T >> 6)) {
        const int gc = idx[c];
        if (gc < 0) continue;
        for (int r = lane; r < m; r += 64) W[(size_t)m * gc + r] = cols[(size_t)c * ms + r];
    }
    JSTAMP(7);
    if constexpr (w2 % 16 == 0) {
        // V(:, block) <- V(:, block) Q on the f64 matrix cores: a wave per tile of 16 rows, out[r][c] = sum_k V[r][k] Q[k][c] as
        // v_mfma_f64_16x16x4 with "A" = Q^T (rows = c) and "B" = V^T (columns = r); lane (lr = lane & 15, lk = lane >> 4) holds
        // out[r0 + lr][ct * 16 + lk + 4 reg].  (Round 5: one row per thread with 2 w x 2 w serial multiply-adds took 13 000 of the 44 600
        // cycles of a block round at w = 8, half the workgroup idle.)
        constexpr int KS = w2 / 4, CT = w2 / 16;
        const int lr = lane & 15, lk = lane >> 4;
        double qf[CT][KS];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ct][ks] = Q[(4 * ks + lk) * w2 + ct * 16 + lr];
        int gk[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gk[ks] = idx[4 * ks + lk];
        int go[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) go[ct][reg] = idx[ct * 16 + lk + 4 * reg];
        for (int r0 = 16 * wave; r0 < n; r0 += 16 * (T >> 6)) {
            const int r = r0 + lr;
            double vf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[ks] = (gk[ks] >= 0 && r < n) ? V[(size_t)n * gk[ks] + r] : 0.0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                jb_double4 acc = (jb_double4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[ct][ks], vf[ks], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (go[ct][reg] >= 0 && r < n) V[(size_t)n * go[ct][reg] + r] = acc[reg];
            }
        }
    } else {
        // V(:, block) <- V(:, block) Q: one row of V per thread
        for (int r = tid; r < n; r += T) {
            double vin[w2];
#pragma unroll
            for (int c = 0; c < w2; ++c) vin[c] = idx[c] >= 0 ? V[(size_t)n * idx[c] + r] : 0.0;
#pragma unroll
            for (int c = 0; c < w2; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < w2; ++k) acc += vin[k] * Q[k * w2 + c];
                if (idx[c] >= 0) V[(size_t)n * idx[c] + r] = acc;
            }
        }
    }
#ifdef T4A_JB_STAMPS
    JSTAMP(8);
    if (blockIdx.x == 1 && lane == 0 && (wave == 0 || wave == 3) && round == 3 && *rotated)
        printf("[jacobi_block stamps] w=%d m=%d n=%d wave %d cycles: load=%llu | lds+dots=%llu sums=%llu rotation=%llu apply=%llu Q=%llu barrier=%llu | writeback=%llu V=%llu\n", w, m, n,
               wave, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6], st_acc[7], st_acc[8]);
#endif
}

__global__ void __launch_bounds__(256) col_norms_kernel(const double* __restrict__ W, int m, int n, double* sig)
{
    __shared__ double red[4];
    const int j = blockIdx.x;
    const double* w = W + (size_t)m * j;
    double a = 0.0;
    for (int r = threadIdx.x; r < m; r += blockDim.x) a += w[r] * w[r];
    a = block_sum(a, red);
    if (threadIdx.x == 0) sig[j] = sqrt(a);
}

// Sort by (sigma descending, index ascending) with a counting rank, normalise U, gather V.
__global__ void __launch_bounds__(256) svd_sort_scatter_kernel(const double* __restrict__ W, int m, const double* V,
                                                               int n, const double* __restrict__ sig, double* U,
                                                               double* S, double* Vs, int* dead, int* n_dead)
{
    __shared__ double red[4];
    const int j = blockIdx.x;
    const double sj = sig[j];
    double cnt = 0.0, smax = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double si = sig[i];
        if (si > sj || (si == sj && i < j)) cnt += 1.0;
        smax = si > smax ? si : smax;
    }
    const int rank = (int)block_sum(cnt, red);
    {
        __shared__ double smax_s;
        if (threadIdx.x == 0) smax_s = 0.0;
        __syncthreads();
        // (non-negative doubles order like their bit patterns)
        atomicMax(reinterpret_cast<unsigned long long*>(&smax_s), (unsigned long long)__double_as_longlong(smax));
        __syncthreads();
        smax = smax_s;
    }
    // a column whose squared norm underflows (|w| <= sqrt(DBL_MIN)) cannot be orthogonalised by the rotations above — their test and
    // their angles are built from products of squared norms — and its direction is rounding residue (an exactly zero column of A comes
    // out of the QR preconditioner with entries around 1e-153): it counts as a zero column, its singular value is reported as it is
    const bool live = sj > 1.4916681462400413e-154 && sj > smax * 1e-100; // (... or lies a hundred decades under the largest one)
    if (threadIdx.x == 0) {
        S[rank] = sj;
        dead[rank] = live ? 0 : 1;
        if (!live) atomicAdd(n_dead, 1);
    }
    const double* w = W + (size_t)m * j;
    double* u = U + (size_t)m * rank;
    for (int r = threadIdx.x; r < m; r += blockDim.x) u[r] = live ? w[r] / sj : 0.0;
    const double* v = V + (size_t)n * j;
    double* vs = Vs + (size_t)n * rank;
    for (int r = threadIdx.x; r < n; r += blockDim.x) vs[r] = v[r];
}

// Replace the dead (sigma == 0) columns of U by unit vectors orthogonal to all live ones (rare path).
__global__ void __launch_bounds__(256) svd_complete_kernel(double* U, int m, int n, int* dead, double* tmp)
{
    __shared__ double red[4];
    __shared__ int s_best;
    __shared__ double sb[256];
    __shared__ int si[256];
    for (int j = 0; j < n; ++j) {
        if (!dead[j]) continue;
        // row with the largest residual 1 - sum_k u(i,k)^2 (first maximum)
        double best = -2.0;
        int besti = m;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            double s = 0.0;
            for (int k = 0; k < n; ++k)
                if (!dead[k]) s += U[i + (size_t)m * k] * U[i + (size_t)m * k];
            const double res = 1.0 - s;
            if (res > best) {
                best = res;
                besti = i;
            }
        }
        // serial combine through LDS (small, rare)
        sb[threadIdx.x] = best;
        si[threadIdx.x] = besti;
        __syncthreads();
        if (threadIdx.x == 0) {
            double bb = -2.0;
            int bi = 0;
            for (int t = 0; t < (int)blockDim.x; ++t)
                if (sb[t] > bb || (sb[t] == bb && si[t] < bi)) {
                    bb = sb[t];
                    bi = si[t];
                }
            s_best = bi;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) tmp[i] = (i == s_best) ? 1.0 : 0.0;
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < n; ++k) {
                if (dead[k]) continue;
                double d = 0.0;
                for (int i = threadIdx.x; i < m; i += blockDim.x) d += U[i + (size_t)m * k] * tmp[i];
                d = block_sum(d, red);
                for (int i = threadIdx.x; i < m; i += blockDim.x) tmp[i] -= d * U[i + (size_t)m * k];
                __syncthreads();
            }
        double nn = 0.0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) nn += tmp[i] * tmp[i];
        nn = sqrt(block_sum(nn, red));
        for (int i = threadIdx.x; i < m; i += blockDim.x) U[i + (size_t)m * j] = nn > 0.0 ? tmp[i] / nn : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) dead[j] = 0;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) jacobi_sweep_end_kernel(int* flags)
{
    if (threadIdx.x == 0) {
        if (flags[3] == 0) {
            if (flags[0] == 0) flags[3] = 1;
            else flags[0] = 0;
        }
    }
}

__global__ void __launch_bounds__(256) nonfinite_kernel(const double* __restrict__ data, size_t count, int* flag)
{
    int bad = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        const double v = data[e];
        if (!(fabs(v) <= 1.79769313486231570e308)) bad = 1;
    }
    if (bad) *flag = 1;
}

// Inf / NaN flag and the largest magnitude in one pass (*absmax_bits zeroed by the caller: the bit pattern of a non-negative double orders
// like its value).  Engine::svd scales a matrix whose largest entry is far from 1 by a power of two (exact) before the iteration: the
// pair test of the Jacobi rotations forms alpha * beta, the product of two squared column norms, which overflows from |a| ~ 1e77 on —
// no pair rotated and the factors came back non-orthogonal without an error (found by tools/soak_svd_small.py, round 6).
__global__ void __launch_bounds__(256) nonfinite_absmax_kernel(const double* __restrict__ data, size_t count, int* flag,
                                                               unsigned long long* absmax_bits)
{
    __shared__ unsigned long long s_max;
    if (threadIdx.x == 0) s_max = 0ull;
    __syncthreads();
    int bad = 0;
    double mx = 0.0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        const double v = fabs(data[e]);
        if (!(v <= 1.79769313486231570e308)) bad = 1;
        else if (v > mx) mx = v;
    }
    if (bad) *flag = 1;
    if (mx > 0.0) atomicMax(&s_max, (unsigned long long)__double_as_longlong(mx));
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(absmax_bits, s_max);
}

// dst = src * 2^(sign * e), e from the largest magnitude a previous nonfinite_absmax_kernel left in *absmax_bits (no host round trip: the
// QR has none); e = 0 is a plain copy, and nothing at all when dst == src.
__global__ void __launch_bounds__(256) scale_pow2_dev_kernel(double* dst, const double* src, size_t count, const unsigned long long* absmax_bits,
                                                             int sign)
{
    const int e = sign * pow2_scale_exponent(__longlong_as_double((long long)*absmax_bits));
    if (e == 0 && dst == src) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = e != 0 ? ldexp(src[i], e) : src[i];
}

__global__ void __launch_bounds__(256) scale_pow2_kernel(double* dst, const double* src, size_t count, int e)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = ldexp(src[i], e);
}

// ------------------------------------------------------------------------------------------------ QR
// Blocked Householder QR, round 3 (qr_backend, tensor4all-tensorbackend/src/backend.rs:742-760; callers
// tensor4all-simplett/src/compression.rs:229-341).  Same reflectors as qr_step_kernel (alpha = -sign(x0) |x|, v = x - alpha e1
// unnormalised, tau = 2 / v.v), but QR_NB columns at a time: one workgroup of 16 waves factors the panel — per column one
// workgroup reduction for the norm, then every wave applies the reflector to "its" remaining panel columns and builds its
// entries of the compact-WY factor T by wave shuffles (no LDS round per column pair) — and the trailing matrix gets
// (I - V T V^T)^T in three GEMMs on the f64 matrix cores.  512 x 256: 8 panels x 4 launches instead of 512 launches.
constexpr int QR_NB = 32;
constexpr int QR_T = 1024;

__global__ void __launch_bounds__(QR_T) qr_panel_kernel(double* A, int m, int j0, int w, double* diag, double* tau, double* v0s,
                                                        double* Vall, double* Tp)
{
    __shared__ double red[QR_T / 64];
    __shared__ double zs[QR_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = QR_T / 64;
    for (int e = tid; e < QR_NB * QR_NB; e += QR_T) Tp[e] = 0.0;
    for (int jj = 0; jj < w; ++jj) {
        const int j = j0 + jj;
        double* x = A + (size_t)m * j;
        double* V = Vall + (size_t)m * j;
        // reflector of column j (rows j .. m-1)
        double tail = 0.0;
        for (int i = j + 1 + tid; i < m; i += QR_T) tail += x[i] * x[i];
        tail = block_sum(tail, red);
        const double x0 = x[j];
        const double nrm = sqrt(x0 * x0 + tail);
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        const double vv = v0 * v0 + tail;
        const double t = (nrm == 0.0) ? 0.0 : 2.0 / vv;
        if (tid == 0) {
            diag[j] = (nrm == 0.0) ? 0.0 : alpha;
            tau[j] = t;
            v0s[j] = v0;
        }
        for (int i = tid; i < m; i += QR_T) V[i] = i < j ? 0.0 : (i == j ? v0 : x[i]);
        __syncthreads(); // V column visible to the whole workgroup
        if (t != 0.0) {
            // H_j on the remaining columns of the panel: one wave per column
            for (int c = j + 1 + wave; c < j0 + w; c += NWV) {
                double* y = A + (size_t)m * c;
                double dot = 0.0;
                for (int i = j + lane; i < m; i += 64) dot += V[i] * y[i];
                dot = wave_sum(dot);
                const double f = t * dot;
                for (int i = j + lane; i < m; i += 64) y[i] -= f * V[i];
            }
            // compact WY: T(0:jj, jj) = -tau T(0:jj, 0:jj) (V(:, 0:jj)^T v_jj), T(jj, jj) = tau
            for (int q = wave; q < jj; q += NWV) {
                const double* Vq = Vall + (size_t)m * (j0 + q);
                double dot = 0.0;
                for (int i = j + lane; i < m; i += 64) dot += Vq[i] * V[i];
                dot = wave_sum(dot);
                if (lane == 0) zs[q] = dot;
            }
        }
        __syncthreads();
        if (tid < jj && t != 0.0) {
            double acc = 0.0;
            for (int r = tid; r < jj; ++r) acc += Tp[tid + QR_NB * r] * zs[r];
            Tp[tid + QR_NB * jj] = -t * acc;
        }
        if (tid == 0) Tp[jj + QR_NB * jj] = t;
        __syncthreads();
    }
}

// The same panel factorisation with the panel RESIDENT IN THE LDS (rows j0 .. m-1 of the panel's w <= 32 columns: up to 560 rows =
// 143 KiB of the 160): the global-memory version above pays four to five dependent L2 round trips per column (norm pass, reflector
// write, rank-1 update of the panel, T column: 7.8 us per column at 512 rows), here a column is two workgroup reductions and one
// round of wave tasks — task c > jj: H_jj on panel column c (dot product by DPP wave sum, update in place), task q < jj: the dot
// product V_q^T v_jj for the compact-WY factor.  A (R above the diagonal, the reflector tails below), Vall (explicit reflectors), diag,
// tau, v0s and T leave the kernel exactly as qr_panel_kernel leaves them.
constexpr int QR_LDS_MAX_ROWS = 560;
__global__ void __launch_bounds__(QR_T) qr_panel_lds_kernel(double* A, int m, int j0, int w, double* diag, double* tau, double* v0s, double* Vall, double* Tp)
{
    extern __shared__ __attribute__((aligned(16))) double qsm[]; // [QR_NB][ld] panel, [QR_NB * QR_NB] T, [QR_NB] z, [QR_T / 64] red
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = QR_T / 64;
    const int mp = m - j0;                 // rows of the panel
    const int ld = mp | 1;                 // (odd leading dimension: the lanes of a wave task walk down a column, tasks sit ld apart)
    double* const P = qsm;
    double* const T = P + (size_t)QR_NB * ld;
    double* const zs = T + QR_NB * QR_NB;
    double* const red = zs + QR_NB;
    for (int e = tid; e < QR_NB * QR_NB; e += QR_T) T[e] = 0.0;
    for (int c = wave; c < w; c += NWV) {
        // (all rows of a column requested before the first one is stored: up to QR_LDS_MAX_ROWS / 64 = 9 per lane)
        const double* src = A + (size_t)m * (j0 + c) + j0;
        double tmp[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) tmp[q] = (lane + 64 * q) < mp ? src[lane + 64 * q] : 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if ((lane + 64 * q) < mp) P[(size_t)c * ld + lane + 64 * q] = tmp[q];
        for (int r = lane + 64 * 9; r < mp; r += 64) P[(size_t)c * ld + r] = src[r];
    }
    __syncthreads();
    for (int jj = 0; jj < w; ++jj) {
        double* x = P + (size_t)jj * ld; // rows jj .. mp-1 hold the column below (and on) the diagonal
        double tail = 0.0;
        for (int r = jj + 1 + tid; r < mp; r += QR_T) tail += x[r] * x[r];
        tail = block_sum(tail, red);
        const double x0 = x[jj];
        const double nrm = sqrt(x0 * x0 + tail);
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        const double vv = v0 * v0 + tail;
        const double t = (nrm == 0.0) ? 0.0 : 2.0 / vv;
        __syncthreads(); // (every thread has read x[jj] before it becomes v0)
        if (tid == 0) {
            diag[j0 + jj] = (nrm == 0.0) ? 0.0 : alpha;
            tau[j0 + jj] = t;
            v0s[j0 + jj] = v0;
            x[jj] = v0; // the reflector sits in the column from row jj on (R's diagonal entry lives in diag[])
        }
        __syncthreads();
        if (t != 0.0) {
            // wave tasks: columns jj+1 .. w-1 get H_jj; columns 0 .. jj-1 give z_q = V_q^T v_jj (rows >= jj: v_jj is zero above)
            for (int task = wave; task < w - 1; task += NWV) {
                if (task >= jj) {
                    double* y = P + (size_t)(task + 1) * ld;
                    double dot = 0.0;
                    for (int r = jj + lane; r < mp; r += 64) dot += x[r] * y[r];
                    dot = wave_sum_dpp(dot);
                    const double f = t * dot;
                    for (int r = jj + lane; r < mp; r += 64) y[r] -= f * x[r];
                } else {
                    const double* vq = P + (size_t)task * ld;
                    double dot = 0.0;
                    for (int r = jj + lane; r < mp; r += 64) dot += vq[r] * x[r];
                    dot = wave_sum_dpp(dot);
                    if (lane == 0) zs[task] = dot;
                }
            }
        }
        __syncthreads();
        if (tid < jj && t != 0.0) { // compact WY: T(0:jj, jj) = -tau T(0:jj, 0:jj) z
            double acc = 0.0;
            for (int r = tid; r < jj; ++r) acc += T[tid + QR_NB * r] * zs[r];
            T[tid + QR_NB * jj] = -t * acc;
        }
        if (tid == 0) T[jj + QR_NB * jj] = t;
        __syncthreads();
    }
    // write-out: A keeps R above the diagonal and the reflector tails below it (its diagonal entry is not read by anybody: R's
    // diagonal is diag[]); Vall gets the explicit reflectors, zero above the diagonal
    for (int c = wave; c < w; c += NWV) {
        double* dstA = A + (size_t)m * (j0 + c) + j0;
        double* dstV = Vall + (size_t)m * (j0 + c);
        const double* col = P + (size_t)c * ld;
        for (int r = lane; r < mp; r += 64) {
            const double v = col[r];
            if (r != c) dstA[r] = v;
            dstV[j0 + r] = r < c ? 0.0 : v;
        }
        for (int r = lane; r < j0; r += 64) dstV[r] = 0.0;
    }
    for (int e = tid; e < QR_NB * QR_NB; e += QR_T) Tp[e] = T[e];
}

// Column c of the compact-WY factor in the LDS: T(0:c, c) <- -tau_c T(0:c, 0:c) z with z = T(0:c, c) as the wave tasks left it (raw
// products V_q^T v_c); one wave, lane = row, every lane reads its terms before any lane writes (program order of one wave).
__device__ __forceinline__ void qr_t_column(double* T, int c, int lane)
{
    const double tc = T[c + QR_NB * c];
    if (lane < c && tc != 0.0) {
        double acc = 0.0;
        for (int r = lane; r < c; ++r) acc += T[lane + QR_NB * r] * T[r + QR_NB * c];
        T[lane + QR_NB * c] = -tc * acc;
    }
}

// Round 5, second version of the LDS-resident panel.  The first one spent 3.9 us per column (125 us per 32-column panel, 1.0 of the
// 1.7 ms of a 512 x 256 factorisation) in a chain of five barriers: norm of the column by the whole workgroup (two barriers), the
// diagonal entry replaced by v0 (two barriers), the wave tasks, the column of T (one barrier each).  Here a column is ONE barrier:
//   * the norm of column jj + 1 is computed by the wave that has just applied H_jj to it (its updated values are in registers): the next
//     iteration starts from two numbers in the LDS instead of a workgroup reduction;
//   * the diagonal entry stays in place, v0 lives in its own array and the lane that meets row jj substitutes it;
//   * a wave runs its two tasks (columns wave and wave + 16 of the 31 others) together: loads, dot products and the two DPP sums interleave;
//   * the raw products z_q = V_q^T v_jj go straight into the strict upper part of T and wave 15 — the one wave with a single task — turns
//     column jj - 1 into -tau T z while the others work on column jj.
// Same reflectors, same T (its column sums in the same order), A / Vall / diag / tau / v0s as qr_panel_kernel leaves them; the norms are
// summed in a different order (one wave instead of sixteen), i.e. equal to rounding.
template <int RPL> // rows per lane: ceil((m - j0) / 64) — the workgroup is bound by the instructions its 16 waves issue on one compute unit
__global__ void __launch_bounds__(QR_T) qr_panel_lds2_kernel(double* A, int m, int j0, int w, double* diag, double* tau, double* v0s, double* Vall, double* Tp)
{
    extern __shared__ __attribute__((aligned(16))) double qsm[]; // [QR_NB][ld] panel, [QR_NB * QR_NB] T, [QR_NB] v0, [QR_T / 64] red, [2] next, [QR_NB] R's diagonal
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = QR_T / 64;
    static_assert(RPL >= 1 && RPL * 64 >= 64 && RPL <= (QR_LDS_MAX_ROWS + 63) / 64, "rows per lane");
    const int mp = m - j0;
    const int ld = mp | 1;
    double* const P = qsm;
    double* const T = P + (size_t)QR_NB * ld;
    double* const v0l = T + QR_NB * QR_NB;
    double* const red = v0l + QR_NB;
    double* const nxt = red + NWV;
    double* const dgl = nxt + 2;
    for (int e = tid; e < QR_NB * QR_NB; e += QR_T) T[e] = 0.0;
    for (int c = wave; c < w; c += NWV) {
        const double* src = A + (size_t)m * (j0 + c) + j0;
        double tmp[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) tmp[q] = (lane + 64 * q) < mp ? src[lane + 64 * q] : 0.0;
#pragma unroll
        for (int q = 0; q < RPL; ++q)
            if ((lane + 64 * q) < mp) P[(size_t)c * ld + lane + 64 * q] = tmp[q];
    }
    __syncthreads();
    double tail = 0.0;
    for (int r = 1 + tid; r < mp; r += QR_T) tail += P[r] * P[r];
    tail = block_sum(tail, red);
    double x0 = P[0];
    for (int jj = 0; jj < w; ++jj) {
        const double* x = P + (size_t)jj * ld;
        const double nrm = sqrt(x0 * x0 + tail);
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        const double vv = v0 * v0 + tail;
        const double t = (nrm == 0.0) ? 0.0 : 2.0 / vv;
        if (tid == 0) { // (into the LDS: a store to global memory here is waited for — vmcnt(0) — in front of the column's barrier)
            dgl[jj] = (nrm == 0.0) ? 0.0 : alpha;
            v0l[jj] = v0;
            T[jj + QR_NB * jj] = t;
        }
        // the wave's two tasks: task k < jj reads column k (z_k = V_k^T v_jj), task k >= jj updates column k + 1 with H_jj
        const int ta = wave, tb = wave + NWV;
        const bool has_a = ta < w - 1, has_b = tb < w - 1;
        const int ca = ta < jj ? ta : ta + 1, cb = tb < jj ? tb : tb + 1;
        const bool next_a = has_a && ta == jj, next_b = has_b && tb == jj; // (the column that comes next: its norm is taken here)
        if ((t != 0.0 && has_a) || next_a || next_b) {
            double* const ya = P + (size_t)ca * ld;
            double* const yb = P + (size_t)(has_b ? cb : ca) * ld;
            double xv[RPL], va[RPL], vb[RPL];
            double da = 0.0, db = 0.0;
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int r = jj + lane + 64 * q;
                const bool in = r < mp;
                xv[q] = in ? x[r] : 0.0;
                va[q] = in ? ya[r] : 0.0;
                vb[q] = (in && has_b) ? yb[r] : 0.0;
            }
            if (lane == 0) xv[0] = v0; // (row jj: the reflector's head, the LDS still holds the column's own entry there)
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                da += xv[q] * va[q];
                db += xv[q] * vb[q];
            }
            da = wave_sum_dpp(da);
            db = wave_sum_dpp(db);
            if (t != 0.0) {
                if (ta >= jj) {
                    const double f = t * da;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        const int r = jj + lane + 64 * q;
                        va[q] -= f * xv[q];
                        if (r < mp) ya[r] = va[q];
                    }
                } else if (lane == 0) {
                    T[ta + QR_NB * jj] = da;
                }
                if (has_b) {
                    if (tb >= jj) {
                        const double f = t * db;
#pragma unroll
                        for (int q = 0; q < RPL; ++q) {
                            const int r = jj + lane + 64 * q;
                            vb[q] -= f * xv[q];
                            if (r < mp) yb[r] = vb[q];
                        }
                    } else if (lane == 0) {
                        T[tb + QR_NB * jj] = db;
                    }
                }
            }
            if (next_a || next_b) { // column jj + 1 as it stands now: its diagonal entry (row jj + 1) and the squared norm below it
                double sq = 0.0;
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const double v = next_a ? va[q] : vb[q];
                    if (lane + 64 * q >= 2) sq += v * v;
                }
                sq = wave_sum_dpp(sq);
                const double head = next_a ? va[0] : vb[0];
                const double h1 = __shfl(head, 1);
                if (lane == 0) {
                    nxt[0] = sq;
                    nxt[1] = h1;
                }
            }
        }
        if (wave == NWV - 1 && jj >= 1) { // column jj - 1 of T: -tau T(0:c, 0:c) z, z = the raw products stored in that column
            qr_t_column(T, jj - 1, lane);
        }
        __syncthreads();
        tail = nxt[0];
        x0 = nxt[1];
    }
    if (wave == 0 && w >= 2) { // the last column of T
        qr_t_column(T, w - 1, lane);
    }
    __syncthreads();
    for (int c = wave; c < w; c += NWV) {
        double* dstA = A + (size_t)m * (j0 + c) + j0;
        double* dstV = Vall + (size_t)m * (j0 + c);
        const double* col = P + (size_t)c * ld;
        const double vc = v0l[c];
        for (int r = lane; r < mp; r += 64) {
            const double v = col[r];
            if (r != c) dstA[r] = v;
            dstV[j0 + r] = r < c ? 0.0 : (r == c ? vc : v);
        }
        for (int r = lane; r < j0; r += 64) dstV[r] = 0.0;
    }
    for (int e = tid; e < QR_NB * QR_NB; e += QR_T) Tp[e] = T[e];
    if (tid < w) {
        diag[j0 + tid] = dgl[tid];
        tau[j0 + tid] = T[tid + QR_NB * tid];
        v0s[j0 + tid] = v0l[tid];
    }
}

__global__ void __launch_bounds__(256) qr_extract_r_kernel(const double* __restrict__ A, int m, int n, int k,
                                                           const double* diag, double* R)
{
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += blockDim.x)
        R[i + (size_t)k * j] = i < j ? A[i + (size_t)m * j] : (i == j ? diag[i] : 0.0);
}

} // namespace

void nonfinite_flag_launch(const double* data, size_t count, int* d_flag, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, data, count, d_flag);
}

void nonfinite_absmax_launch(const double* data, size_t count, int* d_flag, unsigned long long* d_absmax_bits, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(nonfinite_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, data, count, d_flag, d_absmax_bits);
}

// dst[i] = src[i] * 2^e (exact; dst may be src)
void scale_pow2_launch(double* dst, const double* src, size_t count, int e, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(scale_pow2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, count, e);
}

void scale_pow2_dev_launch(double* dst, const double* src, size_t count, const unsigned long long* d_absmax_bits, int sign, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(scale_pow2_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, count, d_absmax_bits, sign);
}

bool jacobi_fits_small(int m, int n) { return n <= 128 && m <= 2048; }

void jacobi_small_launch(double* W, int m, double* V, int n, int max_sweeps, hipStream_t stream)
{
    static std::atomic<bool> attr_set{false}; // (launches come from several host threads; setting the attribute twice is harmless)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_small_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        attr_set = true;
    }
    const int np = n + (n & 1);
    const size_t bytes = ((size_t)m * n + (size_t)n * n) * 8;
    const int use_lds = bytes <= 144 * 1024 ? 1 : 0;
    int T = 64 * (np / 2 > 0 ? np / 2 : 1);
    if (T > 1024) T = 1024;
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(1), dim3(T), use_lds ? bytes : 0, stream, W, m, V, n, np, max_sweeps,
                       use_lds);
}

// One workgroup, sixteen lanes per column pair (jacobi_groups_kernel): W (m x n, m >= n) and V together in the LDS.
// The instantiation <16, MR, VR> for a shape: V columns of 32 rows for n <= 32, of 64 for n <= 64, of 96 for n <= 96; W columns of
// 32 / 64 / 128 / 224 rows (n <= 64) or 96 rows (n <= 96: all the LDS holds).
namespace {
struct JgPlan {
    int MR, VR;
};
bool jg_plan(int m, int n, JgPlan* p)
{
    if (n < 2 || m < n) return false;
    if (n <= 32) {
        *p = JgPlan{m <= 32 ? 2 : (m <= 64 ? 4 : (m <= 128 ? 8 : 14)), 2};
        return m <= 224;
    }
    if (n <= 64) {
        *p = JgPlan{m <= 64 ? 4 : (m <= 128 ? 8 : 14), 4};
        return m <= 224;
    }
    *p = JgPlan{6, 6};
    return n <= 96 && m <= 96;
}
} // namespace
bool jacobi_fits_groups(int m, int n)
{
    JgPlan p;
    return jg_plan(m, n, &p);
}

// V is an OUTPUT only (the kernel starts from the identity); *d_nonfinite is set (and W left as it was) when W holds an Inf or a NaN;
// d_nonfinite[1] receives the number of sweeps, d_nonfinite[2] the exponent e when the kernel iterated on 2^-e W (W comes back that way).  false: the shape has no instantiation (jacobi_fits_groups), nothing was launched.
bool jacobi_groups_launch(double* W, int m, double* V, int n, int max_sweeps, int* d_nonfinite, hipStream_t stream)
{
    JgPlan p;
    if (!jg_plan(m, n, &p)) return false;
    const int np = n + (n & 1);
    const int waves = (np / 2 + 3) / 4;
    const size_t lds = (size_t)np * 16 * (p.MR + p.VR) * 8; // <= 144 KB by construction
    // the dynamic-LDS limit of an instantiation is raised once per process (launches come from several host threads: setting it twice is
    // harmless, the flag only saves the runtime call)
    static std::atomic<bool> attr_set[8];
    auto go = [&](auto kern, int id) {
        if (!attr_set[id].load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
            attr_set[id].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), lds, stream, W, m, V, n, np, max_sweeps, d_nonfinite);
    };
    switch (p.VR * 100 + p.MR) {
    case 202: go(&jacobi_groups_kernel<16, 2, 2>, 0); break;
    case 204: go(&jacobi_groups_kernel<16, 4, 2>, 1); break;
    case 208: go(&jacobi_groups_kernel<16, 8, 2>, 2); break;
    case 214: go(&jacobi_groups_kernel<16, 14, 2>, 3); break;
    case 404: go(&jacobi_groups_kernel<16, 4, 4>, 4); break;
    case 408: go(&jacobi_groups_kernel<16, 8, 4>, 5); break;
    case 414: go(&jacobi_groups_kernel<16, 14, 4>, 6); break;
    case 606: go(&jacobi_groups_kernel<16, 6, 6>, 7); break;
    default: return false;
    }
    return true;
}

// flags: [0] a pair rotated in the running sweep [3] converged.  Behind every sweep: a sweep without a rotation sets [3] — every kernel of a
// later sweep of the same batch returns at once —, otherwise [0] is cleared for the next one.  The host looks at the flags once per BATCH of
// sweeps instead of once per sweep (a copy and a stream synchronisation: ~50 us each, eleven of them in a 256-column decomposition).
void jacobi_sweep_end_launch(int* d_flags, hipStream_t stream)
{
    hipLaunchKernelGGL(jacobi_sweep_end_kernel, dim3(1), dim3(64), 0, stream, d_flags);
}

void jacobi_sweep_launch(double* W, int m, double* V, int n, int* d_rotated, hipStream_t stream)
{
    const int np = n + (n & 1);
    for (int round = 0; round < np - 1; ++round)
        hipLaunchKernelGGL(jacobi_round_kernel, dim3(np / 2), dim3(256), 0, stream, W, m, V, n, np, round, d_rotated);
}

// Blocked sweep: block width by the rows that fit the LDS next to Q (2 w columns of m doubles), one launch per block round.
// Returns false when the columns are too long for the LDS (m > 4096): the caller keeps the launch-per-round path.
bool jacobi_block_sweep_launch(double* W, int m, double* V, int n, int* d_rotated, hipStream_t stream)
{
    static const int w_max = diag_env("T4A_SVD_BLOCK_W") ? std::atoi(diag_env("T4A_SVD_BLOCK_W")) : 8;
    int w = w_max >= 16 ? 16 : (w_max >= 8 ? 8 : (w_max >= 4 ? 4 : 2));
    const int ms = (m + 1) & ~1; // (a column's stride in the LDS: jacobi_block_kernel)
    while (w > 1 && (size_t)2 * w * ms * 8 > (size_t)136 * 1024) w >>= 1;
    if ((size_t)2 * w * ms * 8 > (size_t)136 * 1024) return false;
    const int nb = (n + w - 1) / w;
    const int nbp = nb < 2 ? 2 : nb + (nb & 1);
    const size_t lds = ((size_t)2 * w * ms + (size_t)4 * w * w) * 8 + (size_t)2 * w * 4 + 16;
    int T = 64 * w;
    if (T < 256) T = 256; // (the extra waves only help with the loads and the V pass)
    static std::once_flag attr_once; // (once per process, not once per sweep: ADVICE round 4; every block width in one go)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    });
    auto go = [&](auto kern) {
        for (int round = 0; round < nbp - 1; ++round)
            hipLaunchKernelGGL(kern, dim3(nbp / 2), dim3(T), lds, stream, W, m, V, n, nbp, round, d_rotated);
    };
    static std::once_flag attr_once2;
    std::call_once(attr_once2, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_block_kernel<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    });
    switch (w) {
    case 16:
        if (m <= 128) go(&jacobi_block_kernel<16, 2>);
        else if (m <= 256) go(&jacobi_block_kernel<16, 4>);
        else if (m <= 512) go(&jacobi_block_kernel<16, 8>);
        else go(&jacobi_block_kernel<16>);
        break;
    case 8:
        if (m <= 128) go(&jacobi_block_kernel<8, 2>);
        else if (m <= 256) go(&jacobi_block_kernel<8, 4>);
        else if (m <= 512) go(&jacobi_block_kernel<8, 8>);
        else go(&jacobi_block_kernel<8>);
        break;
    case 4: go(&jacobi_block_kernel<4>); break;
    case 2: go(&jacobi_block_kernel<2>); break;
    default: go(&jacobi_block_kernel<1>); break;
    }
    return true;
}

void svd_finalize_launch(const double* W, int m, const double* V, int n, double* sig_tmp, double* U, double* S,
                         double* Vs, int* d_dead, int* d_ndead, hipStream_t stream)
{
    hipLaunchKernelGGL(col_norms_kernel, dim3(n), dim3(256), 0, stream, W, m, n, sig_tmp);
    hipLaunchKernelGGL(svd_sort_scatter_kernel, dim3(n), dim3(256), 0, stream, W, m, V, n, sig_tmp, U, S, Vs, d_dead,
                       d_ndead);
}

void svd_complete_launch(double* U, int m, int n, int* d_dead, double* tmp_m, hipStream_t stream)
{
    hipLaunchKernelGGL(svd_complete_kernel, dim3(1), dim3(256), 0, stream, U, m, n, d_dead, tmp_m);
}

// Blocked factorisation in place (reflectors below the diagonal as before, R above it, its diagonal in diag[]); Vall (m x k)
// receives the explicit reflector matrix, Tall (ceil(k / QR_NB) blocks of QR_NB x QR_NB) the compact-WY factors, W / W2 are
// QR_NB x max(n, k) scratch.
void qr_factor_launch(double* A, int m, int n, double* diag, double* tau, double* v0s, double* Vall, double* Tall, double* W, double* W2,
                      hipStream_t stream)
{
    const int k = m < n ? m : n;
    for (int j0 = 0, pi = 0; j0 < k; j0 += QR_NB, ++pi) {
        const int w = (k - j0) < QR_NB ? (k - j0) : QR_NB;
        double* Tp = Tall + (size_t)pi * QR_NB * QR_NB;
        static const bool no_lds_panel = diag_env("T4A_QR_NO_LDS_PANEL") != nullptr;
        if (!no_lds_panel && m - j0 <= QR_LDS_MAX_ROWS) {
            const int ld = (m - j0) | 1;
            const size_t lds = ((size_t)QR_NB * ld + QR_NB * QR_NB + QR_NB + QR_T / 64 + 2 + QR_NB) * sizeof(double);
            static std::once_flag attr_once;
            std::call_once(attr_once, [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qr_panel_lds2_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            });
            static const bool old_panel = diag_env("T4A_QR_OLD_PANEL") != nullptr; // (the five-barrier version, for comparison)
            const int rpl = (m - j0 + 63) / 64;
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(1), dim3(QR_T), lds, stream, A, m, j0, w, diag, tau, v0s, Vall, Tp); };
            if (old_panel) go(&qr_panel_lds_kernel);
            else if (rpl <= 1) go(&qr_panel_lds2_kernel<1>);
            else if (rpl <= 2) go(&qr_panel_lds2_kernel<2>);
            else if (rpl <= 3) go(&qr_panel_lds2_kernel<3>);
            else if (rpl <= 4) go(&qr_panel_lds2_kernel<4>);
            else if (rpl <= 6) go(&qr_panel_lds2_kernel<6>);
            else go(&qr_panel_lds2_kernel<9>);
        } else {
            hipLaunchKernelGGL(qr_panel_kernel, dim3(1), dim3(QR_T), 0, stream, A, m, j0, w, diag, tau, v0s, Vall, Tp);
        }
        const int n2 = n - j0 - w, mp = m - j0;
        if (n2 <= 0) continue;
        const double* Vp = Vall + j0 + (size_t)m * j0; // rows j0 .., columns j0 .. j0 + w - 1 (zero above)
        double* A2 = A + j0 + (size_t)m * (j0 + w);
        GemmDesc g;
        g.batch = 1;
        g.strideA = g.strideB = g.strideC = 0;
        // W = Vp^T A2   (w x n2)
        g.m = w; g.n = n2; g.k = mp;
        g.A = Vp; g.lda = m; g.transA = 1;
        g.B = A2; g.ldb = m; g.transB = 0;
        g.C = W; g.ldc = QR_NB; g.alpha = 1.0; g.beta = 0.0;
        gemm_launch(g, stream);
        // W2 = T^T W
        g.m = w; g.n = n2; g.k = w;
        g.A = Tp; g.lda = QR_NB; g.transA = 1;
        g.B = W; g.ldb = QR_NB; g.transB = 0;
        g.C = W2; g.ldc = QR_NB; g.alpha = 1.0; g.beta = 0.0;
        gemm_launch(g, stream);
        // A2 -= Vp W2
        g.m = mp; g.n = n2; g.k = w;
        g.A = Vp; g.lda = m; g.transA = 0;
        g.B = W2; g.ldb = QR_NB; g.transB = 0;
        g.C = A2; g.ldc = m; g.alpha = -1.0; g.beta = 1.0;
        gemm_launch(g, stream);
    }
}

void qr_form_launch(const double* A, int m, int n, const double* diag, const double* Vall, const double* Tall, double* W, double* W2, double* Q,
                    double* R, hipStream_t stream)
{
    const int k = m < n ? m : n;
    if (k == 0) return;
    hipLaunchKernelGGL(qr_extract_r_kernel, dim3(n), dim3(256), 0, stream, A, m, n, k, diag, R);
    set_identity_launch(Q, m, k, m, stream);
    const int np = (k + QR_NB - 1) / QR_NB;
    for (int pi = np - 1; pi >= 0; --pi) { // Q <- (I - V T V^T) Q, panels in reverse; columns left of j0 are still unit vectors
        const int j0 = pi * QR_NB;
        const int w = (k - j0) < QR_NB ? (k - j0) : QR_NB;
        const int nq = k - j0, mp = m - j0;
        const double* Vp = Vall + j0 + (size_t)m * j0;
        const double* Tp = Tall + (size_t)pi * QR_NB * QR_NB;
        double* Qs = Q + j0 + (size_t)m * j0;
        GemmDesc g;
        g.batch = 1;
        g.strideA = g.strideB = g.strideC = 0;
        g.m = w; g.n = nq; g.k = mp;
        g.A = Vp; g.lda = m; g.transA = 1;
        g.B = Qs; g.ldb = m; g.transB = 0;
        g.C = W; g.ldc = QR_NB; g.alpha = 1.0; g.beta = 0.0;
        gemm_launch(g, stream);
        g.m = w; g.n = nq; g.k = w;
        g.A = Tp; g.lda = QR_NB; g.transA = 0;
        g.B = W; g.ldb = QR_NB; g.transB = 0;
        g.C = W2; g.ldc = QR_NB; g.alpha = 1.0; g.beta = 0.0;
        gemm_launch(g, stream);
        g.m = mp; g.n = nq; g.k = w;
        g.A = Vp; g.lda = m; g.transA = 0;
        g.B = W2; g.ldb = QR_NB; g.transB = 0;
        g.C = Qs; g.ldc = m; g.alpha = -1.0; g.beta = 1.0;
        gemm_launch(g, stream);
    }
}

} // namespace t4a
