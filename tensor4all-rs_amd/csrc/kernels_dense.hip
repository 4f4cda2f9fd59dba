// kernels_dense.hip — dense f64 kernels under the TCI2 sweep (K3/K4/K5 of SURVEY.md §2):
//   gemm (f64 MFMA 16x16x4, LDS-staged)      <- mat_mul / batched_mat_mul_same_shape (tensorbackend/src/matrix.rs:1488,1538)
//   batched left triangular solve            <- triangular_solve_matrix (tensorbackend/src/backend.rs:924)
//   batched partial-pivot LU                 <- solve_matrix (backend.rs:865, tenferro `solve`)
//   transpose / gather / scatter utilities   <- submatrix, transpose, apply_*_permutation (matrix.rs:1126,1233;
//                                               core/src/matrix_luci.rs:156-174)
//   batched tensor-train evaluation          <- AbstractTensorTrain::evaluate (simplett/src/traits.rs:146-212)
// Values of these ops are tolerance-level in the reference (third-party tenferro); pivot choice in the LU is
// "first maximum of |a_ik|".  Built with -ffp-contract=off so the non-MFMA kernels round like the CPU oracle.
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include "common.hpp"

#include <cstdlib>

namespace t4a {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// GEMM: 64x64 block tile, BK = 16, 4 waves (2x2), each wave a 32x32 tile = 2x2 MFMA 16x16x4 blocks.
// The MFMA computes D' = B^T A^T = C^T so that a lane's accumulator column index runs over C ROWS
// (lane&15 = 16 consecutive rows of one C column -> 128-byte contiguous stores).
// ------------------------------------------------------------------------------------------------
// BK = 32: 32 MFMAs (2 048 cycles) per wave and k-step cover the global-load latency.  Measured alternatives at 1024^3 /
// 2048^3 (round 2, 64 x 64 tiles): BK 16 + pad 4 84 / 462 us, BK 32 + pad 4 80 / 463 us, BK 16 + pad 16 (bank-conflict-free
// operand reads, three workgroups per CU) 103 / 559 us, BK 32 + pad 16 99 / 559 us.
// The block tile is 64 x BN with BN = 64 or 32: a grid of 64 x 64 tiles that leaves the chip with fewer than two workgroups
// per compute unit (1024^3: 256 tiles on 256 CUs, one wave per SIMD, every LDS / barrier stall exposed) is launched with
// 64 x 32 tiles instead.  Interior tiles advance per-thread pointers and skip every bounds test.
constexpr int GBM = 64, GBK = 32, GPAD = 4;

template <int BN>
__global__ void __launch_bounds__(256) gemm_kernel(GemmDesc d)
{
    // double-buffered LDS tiles: the global loads of tile k + 1 are issued before the MFMAs of tile k and parked in
    // registers, so their latency hides behind the MFMAs (64 cycles each on gfx950) a wave issues per tile; one barrier
    // per k-step
    extern __shared__ __attribute__((aligned(16))) char gemm_smem[];
    typedef double TileA[GBK][GBM + GPAD];
    typedef double TileB[GBK][BN + GPAD];
    TileA* As = reinterpret_cast<TileA*>(gemm_smem);                      // As[buf][k][m]
    TileB* Bs = reinterpret_cast<TileB*>(gemm_smem + 2 * sizeof(TileA));  // Bs[buf][k][n]
    constexpr int WN = BN / 2;  // columns of a wave's tile
    constexpr int NI = WN / 16; // MFMA blocks of a wave along n
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // XCD-aware tile order: workgroups b, b + 8, b + 16, ... of a launch land on the same XCD (its own L2), so XCD x takes
    // the x-th eighth of the tiles in column-major tile order — a band of B columns with every A row block, all of whose
    // workgroups walk k together: operands are fetched into each L2 once instead of by every XCD
    int tile_m = blockIdx.x, tile_n = blockIdx.y;
    {
        const int tm = gridDim.x, T = gridDim.x * gridDim.y;
        if ((T & 7) == 0) {
            const int bid = blockIdx.x + tm * blockIdx.y;
            const int t = (bid & 7) * (T >> 3) + (bid >> 3);
            tile_m = t % tm;
            tile_n = t / tm;
        }
    }
    const int m0 = tile_m * GBM, n0 = tile_n * BN;
    const int bz = (int)blockIdx.z / d.ksplit, kslice = (int)blockIdx.z - bz * d.ksplit; // (split-K: slices of one problem are neighbours in z)
    const double* A = d.A + (size_t)bz * d.strideA;
    const double* B = d.B + (size_t)bz * d.strideB;
    double* C = d.C + (size_t)bz * d.strideC;
    // element strides of op(A)(m,k) and op(B)(k,n)
    const long long sam = d.transA ? d.lda : 1, sak = d.transA ? 1 : d.lda;
    const long long sbk = d.transB ? d.ldb : 1, sbn = d.transB ? 1 : d.ldb;

    double4_t acc[2][NI];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // per-thread staging coordinates: the fast thread index runs along the unit-stride index of the operand
    constexpr int NQA = GBM * GBK / 256, NQB = BN * GBK / 256; // elements per thread and operand tile
    int ar[NQA], ak[NQA], bc[NQB], bk[NQB];
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
        if (!d.transA) {
            ar[q] = tid & 63;
            ak[q] = (tid >> 6) + 4 * q;
        } else {
            ak[q] = tid & (GBK - 1);
            ar[q] = tid / GBK + (256 / GBK) * q;
        }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        if (!d.transB) {
            bk[q] = tid & (GBK - 1);
            bc[q] = tid / GBK + (256 / GBK) * q;
        } else {
            bc[q] = tid & (BN - 1);
            bk[q] = tid / BN + (256 / BN) * q;
        }
    }
    // interior tile: every row / column of the tile exists, so only the last (partial) k-tile needs bounds tests
    const bool interior = (m0 + GBM <= d.m) && (n0 + BN <= d.n);
    const double* pa[NQA];
    const double* pb[NQB];
#pragma unroll
    for (int q = 0; q < NQA; ++q) pa[q] = A + (long long)(m0 + (interior ? ar[q] : 0)) * sam + (long long)ak[q] * sak;
#pragma unroll
    for (int q = 0; q < NQB; ++q) pb[q] = B + (long long)bk[q] * sbk + (long long)(n0 + (interior ? bc[q] : 0)) * sbn;
    const long long stepA = (long long)GBK * sak, stepB = (long long)GBK * sbk;

    double ra[NQA], rb[NQB];
    auto load_tile_checked = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int gm = m0 + ar[q], gk = k0 + ak[q];
            ra[q] = (gm < d.m && gk < d.k) ? A[(long long)gm * sam + (long long)gk * sak] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            const int gn = n0 + bc[q], gk = k0 + bk[q];
            rb[q] = (gn < d.n && gk < d.k) ? B[(long long)gk * sbk + (long long)gn * sbn] : 0.0;
        }
    };
    auto load_tile_fast = [&]() { // the pointers stand at the tile to load
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            ra[q] = *pa[q];
            pa[q] += stepA;
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            rb[q] = *pb[q];
            pb[q] += stepB;
        }
    };
    auto load_tile = [&](int kt_) {
        if (interior && (kt_ + 1) * GBK <= d.k) load_tile_fast();
        else load_tile_checked(kt_ * GBK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQA; ++q) As[buf][ak[q]][ar[q]] = ra[q];
#pragma unroll
        for (int q = 0; q < NQB; ++q) Bs[buf][bk[q]][bc[q]] = rb[q];
    };

    const int nk_all = (d.k + GBK - 1) / GBK;
    const int nk_per = (nk_all + d.ksplit - 1) / d.ksplit;
    const int kt0 = kslice * nk_per;
    const int nk = (kt0 + nk_per < nk_all ? kt0 + nk_per : nk_all);
    if (kt0 > 0) { // (split-K: the pointers of the fast path start at this slice's first k-tile)
#pragma unroll
        for (int q = 0; q < NQA; ++q) pa[q] += (long long)kt0 * stepA;
#pragma unroll
        for (int q = 0; q < NQB; ++q) pb[q] += (long long)kt0 * stepB;
    }
    if (nk > kt0) {
        load_tile(kt0);
        store_tile(kt0 & 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1); // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 4) {
            const int kk = ks + (lane >> 4);
            double bn[NI], am[2];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bn[ni] = Bs[buf][kk][wn * WN + ni * 16 + (lane & 15)];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) am[mi] = As[buf][kk][wm * 32 + mi * 16 + (lane & 15)];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    // MFMA "A" operand = B^T (rows = n), "B" operand = A^T (cols = m)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[ni], am[mi], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1); // (the other buffer: last read before the previous barrier)
        __syncthreads();
    }
    // D'[i' = n][j' = m]: lane -> j' = lane&15 (C row), i' = (lane>>4) + 4*reg (C column)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gm = m0 + wm * 32 + mi * 16 + (lane & 15);
                const int gn = n0 + wn * WN + ni * 16 + (lane >> 4) + 4 * reg;
                if (gm < d.m && gn < d.n) {
                    if (d.ksplit > 1) {
                        d.partial[((size_t)blockIdx.z * d.n + gn) * d.m + gm] = acc[mi][ni][reg];
                    } else {
                        double* cp = C + (size_t)gn * d.ldc + gm;
                        double v = d.alpha * acc[mi][ni][reg];
                        if (d.beta != 0.0) v = v + d.beta * (*cp);
                        *cp = v;
                    }
                }
            }
}

// second pass of a split-K product: C = alpha * sum_s partial[s] + beta * C (slices summed in order: deterministic)
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(GemmDesc d)
{
    const size_t mn = (size_t)d.m * d.n;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int bz = blockIdx.y;
    if (e >= mn) return;
    const double* p = d.partial + (size_t)bz * d.ksplit * mn + e;
    double s = 0.0;
    for (int k = 0; k < d.ksplit; ++k) s += p[(size_t)k * mn];
    const size_t gn = e / d.m, gm = e - gn * d.m;
    double* cp = d.C + (size_t)bz * d.strideC + gn * d.ldc + gm;
    double v = d.alpha * s;
    if (d.beta != 0.0) v = v + d.beta * (*cp);
    *cp = v;
}

// ------------------------------------------------------------------------------------------------
// LUCI factors of a SMALL factorisation in one launch (factors_from_rrlu, tensor4all-core/src/matrix_luci.rs:176-279; rank <= 16):
//   left-orthogonal:   left = P_row^T [I_r ; L21 L11^{-1}]      right = (L11 U) P_col^T
//   right-orthogonal:  left = P_row^T (L U11)                   right = [I_r , U11^{-1} U12] P_col^T
// One thread per row of `left` / column of `right`; the leading r x r block sits in LDS.  The general path (engine.hip) is a
// chain of ten launches (transposes, trsm, gemm, extracts, scatters): 50 us for a 4 x 4 bond, which is what a launch-bound
// small problem spends most of its time on.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) luci_factors_small_kernel(const double* __restrict__ lu, int M, int N, int rk,
                                                                 const int* __restrict__ row_perm, const int* __restrict__ col_perm,
                                                                 int left_orth, double* left, double* right)
{
    extern __shared__ __attribute__((aligned(16))) double fsm[]; // fsm[i + rk * j] = lu(i, j), i, j < rk
    const int tid = threadIdx.x, T = blockDim.x;
    for (int e = tid; e < rk * rk; e += T) fsm[e] = lu[(e % rk) + (size_t)M * (e / rk)];
    __syncthreads();
    const int g = blockIdx.x * T + tid;
    if (g < M) { // row g (permuted order) of `left`
        const int i = g;
        double* out = left + row_perm[i];
        if (left_orth) {
            if (i < rk) {
                for (int j = 0; j < rk; ++j) out[(size_t)M * j] = (i == j) ? 1.0 : 0.0;
            } else { // x L11 = L21(i, :), L11 unit lower: back substitution from the last column
                for (int j = rk - 1; j >= 0; --j) {
                    double v = lu[i + (size_t)M * j];
                    for (int k = j + 1; k < rk; ++k) v = v - out[(size_t)M * k] * fsm[k + rk * j];
                    out[(size_t)M * j] = v;
                }
            }
        } else { // (L U11)(i, j) = sum_{k <= min(i, j)} L(i, k) U11(k, j), U11 unit upper, L keeps its diagonal
            for (int j = 0; j < rk; ++j) {
                const int kmax = i < j ? i : j;
                double acc = 0.0;
                for (int k = 0; k <= kmax; ++k) {
                    const double lik = (i < rk) ? fsm[i + rk * k] : lu[i + (size_t)M * k];
                    const double ukj = (k == j) ? 1.0 : fsm[k + rk * j];
                    acc = acc + lik * ukj;
                }
                out[(size_t)M * j] = acc;
            }
        }
    }
    if (g < N) { // column g (permuted order) of `right`
        const int j = g;
        double* out = right + (size_t)rk * col_perm[j];
        if (left_orth) { // (L11 U)(i, j) = sum_{k <= min(i, j)} L11(i, k) U(k, j), L11 unit lower
            for (int i = 0; i < rk; ++i) {
                const int kmax = i < j ? i : j;
                double acc = 0.0;
                for (int k = 0; k <= kmax; ++k) {
                    const double lik = (k == i) ? 1.0 : fsm[i + rk * k];
                    const double ukj = (j < rk) ? fsm[k + rk * j] : lu[k + (size_t)M * j];
                    acc = acc + lik * ukj;
                }
                out[i] = acc;
            }
        } else {
            if (j < rk) {
                for (int i = 0; i < rk; ++i) out[i] = (i == j) ? 1.0 : 0.0;
            } else { // U11 x = U12(:, j), U11 unit upper: back substitution from the last row
                for (int i = rk - 1; i >= 0; --i) {
                    double v = lu[i + (size_t)M * j];
                    for (int k = i + 1; k < rk; ++k) v = v - fsm[i + rk * k] * out[k];
                    out[i] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ in, int rows, int cols, int ldi,
                                                        double* __restrict__ out, int ldo)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + tx, c = c0 + q;
        tile[q][tx] = (r < rows && c < cols) ? in[(size_t)c * ldi + r] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int c = c0 + tx, r = r0 + q; // out(c, r)
        if (r < rows && c < cols) out[(size_t)r * ldo + c] = tile[tx][q];
    }
}

__global__ void __launch_bounds__(256) gather_kernel(const double* __restrict__ in, int ldi, const int* rows,
                                                     int nrows, const int* cols, int ncols, double* __restrict__ out,
                                                     int ldo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int r = rows ? rows[i] : i;
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
        const int c = cols ? cols[j] : j;
        out[(size_t)j * ldo + i] = in[(size_t)c * ldi + r];
    }
}

__global__ void __launch_bounds__(256) scatter_rows_kernel(const double* __restrict__ in, int ldi, const int* rows,
                                                           int nrows, int ncols, double* __restrict__ out, int ldo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int r = rows[i];
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) out[(size_t)j * ldo + r] = in[(size_t)j * ldi + i];
}

__global__ void __launch_bounds__(256) scatter_cols_kernel(const double* __restrict__ in, int ldi, int nrows,
                                                           const int* cols, int ncols, double* __restrict__ out,
                                                           int ldo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    for (int j = blockIdx.y; j < ncols; j += gridDim.y) out[(size_t)cols[j] * ldo + i] = in[(size_t)j * ldi + i];
}

__global__ void __launch_bounds__(256) fill_kernel(double* p, size_t count, double value)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        p[i] = value;
}

__global__ void __launch_bounds__(256) identity_kernel(double* p, int m, int n, int ld)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    for (int j = blockIdx.y; j < n; j += gridDim.y) p[(size_t)j * ld + i] = (i == j) ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------------
// Batched left triangular solve T X = B (X overwrites B).  blockIdx.y = problem, blockIdx.x = chunk of
// right-hand-side columns kept in LDS; column-oriented substitution (axpy form, k ascending for lower,
// descending for upper) so that every element sees its updates in a fixed order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) trsm_left_kernel(const TrsmProblem* problems, int chunk_w)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* Bs = (double*)smem_raw; // n x cw, ld = n
    const TrsmProblem pr = problems[blockIdx.y];
    if (pr.skip_flag && *pr.skip_flag != 0) return;
    const int n = pr.n;
    const int c0 = blockIdx.x * chunk_w;
    if (c0 >= pr.nrhs || n <= 0) return;
    const int cw = (pr.nrhs - c0) < chunk_w ? (pr.nrhs - c0) : chunk_w;
    const int tid = threadIdx.x, T = blockDim.x;
    for (int e = tid; e < n * cw; e += T) {
        const int i = e % n, c = e / n;
        Bs[(size_t)c * n + i] = pr.B[(size_t)(c0 + c) * pr.ldb + i];
    }
    __syncthreads();
    for (int step = 0; step < n; ++step) {
        const int k = pr.lower ? step : (n - 1 - step);
        const double* tk = pr.T + (size_t)k * pr.ldt;
        if (!pr.unit_diag) {
            const double dkk = tk[k];
            for (int c = tid; c < cw; c += T) Bs[(size_t)c * n + k] = Bs[(size_t)c * n + k] / dkk;
            __syncthreads();
        }
        const int lo = pr.lower ? k + 1 : 0;
        const int cnt = pr.lower ? (n - 1 - k) : k;
        for (int e = tid; e < cnt * cw; e += T) {
            const int i = lo + e % cnt, c = e / cnt;
            const double prod = tk[i] * Bs[(size_t)c * n + k];
            Bs[(size_t)c * n + i] = Bs[(size_t)c * n + i] - prod;
        }
        __syncthreads();
    }
    for (int e = tid; e < n * cw; e += T) {
        const int i = e % n, c = e / n;
        pr.B[(size_t)(c0 + c) * pr.ldb + i] = Bs[(size_t)c * n + i];
    }
}

// Blocked variant for large systems (n >= 64): the substitution runs inside 16 x 16 diagonal blocks (held in LDS) and the
// bulk of the work — B[rest] -= T[rest, block] * X[block] — goes to the f64 matrix cores (v_mfma_f64_16x16x4), the triangular
// factor streamed from L2 in 128-byte runs.  Values agree with the column-oriented kernel to rounding (different summation
// order); the reference pins triangular_solve only to 1e-12 on 2 x 2 systems (backend/tests/mod.rs:119-325).
constexpr int TRB = 16;
__global__ void __launch_bounds__(256) trsm_left_mfma_kernel(const TrsmProblem* problems, int chunk_w)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const TrsmProblem pr = problems[blockIdx.y];
    if (pr.skip_flag && *pr.skip_flag != 0) return;
    const int n = pr.n;
    const int c0 = blockIdx.x * chunk_w;
    if (c0 >= pr.nrhs || n <= 0) return;
    const int cw = (pr.nrhs - c0) < chunk_w ? (pr.nrhs - c0) : chunk_w;
    const int ld = n | 1;                       // odd leading dimension: conflict-free operand reads
    double* Bs = (double*)smem_raw;             // n x cw
    double* D = Bs + (size_t)ld * chunk_w;      // TRB x TRB diagonal block, ld = TRB + 1
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    for (int e = tid; e < n * cw; e += T) {
        const int i = e % n, c = e / n;
        Bs[(size_t)c * ld + i] = pr.B[(size_t)(c0 + c) * pr.ldb + i];
    }
    const int nblk = (n + TRB - 1) / TRB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int k0 = pr.lower ? blk * TRB : ((n - (blk + 1) * TRB) > 0 ? n - (blk + 1) * TRB : 0);
        const int k1 = pr.lower ? ((k0 + TRB) < n ? k0 + TRB : n) : n - blk * TRB;
        const int bw = k1 - k0;
        __syncthreads();
        for (int e = tid; e < bw * bw; e += T) {
            const int i = e % bw, j = e / bw;
            D[j * (TRB + 1) + i] = pr.T[(size_t)(k0 + j) * pr.ldt + k0 + i];
        }
        __syncthreads();
        // substitution inside the diagonal block
        for (int st = 0; st < bw; ++st) {
            const int kk = pr.lower ? st : (bw - 1 - st);
            if (!pr.unit_diag) {
                const double dkk = D[kk * (TRB + 1) + kk];
                for (int c = tid; c < cw; c += T) Bs[(size_t)c * ld + k0 + kk] = Bs[(size_t)c * ld + k0 + kk] / dkk;
                __syncthreads();
            }
            const int lo = pr.lower ? kk + 1 : 0;
            const int cnt = pr.lower ? (bw - 1 - kk) : kk;
            for (int e = tid; e < cnt * cw; e += T) {
                const int i = lo + e % cnt, c = e / cnt;
                const double prod = D[kk * (TRB + 1) + i] * Bs[(size_t)c * ld + k0 + kk];
                Bs[(size_t)c * ld + k0 + i] = Bs[(size_t)c * ld + k0 + i] - prod;
            }
            __syncthreads();
        }
        // remaining rows: Bs[rows, :] -= T[rows, k0:k1] * X[k0:k1, :]   ("A" = X^T: i' = column, "B" = T^T: j' = row)
        const int r_lo = pr.lower ? k1 : 0, r_hi = pr.lower ? n : k0;
        for (int c0t = 0; c0t < cw; c0t += 16) {
            double xfrag[TRB / 4];
#pragma unroll
            for (int ks = 0; ks < TRB / 4; ++ks) {
                const int j = 4 * ks + lk, c = c0t + lr;
                xfrag[ks] = (j < bw && c < cw) ? Bs[(size_t)c * ld + k0 + j] : 0.0;
            }
            for (int i0 = r_lo + 16 * wave; i0 < r_hi; i0 += 16 * nwaves) {
                double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < TRB / 4; ++ks) {
                    const int j = 4 * ks + lk, i = i0 + lr;
                    const double tv = (j < bw && i < r_hi) ? pr.T[(size_t)(k0 + j) * pr.ldt + i] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xfrag[ks], tv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = i0 + lr, c = c0t + lk + 4 * reg;
                    if (i < r_hi && c < cw) Bs[(size_t)c * ld + i] = Bs[(size_t)c * ld + i] - acc[reg];
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < n * cw; e += T) {
        const int i = e % n, c = e / n;
        pr.B[(size_t)(c0 + c) * pr.ldb + i] = Bs[(size_t)c * ld + i];
    }
}

// ------------------------------------------------------------------------------------------------
// Batched partial-pivot LU (one workgroup per problem, right-looking, in place in global memory).
// Pivot = first maximum of |a_ik| over i >= k.  Row swaps are applied to the whole rows of A and to the
// optional right-hand sides B.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) lu_kernel(const LuProblem* problems)
{
    __shared__ double red_v[16];
    __shared__ int red_i[16];
    __shared__ int piv_s;
    __shared__ double pivval_s;
    const LuProblem pr = problems[blockIdx.x];
    const int n = pr.n;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    double* A = pr.A;
    const int lda = pr.lda;
    if (pr.pmax_bits) { // zero-pivot-matrix guard (tensorci2.rs:1154-1157): every |p| < EPS
        const double pmax = __longlong_as_double((long long)*pr.pmax_bits);
        if (pmax < 2.220446049250313e-16) {
            if (tid == 0) pr.info[0] = -1;
            return;
        }
    }
    if (tid == 0) pr.info[0] = 0;
    for (int k = 0; k < n; ++k) {
        // (1) pivot search in column k
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int i = k + tid; i < n; i += T) {
            const double v = fabs(A[(size_t)k * lda + i]);
            if (v > bv || (v == bv && i < bi)) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            double v = red_v[0];
            int idx = red_i[0];
            for (int q = 1; q < nw; ++q)
                if (red_v[q] > v || (red_v[q] == v && red_i[q] < idx)) {
                    v = red_v[q];
                    idx = red_i[q];
                }
            piv_s = idx;
            pr.piv[k] = idx;
            if (!(v > 0.0)) {
                if (pr.info[0] == 0) pr.info[0] = k + 1;
            }
        }
        __syncthreads();
        const int p = piv_s;
        // (2) swap rows k <-> p of A (all columns) and of B
        if (p != k && p < n) {
            for (int c = tid; c < n; c += T) {
                const double t = A[(size_t)c * lda + k];
                A[(size_t)c * lda + k] = A[(size_t)c * lda + p];
                A[(size_t)c * lda + p] = t;
            }
            if (pr.B)
                for (int c = tid; c < pr.nrhs; c += T) {
                    const double t = pr.B[(size_t)c * pr.ldb + k];
                    pr.B[(size_t)c * pr.ldb + k] = pr.B[(size_t)c * pr.ldb + p];
                    pr.B[(size_t)c * pr.ldb + p] = t;
                }
        }
        __syncthreads();
        if (tid == 0) pivval_s = A[(size_t)k * lda + k];
        __syncthreads();
        const double piv = pivval_s;
        if (piv == 0.0 || piv != piv) continue; // singular column: leave it (info already set)
        // (3) scale the column
        for (int i = k + 1 + tid; i < n; i += T) A[(size_t)k * lda + i] = A[(size_t)k * lda + i] / piv;
        __syncthreads();
        // (4) rank-1 update of the trailing block
        const int rem = n - k - 1;
        for (long long e = tid; e < (long long)rem * rem; e += T) {
            const int i = k + 1 + (int)(e % rem), c = k + 1 + (int)(e / rem);
            const double prod = A[(size_t)k * lda + i] * A[(size_t)c * lda + k];
            A[(size_t)c * lda + i] = A[(size_t)c * lda + i] - prod;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Blocked partial-pivot LU with the forward substitution of the right-hand sides folded in:
//   A = P^T L U in place,   B <- L^{-1} P B.
// Right-looking over column panels of width nb; per panel one `lu_panel_kernel` launch (one workgroup per problem
// factors the m x w panel in its registers) and one `lu_update_kernel` launch (one workgroup per nb-wide column tile of [A | B]:
// applies the panel's row swaps, then the w rank-1 updates of its columns with the L panel held in LDS).  Every
// element sees exactly the update sequence of the unblocked algorithm (k ascending, separately rounded multiply and
// subtract), so the factors are bitwise those of `lu_kernel` followed by the unit-lower `trsm_left_kernel`.
// ------------------------------------------------------------------------------------------------
// Panel factorisation, round 3: the m x w panel (w <= NB columns, m <= 256 RP rows) lives in the REGISTERS of one 256-thread
// workgroup — thread t holds rows t + 256 r (r < RP), NB values each, NB * RP = 32 doubles per thread for every panel width —
// instead of in LDS.  Per column: pivot search by wave shuffles + one four-entry LDS round, the pivot row and row j trade
// places through two NB-double LDS rows (the pivot row stays there as the broadcast operand of the update), scaling and the
// rank-1 update of the panel's remaining columns in registers.  Two barriers per column where the LDS-resident version needed
// six (2.7 us per column, half of fill_site_tensors' device time in round 2).  Same pivots, same operation order per element
// (separately rounded multiply and subtract, IEEE division): bitwise the factors of the LDS version and of lu_kernel.
struct PanelShared {
    double* U;     // [2][NB] pivot rows (alternating)
    double* V;     // [2][NB] displaced rows
    double* red_v; // [2][4]
    int* red_i;    // [2][4]
};

// one column of the panel; J is a template parameter so that every register index is a compile-time constant (the panel must
// stay in registers: as a run-time loop the compiler put it into scratch memory)
// wave reductions through DPP (row reductions + two row broadcasts; lane 63 holds the result): a 64-lane maximum is six dependent
// v_max / v_min instead of six ds_bpermute round trips per operand (__shfl_xor).  Lanes without a source keep their own value.
template <int CTRL> __device__ __forceinline__ int pnl_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double pnl_dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = pnl_dpp_i32<CTRL>((int)(b & 0xFFFFFFFFll));
    const int hi = pnl_dpp_i32<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double pnl_wave_max_f64(double x) // no NaN among the operands
{
    x = fmax(x, pnl_dpp_f64<0xB1>(x));  // quad_perm [1,0,3,2]
    x = fmax(x, pnl_dpp_f64<0x4E>(x));  // quad_perm [2,3,0,1]
    x = fmax(x, pnl_dpp_f64<0x141>(x)); // row_half_mirror
    x = fmax(x, pnl_dpp_f64<0x140>(x)); // row_mirror
    x = fmax(x, pnl_dpp_f64<0x142>(x)); // row_bcast15
    x = fmax(x, pnl_dpp_f64<0x143>(x)); // row_bcast31
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int pnl_wave_min_i32(int x)
{
    int o;
    o = pnl_dpp_i32<0xB1>(x);
    x = o < x ? o : x;
    o = pnl_dpp_i32<0x4E>(x);
    x = o < x ? o : x;
    o = pnl_dpp_i32<0x141>(x);
    x = o < x ? o : x;
    o = pnl_dpp_i32<0x140>(x);
    x = o < x ? o : x;
    o = pnl_dpp_i32<0x142>(x);
    x = o < x ? o : x;
    o = pnl_dpp_i32<0x143>(x);
    x = o < x ? o : x;
    return __builtin_amdgcn_readlane(x, 63);
}

template <int NB, int RP, int J>
__device__ __forceinline__ void lu_panel_column(double (&a)[RP][NB], const PanelShared& sh, const LuProblem& pr, int kb, int m, int w, int tid,
                                                int& first_bad)
{
    if (J >= w) return; // (uniform)
    constexpr int par = J & 1;
    const int lane = tid & 63, wave = tid >> 6;
    double* U = sh.U + par * NB;
    double* V = sh.V + par * NB;
    // pivot search: largest |a_iJ| over rows i >= J, smallest row index among equals; a NaN never wins
    double bv = -1.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int i = tid + 256 * r;
        if (i >= J && i < m) {
            const double v = fabs(a[r][J]);
            if (v > bv || (v == bv && i < bi)) {
                bv = v;
                bi = i;
            }
        }
    }
    {
        // (same winner as the shuffle tournament it replaces: the largest |a|, the smallest row among equals)
        const double wv = pnl_wave_max_f64(bv);
        const int wi = pnl_wave_min_i32(bv == wv ? bi : 0x7fffffff);
        bv = wv;
        bi = wi;
    }
    if (lane == 0) {
        sh.red_v[par * 4 + wave] = bv;
        sh.red_i[par * 4 + wave] = bi;
    }
    __syncthreads();
    double v = sh.red_v[par * 4];
    int p = sh.red_i[par * 4];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const double qv = sh.red_v[par * 4 + q];
        const int qi = sh.red_i[par * 4 + q];
        if (qv > v || (qv == v && qi < p)) {
            v = qv;
            p = qi;
        }
    }
    if (tid == 0) {
        pr.piv[kb + J] = kb + p;
        if (!(v > 0.0) && first_bad == 0) first_bad = kb + J + 1;
    }
    const int prow = (p < m) ? p : J; // (no finite candidate: the column keeps its order)
    // the pivot row goes to U (everybody's operand below), row J to V when the two trade places
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int i = tid + 256 * r;
        if (i == prow) {
#pragma unroll
            for (int c = 0; c < NB; ++c) U[c] = a[r][c];
        }
        if (prow != J && i == J) {
#pragma unroll
            for (int c = 0; c < NB; ++c) V[c] = a[r][c];
        }
    }
    __syncthreads();
    if (prow != J) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const int i = tid + 256 * r;
            if (i == J) {
#pragma unroll
                for (int c = 0; c < NB; ++c) a[r][c] = U[c];
            }
            if (i == prow) {
#pragma unroll
                for (int c = 0; c < NB; ++c) a[r][c] = V[c];
            }
        }
    }
    const double piv = U[J];
    if (piv == 0.0 || piv != piv) return; // singular column: left as it is (uniform)
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int i = tid + 256 * r;
        if (i > J && i < m) {
            const double l = a[r][J] / piv;
            a[r][J] = l;
#pragma unroll
            for (int c = J + 1; c < NB; ++c) {
                const double prod = l * U[c];
                a[r][c] = a[r][c] - prod;
            }
        }
    }
}

template <int NB, int RP, int... Js>
__device__ __forceinline__ void lu_panel_columns(double (&a)[RP][NB], const PanelShared& sh, const LuProblem& pr, int kb, int m, int w, int tid,
                                                 int& first_bad, std::integer_sequence<int, Js...>)
{
    (lu_panel_column<NB, RP, Js>(a, sh, pr, kb, m, w, tid, first_bad), ...);
}

// Panel factorisation, round 3: the m x w panel (w <= NB columns, m <= 256 RP rows) lives in the REGISTERS of one 256-thread
// workgroup — thread t holds rows t + 256 r (r < RP), NB values each, NB * RP = 32 doubles per thread for every panel width —
// instead of in LDS.  Per column: pivot search by wave shuffles + one four-entry LDS round, the pivot row and row j trade
// places through two NB-double LDS rows (the pivot row stays there as the broadcast operand of the update), scaling and the
// rank-1 update of the panel's remaining columns in registers.  Two barriers per column where the LDS-resident version needed
// six (2.7 us per column, half of fill_site_tensors' device time in round 2).  Same pivots, same operation order per element
// (separately rounded multiply and subtract, IEEE division): bitwise the factors of the LDS version and of lu_kernel.
template <int NB, int RP>
__global__ void __launch_bounds__(256) lu_panel_kernel(const LuProblem* problems, int kb)
{
    __shared__ double U[2 * NB], V[2 * NB];
    __shared__ double red_v[8];
    __shared__ int red_i[8];
    const LuProblem pr = problems[blockIdx.x];
    const int n = pr.n;
    const int tid = threadIdx.x;
    if (kb == 0) {
        if (pr.pmax_bits) { // zero-pivot-matrix guard (tensorci2.rs:1154-1157): every |p| < EPS
            const double pmax = __longlong_as_double((long long)*pr.pmax_bits);
            if (pmax < 2.220446049250313e-16) {
                if (tid == 0) pr.info[0] = -1;
                return;
            }
        }
        if (tid == 0) pr.info[0] = 0;
    } else if (pr.info[0] == -1) {
        return;
    }
    if (kb >= n) return;
    const int m = n - kb, w = (n - kb) < NB ? (n - kb) : NB;
    double* A = pr.A;
    const int lda = pr.lda;
    double a[RP][NB];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int i = tid + 256 * r;
#pragma unroll
        for (int c = 0; c < NB; ++c) a[r][c] = (i < m && c < w) ? A[(size_t)(kb + c) * lda + kb + i] : 0.0;
    }
    int first_bad = 0; // (thread 0) first column without a usable pivot, 1-based
    PanelShared sh{U, V, red_v, red_i};
    lu_panel_columns<NB, RP>(a, sh, pr, kb, m, w, tid, first_bad, std::make_integer_sequence<int, NB>{});
    if (tid == 0 && first_bad != 0 && pr.info[0] == 0) pr.info[0] = first_bad;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int i = tid + 256 * r;
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (i < m && c < w) A[(size_t)(kb + c) * lda + kb + i] = a[r][c];
    }
}

// Work items (tile, problem) are handed out through a ticket counter when `ticket` != nullptr, and workgroups that landed on
// XCD `avoid_xcc` return without taking any: fill_site_tensors runs beside the bond chain, whose single-XCD rrLU launches keep
// one XCD occupied almost without a gap (8 us between launches).  A statically mapped grid would leave an eighth of its
// workgroups waiting for that XCD — this kernel's 130 KiB of LDS do not fit next to an rrLU workgroup — and one launch took as
// long as two factorisations (measured: 1.35 ms instead of 0.2 ms); a workgroup that only has to return gets its slot at once.
__global__ void __launch_bounds__(256) lu_update_kernel(const LuProblem* problems, int kb, int nb, int tiles, int n_problems, int avoid_xcc,
                                                        unsigned* ticket, int a_only)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_item;
    if (avoid_xcc >= 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((int)(xcc & 0xF) == avoid_xcc) return;
    }
    const int total_items = tiles * n_problems;
    bool first = true;
    for (;;) {
    __syncthreads(); // (the LDS tiles of the previous item are free)
    if (threadIdx.x == 0) s_item = ticket ? (int)atomicAdd(ticket, 1u) : (first ? (int)blockIdx.x : total_items);
    __syncthreads();
    first = false;
    const int item = s_item;
    if (item >= total_items) break;
    const LuProblem pr = problems[item / tiles];
    const int n = pr.n;
    if (kb >= n || pr.info[0] == -1) continue;
    // tiles: ceil(n / nb) tiles over the columns of A, then ceil(nrhs / nb) tiles over the columns of B
    const int ta = (n + nb - 1) / nb;
    const int nrhs = pr.B ? pr.nrhs : 0;
    const int t = item % tiles;
    const bool in_a = t < ta;
    const int c0 = in_a ? t * nb : (t - ta) * nb; // first column inside A resp. B
    if (!in_a && (a_only || c0 >= nrhs)) continue; // (a_only: the right-hand sides are solved behind the factorisation, lu_solve_kernel)
    if (in_a && c0 == kb) continue;               // the panel itself
    const int m = n - kb, w = (n - kb) < nb ? (n - kb) : nb;
    const int lim = in_a ? n : nrhs;
    const int tc = (lim - c0) < nb ? (lim - c0) : nb;
    const bool left = in_a && c0 < kb;            // columns of L already final: row swaps only
    const int tid = threadIdx.x, T = blockDim.x;
    const int ldp = m | 1;
    double* L = (double*)smem_raw;               // m x w panel (L below, U11 on / above its diagonal)
    double* Tt = L + (size_t)ldp * nb;           // m x tc tile
    // column c of the tile in global memory (rows kb..n-1)
    auto gcol = [&](int c) -> double* {
        return in_a ? pr.A + (size_t)(c0 + c) * pr.lda + kb : pr.B + (size_t)(c0 + c) * pr.ldb + kb;
    };
    {
        // tile and panel into the LDS: a thread requests its row of ALL columns before it stores the first value (round 5: as
        // load -> store loops these were up to 64 dependent memory round trips at the head of every work item)
        const double* const tbase = in_a ? pr.A + (size_t)c0 * pr.lda + kb : pr.B + (size_t)c0 * pr.ldb + kb;
        const size_t tstride = in_a ? (size_t)pr.lda : (size_t)pr.ldb;
        const double* const lbase = pr.A + (size_t)kb * pr.lda + kb;
        // (four columns at a time, tile first, then panel.  The kernel must stay inside 96 vector registers: beside a bond chain the
        // workgroups the dispatcher places on the chain's XCD only have to RETURN (avoid_xcc), but they get their slot at once only if
        // a wave of this kernel fits next to the rrLU workgroup's two waves per SIMD (2 x 206 of 512 registers).  With all 32 columns of
        // tile and panel in flight (169 registers, then 107) every launch waited for the running rrLU launch to end: 43 -> 430 us per
        // launch, the fill beside the chain 2.2 -> 6.2 ms of device time per sweep.)
        constexpr int UPD_BATCH = 4;
        for (int i = tid; i < m; i += T) {
            double tv[UPD_BATCH];
#pragma unroll 1
            for (int cb = 0; cb < tc; cb += UPD_BATCH) {
#pragma unroll
                for (int c = 0; c < UPD_BATCH; ++c) tv[c] = (cb + c) < tc ? tbase[(size_t)(cb + c) * tstride + i] : 0.0;
#pragma unroll
                for (int c = 0; c < UPD_BATCH; ++c)
                    if ((cb + c) < tc) Tt[(size_t)(cb + c) * ldp + i] = tv[c];
            }
            if (!left) {
#pragma unroll 1
                for (int cb = 0; cb < w; cb += UPD_BATCH) {
#pragma unroll
                    for (int c = 0; c < UPD_BATCH; ++c) tv[c] = (cb + c) < w ? lbase[(size_t)(cb + c) * pr.lda + i] : 0.0;
#pragma unroll
                    for (int c = 0; c < UPD_BATCH; ++c)
                        if ((cb + c) < w) L[(size_t)(cb + c) * ldp + i] = tv[c];
                }
            }
        }
    }
    __syncthreads();
    // (a) the panel's row swaps, in order
    if (tid < tc) {
        double* col = Tt + (size_t)tid * ldp;
        for (int j = 0; j < w; ++j) {
            const int p = pr.piv[kb + j] - kb;
            if (p != j && p >= 0 && p < m) {
                const double t = col[j];
                col[j] = col[p];
                col[p] = t;
            }
        }
    }
    __syncthreads();
    if (!left) {
        // (b) rows kb..kb+w-1 of U (and of L^{-1} P B): forward substitution with the unit lower L11, k ascending
        for (int j = 0; j < w; ++j) {
            const double piv = L[(size_t)j * ldp + j];
            if (!(piv == 0.0 || piv != piv)) { // a singular column was skipped by the panel kernel as well
                const int cnt = w - j - 1;
                for (int e = tid; e < cnt * tc; e += T) {
                    const int i = j + 1 + e % cnt, c = e / cnt;
                    const double prod = L[(size_t)j * ldp + i] * Tt[(size_t)c * ldp + j];
                    Tt[(size_t)c * ldp + i] = Tt[(size_t)c * ldp + i] - prod;
                }
            }
            __syncthreads();
        }
        // (c) trailing rows: C[rem x tc] -= L21[rem x w] * U12[w x tc] on the f64 matrix cores (v_mfma_f64_16x16x4): every
        // wave takes 16-row strips, both operands come straight from the LDS tiles.  (Round 1 did this with w separately rounded
        // rank-1 updates per element on the vector ALUs to stay bitwise equal to the CPU restatement of `solve`; that routine
        // restates a third-party operation the reference itself only pins to 1e-10, so the cores are tolerance-level anyway.)
        // MFMA operand roles as in gemm_kernel: "A" = U12^T (i' = tile column), "B" = L21^T (j' = strip row).
        const int rem = m - w;
        const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
        const int lr = lane & 15, lk = lane >> 4;
        for (int c0t = 0; c0t < tc; c0t += 16) {
            // U12 fragments of this 16-column tile for all k-steps (w <= 32: at most 8), zero outside the tile / singular pivots
            double ufrag[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int j = 4 * ks + lk, c = c0t + lr;
                double v = 0.0;
                if (j < w && c < tc) {
                    const double piv = L[(size_t)j * ldp + j];
                    if (!(piv == 0.0 || piv != piv)) v = Tt[(size_t)c * ldp + j];
                }
                ufrag[ks] = v;
            }
            for (int i0 = w + 16 * wave; i0 < m; i0 += 16 * nwaves) {
                double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int j = 4 * ks + lk, i = i0 + lr;
                    const double lv = (j < w && i < m) ? L[(size_t)j * ldp + i] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ufrag[ks], lv, acc, 0, 0, 0);
                }
                // acc[reg] = sum_j L[i][j] U[j][c] for i = i0 + lr, c = c0t + lk + 4 reg
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = i0 + lr, c = c0t + lk + 4 * reg;
                    if (i < m && c < tc) Tt[(size_t)c * ldp + i] = Tt[(size_t)c * ldp + i] - acc[reg];
                }
            }
        }
        (void)rem;
        __syncthreads();
    }
    for (int c = 0; c < tc; ++c) {
        double* g = gcol(c);
        for (int i = tid; i < m; i += T) g[i] = Tt[(size_t)c * ldp + i];
    }
    } // next work item
}


// ------------------------------------------------------------------------------------------------
// Fused solve behind the blocked LU (round 5): B <- U^{-1} L^{-1} P B for one chunk of CW right-hand sides per workgroup, the chunk
// resident in the LDS from the row gather to the write-back.  What it replaces: the right-hand-side tiles of lu_update_kernel (two
// thirds of its work items: row swaps + 32 barrier-separated substitution steps + one launch per panel) and trsm_left_mfma_kernel
// for the upper solve (its triangular operand came from memory inside the innermost loop, one exposed round trip per 16-row strip).
//   * rows: the LU's transpositions are composed into ONE permutation first — 32 swaps per group on a private copy of the identity
//     (one thread per group), then perm[i] = g_1[g_2[... g_G[i]]] by every thread — and the chunk is gathered through it;
//   * per 16-row diagonal block: the block of the factor goes to the LDS (fetched one block ahead), each WAVE substitutes its own 16
//     columns in registers (lane = column x row quarter, the solved unknown travels by a quad broadcast: no workgroup barrier inside
//     a block), then the rows outside the block get  B[rows] -= T[rows, block] X[block]  on the f64 matrix cores; the T strips of a
//     block are requested before its substitution starts and consumed behind it;
//   * the upper factor's diagonal enters as a reciprocal (one division per row instead of one per row and column).
// Values agree with the two-kernel path to rounding (different summation order) — tensor4all-tensorbackend pins `solve` to 1e-10 /
// 1e-12 on small systems only (backend/tests/mod.rs:58-325); fill_site_tensors (tensorci2.rs:1130-1182) is compared at 1e-10.
// ------------------------------------------------------------------------------------------------
constexpr int SVB = 16;      // rows of a diagonal block
constexpr int SV_MAXS = 8;   // 16-row strips per wave and block held in registers (n <= 512)
template <int Q> __device__ __forceinline__ double quad_bcast_f64(double v)
{
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6); // quad_perm [Q, Q, Q, Q]
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// one step of the substitution inside a diagonal block: unknown KK (local row) is final, the rows behind it get their update.
// b[t] = local row 4 g + t of this lane's column; D: the block, column-major with stride SVB + 1; rd: reciprocals of its diagonal
template <int KK, bool LOWER> __device__ __forceinline__ void solve_block_step(double (&b)[4], const double (&dr)[SVB][4], const double (&rdr)[SVB], int g)
{
    constexpr int go = KK >> 2, to = KK & 3;
    double xk = quad_bcast_f64<go>(b[to]);
    if (!LOWER) xk = xk * rdr[KK];
    if (g == go) b[to] = xk;
    // (D holds zeros outside the strict triangle of this pass: rows that are not behind KK get b - 0 x)
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = __builtin_fma(-dr[KK][t], xk, b[t]);
}
template <bool LOWER, int... KKs>
__device__ __forceinline__ void solve_block_steps(double (&b)[4], const double (&dr)[SVB][4], const double (&rdr)[SVB], int g, std::integer_sequence<int, KKs...>)
{
    if constexpr (LOWER) (solve_block_step<KKs, true>(b, dr, rdr, g), ...);
    else (solve_block_step<SVB - 1 - KKs, false>(b, dr, rdr, g), ...);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lu_solve_kernel(const LuProblem* problems, int cw, int n_problems, int chunks, int avoid_xcc, unsigned* ticket)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_item;
    if (avoid_xcc >= 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((int)(xcc & 0xF) == avoid_xcc) return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4; // MFMA roles
    const int cl = lane >> 2, g = lane & 3;   // substitution roles: column of the wave's group, row quarter
    const int total_items = chunks * n_problems;
    bool first = true;
    for (;;) {
        __syncthreads(); // (the LDS of the previous item is free)
        if (tid == 0) s_item = ticket ? (int)atomicAdd(ticket, 1u) : (first ? (int)blockIdx.x : total_items);
        __syncthreads();
        first = false;
        const int item = s_item;
        if (item >= total_items) break;
        const LuProblem pr = problems[item / chunks];
        const int n = pr.n;
        const int c0 = (item % chunks) * cw;
        if (n <= 0 || !pr.B || c0 >= pr.nrhs || pr.info[0] == -1) continue; // (flagged by the zero-pivot-matrix guard: the packing writes a zero core)
        const int cwr = (pr.nrhs - c0) < cw ? (pr.nrhs - c0) : cw;
        const int ld = n | 1; // odd leading dimension: conflict-free operand reads
        double* Bs = reinterpret_cast<double*>(smem_raw);            // [cw][ld]
        double* D = Bs + (size_t)ld * cw + ((ld * cw) & 1);           // [SVB][SVB], 16-byte aligned: a lane's four rows of a column are two 128-bit reads
        double* rd = D + SVB * SVB;                                   // [SVB]
        unsigned short* perm_s = reinterpret_cast<unsigned short*>(rd + SVB); // [n]
        unsigned short* piv_s = perm_s + ((n + 3) & ~3);             // [n]
        // (pointers that come out of a descriptor in memory are generic to the compiler: flat loads, which count against the LDS
        // counter as well — every wait for an LDS read would then wait for all outstanding reads of the factor.  Named global.)
        typedef const double __attribute__((address_space(1)))* gcptr;
        typedef double __attribute__((address_space(1)))* gptr;
        const gcptr T = (gcptr)pr.A;
        const gptr Bg = (gptr)pr.B;
        const int ldt = pr.lda;
        // ---- the row permutation of the factorisation (scratch: the chunk's own space, not yet in use) ----
        {
            const int G = (n + 31) / 32;
            unsigned short* gs = reinterpret_cast<unsigned short*>(smem_raw); // [G][n]
            for (int e = tid; e < G * n; e += 256) gs[e] = (unsigned short)(e % n);
            for (int i = tid; i < n; i += 256) {
                const int pv = ((const int __attribute__((address_space(1)))*)pr.piv)[i];
                piv_s[i] = (unsigned short)((pv >= 0 && pv < n) ? pv : i);
            }
            __syncthreads();
            if (tid < G) {
                unsigned short* mine = gs + (size_t)tid * n;
                const int k_hi = (32 * tid + 32) < n ? (32 * tid + 32) : n;
                for (int k = 32 * tid; k < k_hi; ++k) {
                    const int pv = piv_s[k];
                    if (pv != k) {
                        const unsigned short t = mine[k];
                        mine[k] = mine[pv];
                        mine[pv] = t;
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                int t = i;
                for (int j = G - 1; j >= 0; --j) t = gs[(size_t)j * n + t];
                perm_s[i] = (unsigned short)t;
            }
            __syncthreads();
        }
        {
            // gather: two columns per round, all their rows requested before the first one is stored
            int pi[SV_MAXS];
#pragma unroll
            for (int t = 0; t < SV_MAXS; ++t) pi[t] = (lane + 64 * t) < n ? (int)perm_s[lane + 64 * t] : 0;
            for (int cb = 2 * wave; cb < cw; cb += 8) {
                double v[2][SV_MAXS];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = cb + h;
                    const gcptr src = Bg + (size_t)(c0 + (c < cwr ? c : 0)) * pr.ldb;
#pragma unroll
                    for (int t = 0; t < SV_MAXS; ++t) v[h][t] = (c < cwr && (lane + 64 * t) < n) ? src[pi[t]] : 0.0;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < SV_MAXS; ++t)
                        if ((lane + 64 * t) < n) Bs[(size_t)(cb + h) * ld + lane + 64 * t] = v[h][t];
            }
        }
        const int nblk = (n + SVB - 1) / SVB;
        const int ngroups = cw / 16;
#ifdef T4A_SOLVE_STAMPS
        unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define SSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define SSTAMP(k) do {} while (0)
#endif
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const bool lower = pass == 0;
            auto block_range = [&](int blk, int& k0, int& k1) {
                if (lower) {
                    k0 = blk * SVB;
                    k1 = (k0 + SVB) < n ? k0 + SVB : n;
                } else {
                    k1 = n - blk * SVB;
                    k0 = (k1 - SVB) > 0 ? k1 - SVB : 0;
                }
            };
            // element (i, j) = (tid % 16, tid / 16) of a diagonal block, identity outside the matrix / the block
            auto load_d = [&](int blk) -> double {
                if (blk >= nblk) return 0.0;
                int k0, k1;
                block_range(blk, k0, k1);
                const int bw = k1 - k0, i = tid & 15, j = tid >> 4;
                if (i < bw && j < bw) return T[(size_t)(k0 + j) * ldt + k0 + i];
                return i == j ? 1.0 : 0.0;
            };
            double dnext = load_d(0);
#pragma unroll 1
            for (int blk = 0; blk < nblk; ++blk) {
                int k0, k1;
                block_range(blk, k0, k1);
                const int bw = k1 - k0;
                const int r_lo = lower ? k1 : 0, r_hi = lower ? n : k0;
                // the strips of T this wave applies behind the substitution: requested now.  Addresses are clamped into the matrix
                // instead of predicated (rows beyond r_hi produce values nobody stores; columns beyond the block meet zeros of X)
                const int avail = r_hi - r_lo - 16 * wave;
                const int ns = avail <= 0 ? 0 : ((avail + 63) >> 6) < SV_MAXS ? ((avail + 63) >> 6) : SV_MAXS; // (uniform) strips of this wave
                double tv[SV_MAXS][4];
                {
                    gcptr colp[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int j = (4 * ks + lk) < bw ? (4 * ks + lk) : bw - 1;
                        colp[ks] = T + (size_t)(k0 + j) * ldt;
                    }
                    const int ib = r_lo + 16 * wave + lr;
#pragma unroll
                    for (int sp = 0; sp < SV_MAXS; ++sp)
                        if (sp < ns) {
                            const int i = (ib + 64 * sp) < n ? (ib + 64 * sp) : n - 1;
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) tv[sp][ks] = colp[ks][i];
                        }
                }
                SSTAMP(0);
                __syncthreads(); // (1) the previous block's updates of the chunk are complete, its D is no longer read
                SSTAMP(1);
                {
                    // (the substitution multiplies unconditionally: only the strict triangle of this pass enters the LDS copy)
                    const int i = tid & 15, j = tid >> 4;
                    D[j * SVB + i] = (lower ? (i > j) : (i < j)) ? dnext : 0.0;
                    if (i == j) rd[i] = 1.0 / dnext;
                }
                dnext = load_d(blk + 1);
                __syncthreads(); // (2)
                SSTAMP(2);
                // ---- substitution inside the block: every wave its own 16-column groups, in registers ----
                // (this lane's rows of the block and the reciprocals of its diagonal come into registers FIRST: read inside the steps, every
                // one of the 16 dependent steps would wait for its own LDS round trips)
                double dr[SVB][4], rdr[SVB];
                if (wave < ngroups) {
#pragma unroll
                    for (int kk = 0; kk < SVB; ++kk) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) dr[kk][t] = D[kk * SVB + 4 * g + t];
                        rdr[kk] = rd[kk];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                for (int grp = wave; grp < ngroups; grp += 4) {
                    const int c = 16 * grp + cl;
                    double b[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int j = 4 * g + t;
                        b[t] = j < bw ? Bs[(size_t)c * ld + k0 + j] : 0.0;
                    }
                    if (lower) solve_block_steps<true>(b, dr, rdr, g, std::make_integer_sequence<int, SVB>{});
                    else solve_block_steps<false>(b, dr, rdr, g, std::make_integer_sequence<int, SVB>{});
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int j = 4 * g + t;
                        if (j < bw) Bs[(size_t)c * ld + k0 + j] = b[t];
                    }
                }
                SSTAMP(3);
                __syncthreads(); // (3)
                SSTAMP(4);
                // ---- rows outside the block: Bs[rows, :] -= T[rows, k0:k1] X[k0:k1, :] ("A" = X^T: i' = column, "B" = T^T: j' = row).
                // One straight-line body per strip count: the MFMA chains of a tile's strips run interleaved, then their results
                // leave through one batch of LDS reads and one of writes ----
                if (ns > 0) {
                    auto tiles = [&](auto nsc) {
                        constexpr int NS = decltype(nsc)::value;
                        for (int c0t = 0; c0t < cw; c0t += 16) {
                            // (the operands of the tile and the values its results are subtracted from: all requested in front of the
                            // MFMA chains, which run while the second batch is on its way)
                            double xf[4];
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) {
                                const int j = 4 * ks + lk;
                                xf[ks] = j < bw ? Bs[(size_t)(c0t + lr) * ld + k0 + j] : 0.0;
                            }
                            double old[NS][4];
#pragma unroll
                            for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) {
                                    const int i = r_lo + 16 * wave + 64 * sp + lr, c = c0t + lk + 4 * reg;
                                    old[sp][reg] = i < r_hi ? Bs[(size_t)c * ld + i] : 0.0;
                                }
                            double4_t acc[NS];
#pragma unroll
                            for (int sp = 0; sp < NS; ++sp) acc[sp] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                                for (int sp = 0; sp < NS; ++sp) acc[sp] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[ks], tv[sp][ks], acc[sp], 0, 0, 0);
#pragma unroll
                            for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) {
                                    const int i = r_lo + 16 * wave + 64 * sp + lr, c = c0t + lk + 4 * reg;
                                    if (i < r_hi) Bs[(size_t)c * ld + i] = old[sp][reg] - acc[sp][reg];
                                }
                        }
                    };
                    switch (ns) {
                    case 1: tiles(std::integral_constant<int, 1>{}); break;
                    case 2: tiles(std::integral_constant<int, 2>{}); break;
                    case 3: tiles(std::integral_constant<int, 3>{}); break;
                    case 4: tiles(std::integral_constant<int, 4>{}); break;
                    case 5: tiles(std::integral_constant<int, 5>{}); break;
                    case 6: tiles(std::integral_constant<int, 6>{}); break;
                    case 7: tiles(std::integral_constant<int, 7>{}); break;
                    default: tiles(std::integral_constant<int, 8>{}); break;
                    }
                }
                SSTAMP(5);
            }
            __syncthreads();
        }
#ifdef T4A_SOLVE_STAMPS
        if (item == (total_items / 2) && (tid & 63) == 0)
            printf("[lu_solve stamps] wave %d n=%d cw=%d cycles: tvissue=%llu bar1=%llu dstore+bar2=%llu diag=%llu bar3=%llu mfma=%llu\n", wave, n, cw,
                   st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5]);
#endif
        for (int c = wave; c < cwr; c += 4) {
            const gptr dst = Bg + (size_t)(c0 + c) * pr.ldb;
            for (int i = lane; i < n; i += 64) dst[i] = Bs[(size_t)c * ld + i];
        }
    } // next work item
}

// ------------------------------------------------------------------------------------------------
// Batched TT evaluation: one workgroup per point, v <- v * A_s[:, idx_s, :] left to right with the
// reference's summation order (l ascending, separately rounded multiply/add).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tt_eval_kernel(const TtCoreDesc* cores, int n_sites, int max_bond,
                                                      const uint32_t* __restrict__ idx, int n_pts, double* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cur = (double*)smem_raw;
    double* nxt = cur + max_bond;
    const int tid = threadIdx.x, T = blockDim.x;
    for (int pt = blockIdx.x; pt < n_pts; pt += gridDim.x) {
        const uint32_t* my = idx + (size_t)pt * n_sites;
        {
            const TtCoreDesc c0 = cores[0];
            for (int r = tid; r < c0.r; r += T) cur[r] = c0.data[(size_t)0 + c0.l * ((size_t)my[0] + (size_t)c0.d * r)];
        }
        __syncthreads();
        for (int s = 1; s < n_sites; ++s) {
            const TtCoreDesc c = cores[s];
            for (int r = tid; r < c.r; r += T) {
                const double* col = c.data + (size_t)c.l * ((size_t)my[s] + (size_t)c.d * r);
                double sum = 0.0;
                for (int l = 0; l < c.l; ++l) {
                    const double prod = cur[l] * col[l];
                    sum = sum + prod;
                }
                nxt[r] = sum;
            }
            __syncthreads();
            double* t = cur;
            cur = nxt;
            nxt = t;
        }
        if (tid == 0) out[pt] = cur[0];
        __syncthreads();
    }
}

} // namespace

// Workspace of the split-K slices, one per stream (launches on one stream are ordered, so consecutive products reuse it; a stream
// recycled to another handle still serialises its users).  Grow-only; a buffer that grows is replaced behind the stream's work.
static double* splitk_workspace(hipStream_t stream, size_t doubles)
{
    // blocks from the process-wide cache (pool.hip): a block that is replaced goes back through pool::dev_free, which waits for
    // the device outside any graph capture of ours before the block can be handed out again
    static std::mutex mu;
    static auto* const bufs = new std::map<hipStream_t, DevBuf<double>>(); // (never destroyed: no device calls during static destruction)
    std::lock_guard<std::mutex> lk(mu);
    DevBuf<double>& b = (*bufs)[stream];
    b.reserve(doubles);
    return b.get();
}

template <int BN> static void gemm_launch_bn(const GemmDesc& d, hipStream_t stream)
{
    dim3 grid((d.m + GBM - 1) / GBM, (d.n + BN - 1) / BN, d.batch * d.ksplit);
    constexpr size_t lds = 2 * sizeof(double) * GBK * ((GBM + GPAD) + (BN + GPAD));
    static std::atomic<bool> attr_set{false}; // (launches come from several host threads; setting the attribute twice is harmless)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_kernel<BN>, grid, dim3(256), lds, stream, d);
}

void gemm_launch(const GemmDesc& d, hipStream_t stream)
{
    if (d.m <= 0 || d.n <= 0 || d.batch <= 0) return;
    // two workgroups per compute unit hide each other's LDS / barrier stalls: narrower tiles when 64 x 64 ones cannot provide them
    static const int force_bn = diag_env("T4A_GEMM_BN") ? std::atoi(diag_env("T4A_GEMM_BN")) : 0;
    const long long tiles64 = (long long)((d.m + GBM - 1) / GBM) * ((d.n + 63) / 64) * d.batch;
    const bool narrow = force_bn ? force_bn == 32 : (tiles64 < 512 && d.n > 32);
    // small outputs leave most of the chip idle (a 256 x 256 result is 32 narrow tiles on 256 compute units): split K over
    // blockIdx.z and sum the slices in a second pass (a separate launch: the fence of an in-kernel reduction cost more than it
    // saved, tools/experiments/README.md).  Slices of >= 2 k-tiles, as many as fill the chip twice, at most 16.
    static const int force_split = diag_env("T4A_GEMM_KSPLIT") ? std::atoi(diag_env("T4A_GEMM_KSPLIT")) : 0;
    const long long tiles = (long long)((d.m + GBM - 1) / GBM) * ((d.n + (narrow ? 31 : 63)) / (narrow ? 32 : 64)) * d.batch;
    const int ktiles = (d.k + GBK - 1) / GBK;
    int ksplit = 1;
    if (force_split > 0) ksplit = force_split;
    else if (tiles < 256 && ktiles >= 8) ksplit = (int)std::min<long long>(std::min<long long>(16, ktiles / 2), (512 + tiles - 1) / tiles);
    if (ksplit > ktiles) ksplit = ktiles;
    if (ksplit <= 1) {
        if (narrow) gemm_launch_bn<32>(d, stream);
        else gemm_launch_bn<64>(d, stream);
        return;
    }
    GemmDesc ds = d;
    ds.ksplit = ksplit;
    ds.partial = splitk_workspace(stream, (size_t)ksplit * d.batch * d.m * d.n);
    if (narrow) gemm_launch_bn<32>(ds, stream);
    else gemm_launch_bn<64>(ds, stream);
    const size_t mn = (size_t)d.m * d.n;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256), d.batch), dim3(256), 0, stream, ds);
}

__global__ void __launch_bounds__(256) luci_left_cores_batched_kernel(LeftCoreJobs jobs)
{
    __shared__ double fsm[LUCI_LEFT_CORES_MAX_RANK * LUCI_LEFT_CORES_MAX_RANK]; // fsm[i + rk * j] = lu(i, j), i, j < rk
    // (dynamic index into a by-value argument: read through the kernel-argument segment, not through a private copy)
#if defined(__HIP_DEVICE_COMPILE__)
    const LeftCoreJob& jb = *(reinterpret_cast<const LeftCoreJob*>((const char*)__builtin_amdgcn_kernarg_segment_ptr()) + blockIdx.y);
#else
    const LeftCoreJob& jb = jobs.j[blockIdx.y];
#endif
    const int M = jb.dims[0];
    const int rk = jb.iresult[0];
    if (jb.dims[2] != 0 || M <= 0 || jb.dims[1] <= 0 || jb.iresult[1] != 0 || jb.iresult[3] != (int)jb.token || rk < 0 || rk > LUCI_LEFT_CORES_MAX_RANK) return;
    if (blockIdx.x * 256 >= M) return;
    const int tid = threadIdx.x;
    for (int e = tid; e < rk * rk; e += 256) fsm[e] = jb.lu[(e % rk) + (size_t)M * (e / rk)];
    __syncthreads();
    const int i = blockIdx.x * 256 + tid; // row i (permuted order)
    if (i >= M) return;
    const int S = jb.S, L = M / S;
    const int row = jb.row_perm[i];
    double* out = jb.core + (row / S) + (size_t)L * (row % S); // core[l, s, r] at l + L (s + S r)
    const size_t rstride = (size_t)L * S;
    if (rk == 0) {
        out[0] = 0.0; // (R = 1: a zero column, tensorci2.rs:1957-1973)
        return;
    }
    if (i < rk) {
        for (int j = 0; j < rk; ++j) out[rstride * j] = (i == j) ? 1.0 : 0.0;
    } else { // x L11 = L21(i, :), L11 unit lower: back substitution from the last column (the order of luci_factors_small_kernel)
        double v[LUCI_LEFT_CORES_MAX_RANK];
#pragma unroll
        for (int j = LUCI_LEFT_CORES_MAX_RANK - 1; j >= 0; --j) {
            if (j < rk) {
                double x = jb.lu[i + (size_t)M * j];
#pragma unroll
                for (int k = j + 1; k < LUCI_LEFT_CORES_MAX_RANK; ++k)
                    if (k < rk) x = x - v[k] * fsm[k + rk * j];
                v[j] = x;
                out[rstride * j] = x;
            }
        }
    }
}

void luci_left_cores_batched_launch(const LeftCoreJobs& jobs, int n_jobs, int max_rows, hipStream_t stream)
{
    if (n_jobs <= 0 || max_rows <= 0) return;
    hipLaunchKernelGGL(luci_left_cores_batched_kernel, dim3((max_rows + 255) / 256, n_jobs), dim3(256), 0, stream, jobs);
}

bool luci_factors_small_launch(const double* lu, int M, int N, int rk, const int* row_perm, const int* col_perm, bool left_orth,
                               double* left, double* right, hipStream_t stream)
{
    if (rk < 1 || rk > 16 || M > 1024 || N > 1024) return false; // (a thread walks rk^2 / 2 dependent steps through L2-resident rows: beyond a few hundred the general path wins)
    const int mx = M > N ? M : N;
    hipLaunchKernelGGL(luci_factors_small_kernel, dim3((mx + 255) / 256), dim3(256), (size_t)rk * rk * sizeof(double), stream, lu, M, N, rk,
                       row_perm, col_perm, left_orth ? 1 : 0, left, right);
    return true;
}

void transpose_launch(const double* in, int rows, int cols, int ldi, double* out, int ldo, hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((rows + 31) / 32, (cols + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, rows, cols, ldi, out, ldo);
}

void gather_launch(const double* in, int ldi, const int* rows, int nrows, const int* cols, int ncols, double* out,
                   int ldo, hipStream_t stream)
{
    if (nrows <= 0 || ncols <= 0) return;
    dim3 grid((nrows + 255) / 256, ncols < 1024 ? ncols : 1024);
    hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, stream, in, ldi, rows, nrows, cols, ncols, out, ldo);
}

void scatter_rows_launch(const double* in, int ldi, const int* rows, int nrows, int ncols, double* out, int ldo,
                         hipStream_t stream)
{
    if (nrows <= 0 || ncols <= 0) return;
    dim3 grid((nrows + 255) / 256, ncols < 1024 ? ncols : 1024);
    hipLaunchKernelGGL(scatter_rows_kernel, grid, dim3(256), 0, stream, in, ldi, rows, nrows, ncols, out, ldo);
}

void scatter_cols_launch(const double* in, int ldi, int nrows, const int* cols, int ncols, double* out, int ldo,
                         hipStream_t stream)
{
    if (nrows <= 0 || ncols <= 0) return;
    dim3 grid((nrows + 255) / 256, ncols < 1024 ? ncols : 1024);
    hipLaunchKernelGGL(scatter_cols_kernel, grid, dim3(256), 0, stream, in, ldi, nrows, cols, ncols, out, ldo);
}

void fill_launch(double* p, size_t count, double value, hipStream_t stream)
{
    if (count == 0) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, count, value);
}

void set_identity_launch(double* p, int m, int n, int ld, hipStream_t stream)
{
    if (m <= 0 || n <= 0) return;
    dim3 grid((m + 255) / 256, n < 1024 ? n : 1024);
    hipLaunchKernelGGL(identity_kernel, grid, dim3(256), 0, stream, p, m, n, ld);
}

void trsm_left_batched_launch(const TrsmProblem* d_problems, int n_problems, int max_n, int max_nrhs,
                              hipStream_t stream)
{
    if (n_problems <= 0 || max_n <= 0 || max_nrhs <= 0) return;
    static std::atomic<bool> attr_set{false}; // (launches come from several host threads; setting the attribute twice is harmless)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_left_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // large systems with enough right-hand sides: blocked substitution with the bulk on the matrix cores
    static const bool no_mfma = diag_env("T4A_TRSM_NO_MFMA") != nullptr;
    if (!no_mfma && max_n >= 64 && max_nrhs >= 16) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_left_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            attr2 = true;
        }
        int cw = 32;
        const size_t per_col = (size_t)(max_n | 1) * 8;
        while (cw > 16 && (per_col * cw + (TRB + 1) * TRB * 8 > 150 * 1024 || (long long)n_problems * max_nrhs / cw < 1024)) cw -= 16;
        const size_t lds = per_col * cw + (size_t)(TRB + 1) * TRB * 8;
        if (lds <= 160 * 1024) {
            dim3 grid((max_nrhs + cw - 1) / cw, n_problems);
            hipLaunchKernelGGL(trsm_left_mfma_kernel, grid, dim3(256), lds, stream, d_problems, cw);
            return;
        }
    }
    int cw = (int)((128 * 1024) / ((size_t)max_n * 8));
    if (cw > 32) cw = 32;
    // the substitution is a chain of max_n dependent steps per workgroup: prefer many thin column chunks (>= ~2000
    // workgroups in flight) over few wide ones
    static const int cw_env = diag_env("T4A_TRSM_CW") ? std::atoi(diag_env("T4A_TRSM_CW")) : 0;
    const long long total_cols = (long long)n_problems * max_nrhs;
    while (cw > 4 && total_cols / cw < 2048) cw /= 2;
    if (cw_env > 0) cw = cw_env;
    if (cw < 1) cw = 1;
    const size_t lds = (size_t)max_n * cw * 8;
    if (lds > 160 * 1024) // one right-hand-side column of the triangular system no longer fits the LDS of a compute unit
        throw Error(T4A_GPU_NOT_IMPLEMENTED, "triangular solve: systems with more than 20480 rows are not supported");
    dim3 grid((max_nrhs + cw - 1) / cw, n_problems);
    hipLaunchKernelGGL(trsm_left_kernel, grid, dim3(256), lds, stream, d_problems, cw);
}

void lu_batched_launch(const LuProblem* d_problems, int n_problems, int max_n, hipStream_t stream)
{
    if (n_problems <= 0) return;
    int T = max_n >= 128 ? 1024 : (max_n >= 32 ? 256 : 64);
    hipLaunchKernelGGL(lu_kernel, dim3(n_problems), dim3(T), 0, stream, d_problems);
}

bool lu_forward_blocked_launch(const LuProblem* d_problems, int n_problems, int max_n, int max_nrhs, hipStream_t stream, int avoid_xcc,
                               unsigned* tickets)
{
    if (n_problems <= 0 || max_n <= 0) return true;
    int nb;
    if (max_n <= 256) nb = 32;
    else if (max_n <= 512) nb = 16;
    else if (max_n <= 1024) nb = 8;
    else return false; // panel does not fit the LDS: the caller falls back to lu_kernel + trsm
    static std::atomic<bool> attr_set{false}; // (launches come from several host threads; setting the attribute twice is harmless)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_update_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipGetLastError();
        attr_set = true;
    }
    const int ldp = max_n | 1;
    const size_t lds_update = (size_t)ldp * nb * 8 * 2;
    const int tiles = (max_n + nb - 1) / nb + (max_nrhs + nb - 1) / nb + 1;
    for (int kb = 0; kb < max_n; kb += nb) {
        if (nb == 32) hipLaunchKernelGGL((lu_panel_kernel<32, 1>), dim3(n_problems), dim3(256), 0, stream, d_problems, kb);
        else if (nb == 16) hipLaunchKernelGGL((lu_panel_kernel<16, 2>), dim3(n_problems), dim3(256), 0, stream, d_problems, kb);
        else hipLaunchKernelGGL((lu_panel_kernel<8, 4>), dim3(n_problems), dim3(256), 0, stream, d_problems, kb);
        // one workgroup per (tile, problem) item; with tickets a few more, so that those which return on the avoided XCD are made up for
        const int items = tiles * n_problems;
        unsigned* tk = tickets ? tickets + kb / nb : nullptr;
        const int grid = tk ? items + items / 7 + 8 : items;
        hipLaunchKernelGGL(lu_update_kernel, dim3(grid), dim3(256), lds_update, stream, d_problems, kb, nb, tiles, n_problems,
                           tk ? avoid_xcc : -1, tk, 0);
    }
    return true;
}

bool lu_solve_blocked_launch(const LuProblem* d_problems, int n_problems, int max_n, int max_nrhs, hipStream_t stream, int avoid_xcc,
                             unsigned* tickets)
{
    if (n_problems <= 0 || max_n <= 0) return true;
    static const bool off = diag_env("T4A_NO_FUSED_SOLVE") != nullptr;
    if (off || max_n < 32 || max_n > 16 * 4 * SV_MAXS || max_nrhs < 16) return false;
    int nb;
    if (max_n <= 256) nb = 32;
    else nb = 16;
    static std::once_flag attr_once; // (launches come from several host threads)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    });
    const int ldp = max_n | 1;
    const size_t lds_update = (size_t)ldp * nb * 8 * 2;
    const int tiles = (max_n + nb - 1) / nb; // the factor's own column tiles only: the right-hand sides are solved behind the factorisation
    for (int kb = 0; kb < max_n; kb += nb) {
        if (nb == 32) hipLaunchKernelGGL((lu_panel_kernel<32, 1>), dim3(n_problems), dim3(256), 0, stream, d_problems, kb);
        else hipLaunchKernelGGL((lu_panel_kernel<16, 2>), dim3(n_problems), dim3(256), 0, stream, d_problems, kb);
        const int items = tiles * n_problems;
        unsigned* tk = tickets ? tickets + kb / nb : nullptr;
        const int grid = tk ? items + items / 7 + 8 : items;
        hipLaunchKernelGGL(lu_update_kernel, dim3(grid), dim3(256), lds_update, stream, d_problems, kb, nb, tiles, n_problems,
                           tk ? avoid_xcc : -1, tk, 1);
    }
    // chunk width of the right-hand sides: as wide as the LDS takes (fewer, longer workgroups: the triangular factor is read once per chunk)
    const size_t extra = (size_t)SVB * (SVB + 1) * 8 + SVB * 8 + 2 * (size_t)((max_n + 3) & ~3) * 2 + 64;
    int cw = 64;
    while (cw > 16 && (size_t)(max_n | 1) * cw * 8 + extra > 156 * 1024) cw -= 16;
    if (cw == 48) cw = 32;
    while (cw > 16 && max_nrhs <= cw / 2) cw /= 2;
    const size_t lds = (size_t)(max_n | 1) * cw * 8 + extra;
    const int chunks = (max_nrhs + cw - 1) / cw;
    const int items = chunks * n_problems;
    unsigned* tk = tickets ? tickets + (LU_MAX_PANEL_STEPS - 1) : nullptr;
    const int grid = tk ? items + items / 7 + 8 : items;
    hipLaunchKernelGGL(lu_solve_kernel, dim3(grid), dim3(256), lds, stream, d_problems, cw, n_problems, chunks, tk ? avoid_xcc : -1, tk);
    return true;
}

void tt_evaluate_launch(const TtCoreDesc* d_cores, int n_sites, int max_bond, const uint32_t* d_idx, int n_pts,
                        double* d_out, hipStream_t stream)
{
    if (n_pts <= 0) return;
    int blocks = n_pts < 4096 ? n_pts : 4096;
    const size_t lds = (size_t)2 * (max_bond > 0 ? max_bond : 1) * 8;
    hipLaunchKernelGGL(tt_eval_kernel, dim3(blocks), dim3(256), lds, stream, d_cores, n_sites, max_bond, d_idx, n_pts,
                       d_out);
}

} // namespace t4a
