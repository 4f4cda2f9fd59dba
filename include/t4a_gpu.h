/*
 * t4a_gpu.h — C ABI of the MI355X (gfx950) backend for the tensor4all-rs TCI2 sweep hot path.
 *
 * This is the drop-in boundary.  The reference has NO existing FFI seam for this path
 * (tensor4all-capi deliberately excludes SimpleTT/TCI: crates/tensor4all-capi/src/lib.rs:13-14), so the
 * seam sits where the reference already crosses crates: the `Matrix<f64>`-level function surface of
 * tensor4all-core / tensor4all-tensorbackend and the `TensorCI2` driver of tensor4all-tensorci.  Each
 * entry point cites the Rust item it replaces.  Conventions are copied from tensor4all-capi:
 *   - status codes (crates/tensor4all-capi/src/lib.rs:49-68), thread-local last-error string fetched on
 *     the same OS thread (lib.rs:79-83, docs/CAPI_DESIGN.md:75-80);
 *   - opaque handles `*_new(..., out)` / `*_release` (docs/CAPI_DESIGN.md:24-52);
 *   - query-then-fill for variable-length outputs (docs/CAPI_DESIGN.md:108-123);
 *   - column-major dense buffers, `m[[r,c]] <-> data[r + nrows*c]`
 *     (crates/tensor4all-tensorbackend/src/matrix.rs:31-53);
 *   - no exception crosses the boundary (lib.rs:139-162).
 * All matrix pointers are HOST pointers unless a name ends in `_device`.  One in-flight call per
 * handle (the reference serialises backend calls behind one mutex, tensorbackend/src/context.rs:96).
 * Only f64 is in scope.
 */
#ifndef T4A_GPU_H
#define T4A_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- status codes: values 0..-7 identical to tensor4all-capi's t4a_status_code ---- */
typedef int32_t t4a_gpu_status;
#define T4A_GPU_SUCCESS 0
#define T4A_GPU_NULL_POINTER (-1)
#define T4A_GPU_INVALID_ARGUMENT (-2)
#define T4A_GPU_BUFFER_TOO_SMALL (-5)
#define T4A_GPU_INTERNAL_ERROR (-6)
#define T4A_GPU_NOT_IMPLEMENTED (-7)
/* extensions: MatrixCIError::{NaNEncountered, SingularMatrix} (core/src/matrixlu.rs:653-662,904,980) */
#define T4A_GPU_NAN_ENCOUNTERED (-8)
#define T4A_GPU_SINGULAR_MATRIX (-9)
/* no usable HIP device / HIP runtime failure: the product path never falls back to the CPU */
#define T4A_GPU_NO_DEVICE (-10)
/* a bounded in-kernel spin gave up (persistent rrLU hand-off) */
#define T4A_GPU_KERNEL_TIMEOUT (-11)
/* the user callback returned a wrong number of values (tensorci2.rs:1872-1874) */
#define T4A_GPU_CALLBACK_ERROR (-12)

/* Copies the calling thread's last error message (NUL terminated). `required_len` (may be NULL)
   receives the length including the terminator.  Mirrors t4a_last_error_message (capi/src/lib.rs:273). */
t4a_gpu_status t4a_gpu_last_error_message(char* buf, size_t buf_len, size_t* required_len);

/* Number of visible HIP devices (0 on a CPU-only host; never an error). */
t4a_gpu_status t4a_gpu_device_count(int32_t* out_count);
/* Select the device used by subsequent calls of this process (one process per GPU). */
t4a_gpu_status t4a_gpu_set_device(int32_t device);
/* Library version string. */
const char* t4a_gpu_version(void);
/* 1: the library was built with -DT4A_DIAG_SWITCHES (experiment switches of tools/ are read from the environment), 0: a production
 * build — every T4A_* experiment variable is a no-op there (a known one found in the environment is reported on stderr once).  The A/B
 * launchers under tools/ call this and refuse to label a run as a variant on a production build. */
int32_t t4a_gpu_diag_switches_enabled(void);

/* The random stream of the reference's seeded searches: `rand 0.9` `StdRng::seed_from_u64(seed)` followed by
 * `rng.random_range(0..dims[i])` for i = 0 .. n-1 (tensorci2.rs:1653-1657 + globalpivot.rs:174-180,
 * adaptive_interpolation.rs:164,472-480, treetci/src/globalpivot.rs:118-122, aci/src/global_guard.rs:71-74).  Host-only (no device
 * needed): this is what every seeded search of this library draws its starting points from (csrc/stdrng.hpp). */
t4a_gpu_status t4a_gpu_stdrng_sample(uint64_t seed, const size_t* dims, size_t n, size_t* out /* n */);
/* One 64-byte ChaCha key-stream block (key: 8 little-endian words, 64-bit block counter, 64-bit stream id, `rounds` = 8 / 12 / 20):
 * the known-answer hook for the published vectors (RFC 8439 2.3.2; zero-key ChaCha20 / ChaCha12).  `StdRng` is the 12-round stream
 * with counter 0, stream 0. */
t4a_gpu_status t4a_gpu_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t* out16);
/* The reference's two other seeded streams, restated from their published algorithms (csrc/smallrng.hpp), host only:
 *   - TreeTCI proposers (tensor4all-treetci/src/proposer.rs:344-409): std `DefaultHasher` = SipHash-1-3 with the zero key over
 *     (seed, tag, edge, history length, pivot counts) -> rand 0.9 `SmallRng::seed_from_u64` (xoshiro256++) -> `random_range` /
 *     `shuffle`;
 *   - ACI initial guess (tensor4all-aci/src/random_tt.rs:31,143-150): `ChaCha8Rng::seed_from_u64` + rand_distr `StandardNormal`.
 * These entry points exist for the known-answer tests (tests/test_cpu_stdrng.py: SipHash paper vector, CPython's zero-key SipHash-2-4,
 * the xoshiro256++ reference outputs, rand's seed_from_u64(0) vector, the ChaCha8 zero-key key stream). */
t4a_gpu_status t4a_gpu_siphash(const uint8_t* msg, size_t len, uint64_t k0, uint64_t k1, int32_t c_rounds, int32_t d_rounds, uint64_t* out);
/* state4 != NULL: the four state words given directly (published vectors); otherwise seed_from_u64(seed) */
t4a_gpu_status t4a_gpu_smallrng_words(uint64_t seed, const uint64_t* state4, size_t n, uint64_t* out /* n */);
t4a_gpu_status t4a_gpu_smallrng_sample(uint64_t seed, const size_t* dims, size_t n, size_t* out /* n */);
/* the permutation `(0..n).collect::<Vec<_>>().shuffle(&mut SmallRng::seed_from_u64(seed))` leaves behind */
t4a_gpu_status t4a_gpu_smallrng_shuffle(uint64_t seed, size_t n, size_t* out /* n */);
/* rng_for_edge's hash (proposer.rs:360-387): the u64 the proposer's SmallRng is seeded with */
t4a_gpu_status t4a_gpu_tree_edge_seed(uint64_t seed, const char* tag, size_t u, size_t v, size_t history_len, size_t n_pivots_i, size_t n_pivots_j,
                                      uint64_t* out);
/* n standard normals and / or n_words key-stream words of ChaCha8Rng::seed_from_u64(seed) (each from a fresh generator) */
t4a_gpu_status t4a_gpu_chacha8_standard_normal(uint64_t seed, size_t n, double* out, size_t n_words, uint32_t* out_words);


/* =====================================================================================
 * Dense kernels (host buffers in, host buffers out; each call is synchronous)
 * ===================================================================================== */

/* rrlu_mut(&mut Matrix<f64>, Option<RrLUOptions>) -> RrLU   (core/src/matrixlu.rs:735-819)
 * In-place full-pivot rank-revealing LU.  On return `a_inout` holds the factored matrix in the
 * physically permuted order of the reference's buffer (L below / U on+above the diagonal; read L,U
 * per extract_lu_from_factorized, matrixlu.rs:614-668).  `max_bond_dim == 0` means usize::MAX.
 * row_perm[m], col_perm[n]: RrLU::row_permutation / col_permutation.  Pivot selection is bit-exact.
 * Returns T4A_GPU_NAN_ENCOUNTERED exactly when the reference returns MatrixCIError::NaNEncountered. */
t4a_gpu_status t4a_gpu_rrlu_f64(double* a_inout, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                double abs_tol, int32_t left_orthogonal, size_t* row_perm, size_t* col_perm,
                                size_t* npivots, double* last_error);

/* matrix_luci_factors_from_matrix(&Matrix<f64>, Option<RrLUOptions>) -> MatrixLuciFactors
 * (core/src/matrix_luci.rs:366-374; factors :256-279).
 * Caller-owned buffers sized for rank = min(m,n): rows/cols [min(m,n)], pivot_errors [min(m,n)+1],
 * left [m*min(m,n)], right [min(m,n)*n].  On return left is m x rank and right is rank x n, both
 * column-major and densely packed for the returned rank. */
t4a_gpu_status t4a_gpu_luci_f64(const double* a, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                double abs_tol, int32_t left_orthogonal, size_t* rank, size_t* rows, size_t* cols,
                                double* pivot_errors, double* left, double* right);

/* matrix_luci_factors_from_blocks(nrows, ncols, fill_block, RrLUOptions) (core/src/matrix_luci.rs:440-456): the lazy
 * block-rook kernel behind PivotSearchStrategy::Rook (matrixluci/block_rook.rs:71-190, factors.rs:43-113).
 * fill_block(ctx, rows, nrows, cols, ncols, out) must write out[i + nrows*j] = A[rows[i], cols[j]]
 * (matrixluci/source.rs:15-24); this backend only ever asks for one full column or one full row per call and
 * never for the whole matrix.  Output buffers as for t4a_gpu_luci_f64. */
typedef void (*t4a_gpu_fill_block_fn)(void* ctx, const size_t* rows, size_t nrows, const size_t* cols, size_t ncols,
                                      double* out);
t4a_gpu_status t4a_gpu_luci_blocks_f64(size_t m, size_t n, t4a_gpu_fill_block_fn fill_block, void* ctx,
                                       size_t max_bond_dim, double rel_tol, double abs_tol, int32_t left_orthogonal,
                                       size_t* rank, size_t* rows, size_t* cols, double* pivot_errors, double* left,
                                       double* right);
/* Same kernel on a dense column-major matrix that is already in memory (LazyBlockRookKernel on a dense source,
 * matrixluci/block_rook/tests.rs:36-52). */
t4a_gpu_status t4a_gpu_luci_rook_f64(const double* a, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                     double abs_tol, int32_t left_orthogonal, size_t* rank, size_t* rows, size_t* cols,
                                     double* pivot_errors, double* left, double* right);

/* mat_mul(&a,&b) (tensorbackend/src/matrix.rs:1488): c[m x n] = a[m x k] * b[k x n] */
t4a_gpu_status t4a_gpu_gemm_f64(const double* a, const double* b, size_t m, size_t k, size_t n, double* c);

/* batched_mat_mul_same_shape(batch,m,k,n,&a,&b) (matrix.rs:1538-1612): operands are column-major
 * [m,k,batch] and [k,n,batch] (batch slowest), result [m,n,batch]. */
t4a_gpu_status t4a_gpu_gemm_batched_f64(size_t batch, size_t m, size_t k, size_t n, const double* a,
                                        const double* b, double* c);

/* triangular_solve_matrix(A,B,left_side,lower,transpose_a,unit_diagonal) (tensorbackend/src/backend.rs:924):
 * solves op(A) X = B (left_side) or X op(A) = B.  A is na x na, B and X are bm x bn. */
t4a_gpu_status t4a_gpu_trsm_f64(const double* a, size_t na, const double* b, size_t bm, size_t bn,
                                int32_t left_side, int32_t lower, int32_t transpose_a, int32_t unit_diagonal,
                                double* x);

/* solve_matrix(&A,&B) (backend.rs:865): A n x n, B n x nrhs, X n x nrhs.  Partial-pivot LU.
 * Returns T4A_GPU_SINGULAR_MATRIX for an exactly singular A. */
t4a_gpu_status t4a_gpu_solve_f64(const double* a, size_t n, const double* b, size_t nrhs, double* x);

/* svd_backend(&a) (tensorbackend/src/backend.rs:709-731): thin SVD, k = min(m,n); u is m x k, s has k entries in
 * non-increasing order, vt is k x n ("backend convention", simplett/src/compression.rs:254-287).  One-sided Jacobi
 * (from 64 columns on: on the transposed triangular factor of a Householder QR); a column pair rotates while the cosine of its
 * angle exceeds sqrt(rows) * eps (LAPACK dgesvj's rule): reconstruction and singular values to ~1e-14 relative to the largest.
 * Returns T4A_GPU_INVALID_ARGUMENT for an empty or non-finite matrix. */
t4a_gpu_status t4a_gpu_svd_f64(const double* a, size_t m, size_t n, double* u, double* s, double* vt);

/* qr_backend(&a) (backend.rs:742-760): thin QR, q is m x k, r is k x n upper trapezoidal.  Householder. */
t4a_gpu_status t4a_gpu_qr_f64(const double* a, size_t m, size_t n, double* q, double* r);

/* Randomized rank-k SVD (north_star: "one-sided Jacobi / randomized SVD"; Halko, Martinsson, Tropp 2011, algorithms 4.4 + 5.1 — the
 * reference itself only has the dense svd_backend above, so this is an ADDITION for truncated factorisations, not a replacement):
 * Y = A Omega with Omega n x (k + oversample) standard normal (drawn from the library's StdRng stream seeded with `seed`), power_iters
 * rounds of Y <- A (A^T Q(Y)) with a QR in between, Q = qr(Y), B = Q^T A, thin SVD of the small B by the one-sided Jacobi, U = Q U_B.
 * Everything between the upload of A and the download of the factors runs on the device (f64-MFMA GEMMs, Householder QR, Jacobi).
 * u is m x k, s has k entries (non-increasing), vt is k x n.  For a matrix of rank <= k the result is the thin SVD to rounding; for a
 * decaying spectrum the error is bounded by the usual (1 + 9 sqrt(k + p) sqrt(min(m, n))) sigma_{k+1} of the randomized range finder.
 * INVALID_ARGUMENT for k == 0, k + oversample > min(m, n) is clamped to min(m, n). */
t4a_gpu_status t4a_gpu_rsvd_f64(const double* a, size_t m, size_t n, size_t k, size_t oversample, size_t power_iters, uint64_t seed,
                                double* u, double* s, double* vt);

/* full_piv_lu_matrix(&a) (backend.rs:1022-1037) for a square n x n matrix: P A Q^T = L U with n x n factors;
 * row k of P (Q) carries its 1 in the column of the k-th pivot row (column), which is how
 * core/src/matrixluci/dense.rs:120-139 reads them.  Elimination order = rrlu_mut with zero tolerances. */
t4a_gpu_status t4a_gpu_full_piv_lu_f64(const double* a, size_t n, double* p, double* l, double* u, double* q);

/* =====================================================================================
 * TCI2 sweep driver (opaque handle; state resident on the device between calls)
 * ===================================================================================== */
typedef struct t4a_gpu_tci2 t4a_gpu_tci2;

/* TCI2Options (tensorci/src/tensorci2.rs:73-170).  max_bond_dim == 0 <=> None. */
typedef struct t4a_gpu_tci2_options {
    double tolerance;                 /* 1e-8 */
    size_t max_iter;                  /* 20 */
    size_t max_bond_dim;              /* 0 = None */
    int32_t pivot_search;             /* 0 = Full, 1 = Rook (lazy block-rook search, tensorci2.rs:286-296) */
    int32_t normalize_error;          /* 1 */
    size_t verbosity;                 /* 0 */
    size_t max_nglobal_pivot;         /* 5 */
    size_t nsearch;                   /* 5 */
    int32_t sweep_strategy;           /* 0 Forward, 1 Backward, 2 BackAndForth (default) */
    int32_t strictly_nested;          /* 0 */
    size_t ncheck_history;            /* 3 */
    double tol_margin_global_search;  /* 10.0 */
    int32_t has_seed;                 /* 0 = None */
    int32_t reserved_;
    uint64_t seed;
} t4a_gpu_tci2_options;

/* Fills `opts` with TCI2Options::default() (tensorci2.rs:152-170). */
t4a_gpu_status t4a_gpu_tci2_options_default(t4a_gpu_tci2_options* opts);

/* TCI2Termination (tensorci2.rs:197-205) */
#define T4A_GPU_TCI2_CONVERGED 0
#define T4A_GPU_TCI2_MAX_BOND_DIMENSION 1
#define T4A_GPU_TCI2_MAX_ITERATIONS 2

/* TensorCI2::new(local_dims) (tensorci2.rs:380-404) */
t4a_gpu_status t4a_gpu_tci2_new(const size_t* local_dims, size_t n_sites, t4a_gpu_tci2** out);
void t4a_gpu_tci2_release(t4a_gpu_tci2* h);

/* The user function `f` of crossinterpolate2 (tensorci2.rs:1513-1524), given either as
 *  (a) a built-in device functor: function id + parameters + integer weight table
 *      (include/t4a_testfunctions.h; weights has n_acc * sum(local_dims) entries), evaluated on the GPU; or
 *  (b) a host batch callback, the `batched_f` contract of the reference: one value per requested index.
 *      idx is n_sites x n_pts column-major (same shape as treetci's GlobalIndexBatch,
 *      treetci/src/update.rs:213-232); returns the number of values written (must equal n_pts),
 *      or a negative value to abort. */
t4a_gpu_status t4a_gpu_tci2_set_builtin_function(t4a_gpu_tci2* h, int32_t fid, int32_t n_acc,
                                                 const double* params /* [T4A_FN_MAX_PARAMS] */,
                                                 const uint64_t* weights);
typedef int64_t (*t4a_gpu_batch_eval_fn)(void* ctx, const uint32_t* idx, size_t n_sites, size_t n_pts, double* out);
t4a_gpu_status t4a_gpu_tci2_set_callback(t4a_gpu_tci2* h, t4a_gpu_batch_eval_fn cb, void* ctx);

/* TensorCI2::add_global_pivots (tensorci2.rs:668-711). pivots: n_sites x n_pivots column-major. */
t4a_gpu_status t4a_gpu_tci2_add_global_pivots(t4a_gpu_tci2* h, const size_t* pivots, size_t n_pivots);

/* crossinterpolate2(f, batched_f, local_dims, initial_pivots, options) (tensorci2.rs:1513-1563) on a fresh
 * handle: add initial pivots (n_pivots == 0 -> the all-zeros index), initialise max_sample_value, run
 * optimize_with_finder (tensorci2.rs:1626-1802) including the final sweep1site. */
t4a_gpu_status t4a_gpu_tci2_crossinterpolate2(t4a_gpu_tci2* h, const size_t* initial_pivots, size_t n_pivots,
                                              const t4a_gpu_tci2_options* options);

/* optimize_with_finder on the current state (tensorci2.rs:1626).  `final_sweep1site == 0` skips the final
 * cleanup sweep (tensorci2.rs:1787) so that callers (bench) can time individual half-sweeps. */
t4a_gpu_status t4a_gpu_tci2_optimize(t4a_gpu_tci2* h, const t4a_gpu_tci2_options* options, int32_t final_sweep1site);

/* TensorCI2::sweep2site(&f,&batched_f,forward,&options) (tensorci2.rs:746-798) */
t4a_gpu_status t4a_gpu_tci2_sweep2site(t4a_gpu_tci2* h, int32_t forward, const t4a_gpu_tci2_options* options);
/* TensorCI2::sweep1site(&f,forward,rel_tol,abs_tol,max_bond_dim,update_tensors) (tensorci2.rs:865-915) */
t4a_gpu_status t4a_gpu_tci2_sweep1site(t4a_gpu_tci2* h, int32_t forward, double rel_tol, double abs_tol,
                                       size_t max_bond_dim /* 0 = usize::MAX */, int32_t update_tensors);
/* TensorCI2::fill_site_tensors(&f) (tensorci2.rs:1065-1186) */
t4a_gpu_status t4a_gpu_tci2_fill_site_tensors(t4a_gpu_tci2* h);
/* TensorCI2::make_canonical (tensorci2.rs:1201-1221) */
t4a_gpu_status t4a_gpu_tci2_make_canonical(t4a_gpu_tci2* h, double rel_tol, double abs_tol, size_t max_bond_dim);

/* accessors (tensorci2.rs:585-649, 713-721) */
t4a_gpu_status t4a_gpu_tci2_len(const t4a_gpu_tci2* h, size_t* out);
t4a_gpu_status t4a_gpu_tci2_rank(const t4a_gpu_tci2* h, size_t* out);
t4a_gpu_status t4a_gpu_tci2_link_dims(const t4a_gpu_tci2* h, size_t* out /* n_sites-1 */);
t4a_gpu_status t4a_gpu_tci2_max_sample_value(const t4a_gpu_tci2* h, double* out);
t4a_gpu_status t4a_gpu_tci2_max_bond_error(const t4a_gpu_tci2* h, double* out);
t4a_gpu_status t4a_gpu_tci2_bond_errors(const t4a_gpu_tci2* h, double* out /* n_sites-1 */);
/* query-then-fill: out == NULL returns the count only */
t4a_gpu_status t4a_gpu_tci2_pivot_errors(const t4a_gpu_tci2* h, size_t* count, double* out);
/* which: 0 = i_set(site), 1 = j_set(site); entries are `width` digits each, returned as
 * width x count column-major (one multi-index per column). out == NULL queries count/width. */
t4a_gpu_status t4a_gpu_tci2_index_set(const t4a_gpu_tci2* h, int32_t which, size_t site, size_t* count,
                                      size_t* width, size_t* out);
/* TensorCI2::from_index_sets resume format (tensorci2.rs:551-582): overwrite one I/J set. */
t4a_gpu_status t4a_gpu_tci2_set_index_set(t4a_gpu_tci2* h, int32_t which, size_t site, size_t count,
                                          const size_t* data);
t4a_gpu_status t4a_gpu_tci2_set_max_sample_value(t4a_gpu_tci2* h, double value);
t4a_gpu_status t4a_gpu_tci2_clear_history(t4a_gpu_tci2* h);
/* site_tensor(p): dims3 = (left, site, right); column-major [left, site, right] like simplett Tensor3
 * (simplett/src/tensor.rs:13-18). out == NULL queries the dims only. */
t4a_gpu_status t4a_gpu_tci2_site_tensor(const t4a_gpu_tci2* h, size_t site, size_t* dims3, double* out);
/* Same tensor copied device-to-device into caller-provided device memory (used for the RCCL all-gather
 * of cores; `out_device` must hold left*site*right doubles). */
t4a_gpu_status t4a_gpu_tci2_site_tensor_device(const t4a_gpu_tci2* h, size_t site, void* out_device);
/* Overwrite a site tensor from device memory (receiving side of the core all-gather). */
t4a_gpu_status t4a_gpu_tci2_set_site_tensor_device(t4a_gpu_tci2* h, size_t site, const size_t* dims3,
                                                   const void* in_device);

/* results of the last optimize / crossinterpolate2 call (TCI2OptimizationResult, tensorci2.rs:236-245) */
t4a_gpu_status t4a_gpu_tci2_n_iterations(const t4a_gpu_tci2* h, size_t* out);
t4a_gpu_status t4a_gpu_tci2_history(const t4a_gpu_tci2* h, size_t* ranks, double* errors);
t4a_gpu_status t4a_gpu_tci2_termination(const t4a_gpu_tci2* h, int32_t* out);

/* to_tensor_train().evaluate(idx) for a batch of points (simplett/src/traits.rs:146-212), evaluated on
 * the device.  idx: n_sites x n_pts column-major. */
t4a_gpu_status t4a_gpu_tci2_evaluate(t4a_gpu_tci2* h, const size_t* idx, size_t n_pts, double* out);
/* to_tensor_train().sum() (traits.rs:231-275) */
t4a_gpu_status t4a_gpu_tci2_sum(t4a_gpu_tci2* h, double* out);

/* Restrict fill_site_tensors to the sites s with s % world == rank (site-sharded multi-GPU mode,
 * SURVEY.md §8e).  world == 1 restores the default.  Cores of foreign sites are left untouched and are
 * expected to arrive through t4a_gpu_tci2_set_site_tensor_device. */
t4a_gpu_status t4a_gpu_tci2_set_site_shard(t4a_gpu_tci2* h, size_t rank, size_t world);

/* Column-block shard of the candidate matrix for HOST-CALLBACK functions (entries of one candidate matrix are
 * independent, tensorci2.rs:1859-1893).  After set_pi_shard(rank, world, gather, ctx) every matrix this handle evaluates through its
 * batch callback (update_pivots, sweep1site, fill_site_tensors) is split into `world` blocks of ceil(N / world) columns: this rank's
 * callback sees only the points of its block (row index outer, column index inner: the order of tensorci2.rs:1862-1869 restricted to
 * the block), `gather` — an all-gather over the process group on HOST buffers: `count` doubles in from every rank, world * count out,
 * rank-major; 0 = success — assembles the whole matrix on every rank, and the rank-revealing LU runs replicated (deterministic: the same
 * pivots everywhere, nothing is broadcast).  Every rank must drive its handle through the same calls.  world == 1 switches it off. */
typedef int32_t (*t4a_gpu_allgather_fn)(void* ctx, const double* send, size_t count, double* recv);
t4a_gpu_status t4a_gpu_tci2_set_pi_shard(t4a_gpu_tci2* h, size_t rank, size_t world, t4a_gpu_allgather_fn gather, void* ctx);
/* [n_gathers, bytes_sent_by_this_rank] since the shard was set. */
t4a_gpu_status t4a_gpu_tci2_pi_shard_stats(t4a_gpu_tci2* h, size_t* out2);
/* The host part of the shard alone (no device, no handle): evaluates the na x nb matrix of the row halves `a` (na x wa digits, placed at
 * site a0) and the column halves `b` (nb x wb digits at site b0; wa + wb = n_sites) through `cb` + `gather` exactly as a sharded handle
 * does; out: na x nb row-major.  world == 1: one plain callback over all points.  (CPU-only test hook, tests/test_cpu_parallel.py.) */
t4a_gpu_status t4a_gpu_pi_shard_eval(size_t rank, size_t world, t4a_gpu_batch_eval_fn cb, void* cb_ctx, t4a_gpu_allgather_fn gather,
                                     void* gather_ctx, size_t n_sites, const uint32_t* a, size_t wa, size_t a0, size_t na,
                                     const uint32_t* b, size_t wb, size_t b0, size_t nb, double* out);

/* add_global_pivots invalidates the site tensors even when the pivot list is empty (tensorci2.rs:707-708), so
 * after optimize_with_finder without a final sweep1site the cores of the last fill_site_tensors are gone.  With
 * keep != 0 an EMPTY pivot list leaves them in place (the index sets did not change, so they are still the cores
 * of the current sets).  Default 0 = reference behaviour. */
t4a_gpu_status t4a_gpu_tci2_set_keep_site_tensors(t4a_gpu_tci2* h, int32_t keep);

/* Multi-GPU core gather without a host stall: copies every site tensor to dst_device + site * stride (stride in
 * doubles, >= the largest site tensor) on the stream of the fill_site_tensors that may still be in flight, and makes
 * `consumer_stream` (a hipStream_t, e.g. the stream an RCCL all-gather is issued from) wait for the copies.  With
 * keep != 0 (above) optimize() leaves its last fill_site_tensors in flight, so the next sweep's bond updates overlap
 * with it; errors of that fill are reported by the next call that reads a site tensor. */
t4a_gpu_status t4a_gpu_tci2_export_site_tensors_async(t4a_gpu_tci2* h, void* dst_device, size_t stride,
                                                      void* consumer_stream);
/* Site-sharded fill_site_tensors (BASELINE.json configs[3]; tensorci2.rs:1065-1186: sites are independent given the I/J
 * sets) without host staging.  After t4a_gpu_tci2_set_site_shard(rank, world):
 *   export: the local sites s = rank + world * k go to dst_device + k * stride (doubles), ordered after the fill that may
 *           still be in flight; `consumer_stream` (the stream the RCCL all-gather is issued from) waits for the copies;
 *   import: the gathered buffer is [world][per_rank][stride]; every remote site is copied into this handle (shapes follow
 *           from the replicated index sets) on the handle's stream, after what `producer_stream` has enqueued so far. */
t4a_gpu_status t4a_gpu_tci2_export_site_shard_async(t4a_gpu_tci2* h, void* dst_device, size_t stride, void* consumer_stream);
t4a_gpu_status t4a_gpu_tci2_import_site_shard_async(t4a_gpu_tci2* h, const void* src_device, size_t stride, size_t per_rank,
                                                    void* producer_stream);

/* =====================================================================================
 * SimpleTensorTrain<f64> (opaque handle; site tensors resident on the device)
 * tensor4all-simplett: tensortrain.rs:97, traits.rs:75-355, compression.rs:375-507, cache.rs:558-744
 * ===================================================================================== */
typedef struct t4a_gpu_tt t4a_gpu_tt;

/* SimpleTensorTrain::new(tensors) (tensortrain.rs:97-124).  dims3 is 3 x n_sites (left, site, right per site),
 * cores the site tensors concatenated, each column-major [left, site, right].  Same validation as the reference:
 * first left == 1, last right == 1, neighbouring bonds equal. */
t4a_gpu_status t4a_gpu_tt_new(const size_t* dims3, size_t n_sites, const double* cores, t4a_gpu_tt** out);
void t4a_gpu_tt_release(t4a_gpu_tt* h);
t4a_gpu_status t4a_gpu_tt_clone(const t4a_gpu_tt* h, t4a_gpu_tt** out);
t4a_gpu_status t4a_gpu_tt_len(const t4a_gpu_tt* h, size_t* out);
t4a_gpu_status t4a_gpu_tt_dims(const t4a_gpu_tt* h, size_t* dims3 /* 3 x n_sites */);
t4a_gpu_status t4a_gpu_tt_site_tensor(const t4a_gpu_tt* h, size_t site, double* out);
/* AbstractTensorTrain::evaluate (traits.rs:146-212) for a batch: idx is n_sites x n_pts column-major. */
t4a_gpu_status t4a_gpu_tt_evaluate(t4a_gpu_tt* h, const size_t* idx, size_t n_pts, double* out);
/* sum (traits.rs:231-275), norm2 = <tt|tt> (traits.rs:289-354) */
t4a_gpu_status t4a_gpu_tt_sum(t4a_gpu_tt* h, double* out);
t4a_gpu_status t4a_gpu_tt_norm2(t4a_gpu_tt* h, double* out);
/* SimpleTensorTrain::compress(&CompressionOptions) (compression.rs:375-507).
 * method: 0 LU, 1 CI, 2 SVD (compression.rs:40-52); max_bond_dim == 0 <=> None. */
#define T4A_GPU_COMPRESS_LU 0
#define T4A_GPU_COMPRESS_CI 1
#define T4A_GPU_COMPRESS_SVD 2
t4a_gpu_status t4a_gpu_tt_compress(t4a_gpu_tt* h, int32_t method, double tolerance, size_t max_bond_dim,
                                   int32_t normalize_error);
/* TTCache::evaluate_many(indices, split) (cache.rs:558-688): split == 0 <=> None (find_split_heuristic,
 * cache.rs:690-744); *used_split (may be NULL) reports the split position that was used. */
t4a_gpu_status t4a_gpu_tt_evaluate_many(t4a_gpu_tt* h, const size_t* idx, size_t n_pts, size_t split, double* out,
                                        size_t* used_split);

/* floating_zone(tt, f, local_dims, init_p, early_stop_tol) (tensorci/src/globalsearch.rs:163-243, walk:
 * tensor4all-core/src/floating_zone.rs:46-103): local search for the multi-index with the largest |f - tt|.  The tensor
 * train is evaluated on the device (TTCache::evaluate_many per site scan), f is the batch callback.  init_p == NULL: random
 * starting point (reference: thread rng of rand 0.9, "parity unpinned"; here splitmix64(seed)). */
t4a_gpu_status t4a_gpu_tt_floating_zone(t4a_gpu_tt* tt, t4a_gpu_batch_eval_fn f, void* ctx, const size_t* local_dims, size_t n_sites,
                                        const size_t* init_p, uint64_t seed, double early_stop_tol, size_t* pivot_out /* n_sites */,
                                        double* error_out);
/* estimate_true_error(tt, f, nsearch, initial_points, rng) (globalsearch.rs:70-118): floating_zone from every starting point,
 * results sorted by descending error, consecutive duplicates removed.  initial_points: n_sites x n_initial column-major or
 * NULL (then nsearch random points from StdRng::seed_from_u64(seed), csrc/stdrng.hpp).  Query-then-fill: *n_out is always set; BUFFER_TOO_SMALL when
 * capacity < *n_out (at most nsearch resp. n_initial results). */
t4a_gpu_status t4a_gpu_tt_estimate_true_error(t4a_gpu_tt* tt, t4a_gpu_batch_eval_fn f, void* ctx, size_t nsearch, const size_t* initial_points,
                                              size_t n_initial, uint64_t seed, size_t* pivots_out /* n_sites x capacity */,
                                              double* errors_out /* capacity */, size_t capacity, size_t* n_out);
/* opt_first_pivot(f, local_dims, first_pivot, max_sweep) (tensorci/src/optfirstpivot.rs:40-74): greedy coordinate search for
 * a large |f|; host logic over the batch callback (one batch per site scan), no device work. */
t4a_gpu_status t4a_gpu_opt_first_pivot(t4a_gpu_batch_eval_fn f, void* ctx, const size_t* local_dims, size_t n_sites, const size_t* first_pivot,
                                       size_t max_sweep, size_t* pivot_out /* n_sites */);

/* TensorCI2::to_tensor_train (tensorci2.rs:640-660): a device-to-device copy of the current site tensors. */
t4a_gpu_status t4a_gpu_tci2_to_tensor_train(t4a_gpu_tci2* h, t4a_gpu_tt** out);
/* TensorCI2::from_tensor_train(tt, TensorCI2FromTensorTrainOptions{tolerance, max_bond_dim, max_iter})
 * (tensorci/src/conversion.rs:66-121).  max_bond_dim == 0 <=> None; defaults 1e-12 / None / 3. */
t4a_gpu_status t4a_gpu_tci2_from_tensor_train(const t4a_gpu_tt* tt, double tolerance, size_t max_bond_dim,
                                              size_t max_iter, t4a_gpu_tci2** out);

/* =====================================================================================
 * Adaptive patching driver (BASELINE config 5): partitionedtt::adaptiveinterpolate
 * crates/tensor4all-partitionedtt/src/adaptive_interpolation.rs:58-262
 * ===================================================================================== */
typedef struct t4a_gpu_ptt t4a_gpu_ptt;

/* adaptiveinterpolate(f, batched_f, site_indices, initial_pivots, AdaptiveInterpolateOptions{tci_options, patch_order,
 * n_initial_pivots, recycle_pivots}).  Sites are identified by their position; `patch_order` is a permutation of
 * 0..n_sites-1 or NULL for the natural order (adaptive_interpolation.rs:336-354).  initial_pivots is n_sites x n_pivots
 * column-major.  One crossinterpolate2 runs on the device per patch; a patch is accepted when it terminated Converged
 * with final error <= tolerance (:357-359), otherwise it is split along the next unprojected site of patch_order.
 * The built-in variant restricts the integer weight tables of the device functor to the active sites of every patch. */
t4a_gpu_status t4a_gpu_adaptive_interpolate_builtin(const size_t* local_dims, size_t n_sites, int32_t fid, int32_t n_acc,
                                                    const double* params, const uint64_t* weights,
                                                    const size_t* initial_pivots, size_t n_pivots,
                                                    const t4a_gpu_tci2_options* tci_options, const size_t* patch_order,
                                                    size_t n_initial_pivots, int32_t recycle_pivots, t4a_gpu_ptt** out);
t4a_gpu_status t4a_gpu_adaptive_interpolate_callback(const size_t* local_dims, size_t n_sites, t4a_gpu_batch_eval_fn cb,
                                                     void* ctx, const size_t* initial_pivots, size_t n_pivots,
                                                     const t4a_gpu_tci2_options* tci_options, const size_t* patch_order,
                                                     size_t n_initial_pivots, int32_t recycle_pivots, t4a_gpu_ptt** out);
void t4a_gpu_ptt_release(t4a_gpu_ptt* h);
/* PartitionedTT::len: number of accepted patches (FIFO acceptance order). */
t4a_gpu_status t4a_gpu_ptt_len(const t4a_gpu_ptt* h, size_t* out);
/* Projector of patch k: `count` projected sites, positions ascending; positions / values may be NULL to query. */
t4a_gpu_status t4a_gpu_ptt_projector(const t4a_gpu_ptt* h, size_t k, size_t* count, size_t* positions, size_t* values);
/* SubDomainTT of patch k as a tensor train over ALL sites (projected sites carry copy-selector tensors,
 * adaptive_interpolation.rs:514-660); the new handle owns a device copy. */
t4a_gpu_status t4a_gpu_ptt_patch_tt(const t4a_gpu_ptt* h, size_t k, t4a_gpu_tt** out);
/* Sum over the patches at a batch of full multi-indices (idx: n_sites x n_pts column-major). */
t4a_gpu_status t4a_gpu_ptt_evaluate(t4a_gpu_ptt* h, const size_t* idx, size_t n_pts, double* out);

/* =====================================================================================
 * Tree tensor cross interpolation: tensor4all-treetci
 * crates/tensor4all-treetci/src/{graph,state,proposer,update,optimize,materialize,globalpivot,api}.rs
 * The batch evaluator of that crate, Fn(GlobalIndexBatch) with a column-major (n_sites, n_points) index buffer
 * (batch.rs:13-66, update.rs:213-232), IS t4a_gpu_batch_eval_fn.
 * ===================================================================================== */
typedef struct t4a_gpu_treetci t4a_gpu_treetci;

/* TreeTciOptions (optimize.rs:13-76).  max_bond_dim == 0 <=> None, has_seed == 0 <=> seed None. */
typedef struct t4a_gpu_treetci_options {
    double tolerance;                 /* 1e-8 */
    size_t max_iter;                  /* 20 */
    size_t max_bond_dim;              /* 0 */
    int32_t normalize_error;          /* 1 */
    int32_t enable_global_pivots;     /* 1 */
    size_t nsearch;                   /* 5 */
    size_t max_nglobal_pivot;         /* 5 */
    double tol_margin_global_search;  /* 10.0 */
    int32_t has_seed;                 /* 0 */
    uint64_t seed;
} t4a_gpu_treetci_options;
t4a_gpu_status t4a_gpu_treetci_options_default(t4a_gpu_treetci_options* opts);

/* TreeTciGraph::new(n_sites, edges) + TreeTCI2::new(local_dims, graph) (graph.rs:51-106, state.rs:66-103).
 * edges: 2 * n_edges site numbers (u0, v0, u1, v1, ...); the graph must be a tree. */
t4a_gpu_status t4a_gpu_treetci_new(const size_t* local_dims, size_t n_sites, const size_t* edges, size_t n_edges,
                                   t4a_gpu_treetci** out);
void t4a_gpu_treetci_release(t4a_gpu_treetci* h);
/* function source: as t4a_gpu_tci2_set_builtin_function / _set_callback */
t4a_gpu_status t4a_gpu_treetci_set_builtin_function(t4a_gpu_treetci* h, int32_t fid, int32_t n_acc, const double* params,
                                                    const uint64_t* weights);
t4a_gpu_status t4a_gpu_treetci_set_callback(t4a_gpu_treetci* h, t4a_gpu_batch_eval_fn cb, void* ctx);
/* TreeTCI2::add_global_pivots (state.rs:110-165): pivots is n_sites x n_pivots column-major. */
t4a_gpu_status t4a_gpu_treetci_add_global_pivots(t4a_gpu_treetci* h, const size_t* pivots, size_t n_pivots);
/* graph.rs:150-156 subregion_vertices(edge): the sorted site lists on the u side and on the v side (query with NULL). */
t4a_gpu_status t4a_gpu_treetci_subregion_vertices(const t4a_gpu_treetci* h, size_t u, size_t v, size_t* n_left,
                                                  size_t* left, size_t* n_right, size_t* right);
/* Proposer used by _candidates / _update_edge / _optimize / _crossinterpolate2 (proposer.rs:43-249): 0 DefaultProposer
 * (neighbour product), 1 SimpleProposer::seeded(seed) (d*chi random candidates per side), 2
 * TruncatedDefaultProposer::seeded(seed) (ordered sample of d*chi default candidates per side — keeps the matrices of
 * branching vertices at d*chi x d*chi).  The reference draws from rand SmallRng seeded through std DefaultHasher; the
 * stream here is splitmix64 ("parity unpinned"). */
t4a_gpu_status t4a_gpu_treetci_set_proposer(t4a_gpu_treetci* h, int32_t kind, uint64_t seed);
/* PivotCandidateProposer::candidates of the selected proposer (default: proposer.rs:57-88): left is (|left key| x n_left), right (|right key| x n_right),
 * both column-major; pass NULL buffers to query the counts. */
t4a_gpu_status t4a_gpu_treetci_candidates(const t4a_gpu_treetci* h, size_t u, size_t v, size_t* n_left, size_t* left,
                                          size_t* n_right, size_t* right);
/* update_edge with the default proposer (update.rs:22-115): candidate matrix on the device -> full-pivot rrLU ->
 * new pivot tables of both sides, bond error, pivot errors, max_sample_value.  rows / cols (>= min(n_left, n_right)
 * entries) receive the selected candidate numbers, pivot_errors rank + 1 values; all three may be NULL. */
t4a_gpu_status t4a_gpu_treetci_update_edge(t4a_gpu_treetci* h, size_t u, size_t v, size_t max_bond_dim, double rel_tol,
                                           double abs_tol, size_t* rank, size_t* rows, size_t* cols, double* pivot_errors);
/* optimize_default (optimize.rs:80-220) / crossinterpolate2 with the default proposer (api.rs:21-96; the tree network
 * is obtained afterwards with _materialize).  ranks / errors need max_iter entries; n_iter receives the sweep count. */
t4a_gpu_status t4a_gpu_treetci_optimize(t4a_gpu_treetci* h, const t4a_gpu_treetci_options* options, size_t* n_iter,
                                        size_t* ranks, double* errors);
t4a_gpu_status t4a_gpu_treetci_crossinterpolate2(t4a_gpu_treetci* h, const size_t* initial_pivots, size_t n_pivots,
                                                 const t4a_gpu_treetci_options* options, size_t* n_iter, size_t* ranks,
                                                 double* errors);
/* find_global_pivots (globalpivot.rs:24-172): out is n_sites x count column-major, at most max_nglobal_pivot columns. */
t4a_gpu_status t4a_gpu_treetci_find_global_pivots(t4a_gpu_treetci* h, size_t nsearch, size_t max_nglobal_pivot,
                                                  double tol_margin, double abs_tol, uint64_t seed, size_t* count,
                                                  size_t* out);
/* state accessors (state.rs:41-58, :168-199).  key: sorted site list of a subtree; out is |key| x count column-major. */
t4a_gpu_status t4a_gpu_treetci_pivots(const t4a_gpu_treetci* h, const size_t* key, size_t key_len, size_t* count,
                                      size_t* out);
t4a_gpu_status t4a_gpu_treetci_bond_errors(const t4a_gpu_treetci* h, double* out /* per edge, sorted edge order */);
t4a_gpu_status t4a_gpu_treetci_pivot_errors(const t4a_gpu_treetci* h, size_t* count, double* out);
t4a_gpu_status t4a_gpu_treetci_flush_pivot_errors(t4a_gpu_treetci* h);
t4a_gpu_status t4a_gpu_treetci_max_sample_value(const t4a_gpu_treetci* h, double* out);
t4a_gpu_status t4a_gpu_treetci_set_max_sample_value(t4a_gpu_treetci* h, double value);
t4a_gpu_status t4a_gpu_treetci_max_bond_error(const t4a_gpu_treetci* h, double* out);
t4a_gpu_status t4a_gpu_treetci_max_bond_dim(const t4a_gpu_treetci* h, size_t* out);
/* to_treetn(state, evaluate, center_site) (materialize.rs:17-166): builds one dense tensor per site on the device,
 * index order [site, incoming bonds (sorted neighbours except the parent), bond to the parent]; non-root sites solve
 * T * P = Pi1 by full-pivot LU, a numerically zero P gives a zero tensor. */
t4a_gpu_status t4a_gpu_treetci_materialize(t4a_gpu_treetci* h, size_t center_site);
/* ndims / dims (<= n_sites + 1 entries) / column-major data of one materialised site tensor; out may be NULL. */
t4a_gpu_status t4a_gpu_treetci_site_tensor(t4a_gpu_treetci* h, size_t site, size_t* ndims, size_t* dims, double* out);
/* TreeTN::evaluate of the materialised network at full multi-indices (idx: n_sites x n_pts column-major). */
t4a_gpu_status t4a_gpu_treetci_evaluate(t4a_gpu_treetci* h, const size_t* idx, size_t n_pts, double* out);

/* =====================================================================================
 * Quantics front end: tensor4all-quanticstci
 * crates/tensor4all-quanticstci/src/{options.rs,quantics_tci.rs}
 * Grid conventions (crate quanticsgrids @ 8214b72, un-vendored; restated from its published algorithm): R_d bits per
 * variable, most significant bit first, 0-based grid indices; unfolding 0 = Interleaved (one binary site per bit level
 * and variable), 1 = Fused (one site per bit level, value = sum_d bit_d * 2^d, first variable least significant).
 * ===================================================================================== */
typedef struct t4a_gpu_qtci t4a_gpu_qtci;
/* f: Fn(&[f64]) -> f64 evaluated for a batch: coords is n_vars x n_pts column-major; returns the number of values written */
typedef int64_t (*t4a_gpu_coord_eval_fn)(void* ctx, const double* coords, size_t n_vars, size_t n_pts, double* out);
/* f: Fn(&[usize]) -> f64 on 0-based grid indices, same batch layout */
typedef int64_t (*t4a_gpu_grididx_eval_fn)(void* ctx, const size_t* grididx, size_t n_vars, size_t n_pts, double* out);

/* QtciOptions (options.rs:9-45).  max_bond_dim == 0 <=> None.  The random initial pivots use rand::rng() in the
 * reference; here a fixed stream unless has_seed is set. */
typedef struct t4a_gpu_qtci_options {
    double tolerance;              /* 1e-8 */
    size_t max_bond_dim;           /* 0 */
    size_t max_iter;               /* 200 */
    size_t n_random_init_pivot;    /* 5 */
    int32_t unfolding_scheme;      /* 0 Interleaved */
    int32_t normalize_error;       /* 1 */
    int32_t has_seed;              /* 0 */
    uint64_t seed;
} t4a_gpu_qtci_options;
t4a_gpu_status t4a_gpu_qtci_options_default(t4a_gpu_qtci_options* opts);

/* quanticscrossinterpolate(&DiscretizedGrid, f, initial_pivots, options) (quantics_tci.rs:175-307).  The grid is given by
 * its builder arguments: bits per variable, bounds (NULL = 0 / 1), include_endpoint and the grid's own unfolding scheme.
 * initial_pivots: n_vars x n_pivots grid indices, column-major; has_pivots == 0 <=> None (first grid point).
 * Every distinct quantics point is evaluated once (cachedata); all misses of one candidate matrix reach `f` in one call. */
t4a_gpu_status t4a_gpu_quanticscrossinterpolate(const size_t* rs, size_t n_vars, const double* lower, const double* upper,
                                                int32_t include_endpoint, int32_t grid_unfolding, t4a_gpu_coord_eval_fn f,
                                                void* ctx, int32_t has_pivots, const size_t* initial_pivots, size_t n_pivots,
                                                const t4a_gpu_qtci_options* options, t4a_gpu_qtci** out);
/* quanticscrossinterpolate_discrete(size, f, initial_pivots, options) (:434-560): power-of-two sizes, equal in every
 * direction; unfolding from the options. */
t4a_gpu_status t4a_gpu_quanticscrossinterpolate_discrete(const size_t* sizes, size_t n_vars, t4a_gpu_grididx_eval_fn f,
                                                         void* ctx, int32_t has_pivots, const size_t* initial_pivots,
                                                         size_t n_pivots, const t4a_gpu_qtci_options* options,
                                                         t4a_gpu_qtci** out);
/* quanticscrossinterpolate_from_arrays(xvals, f, ..) (:309-432): xvals concatenated, sizes[d] values per variable;
 * uniform coordinates -> discretized grid with the end point included, otherwise coordinate lookup on an inherent grid. */
t4a_gpu_status t4a_gpu_quanticscrossinterpolate_from_arrays(const double* xvals, const size_t* sizes, size_t n_vars,
                                                            t4a_gpu_coord_eval_fn f, void* ctx, int32_t has_pivots,
                                                            const size_t* initial_pivots, size_t n_pivots,
                                                            const t4a_gpu_qtci_options* options, t4a_gpu_qtci** out);
void t4a_gpu_qtci_release(t4a_gpu_qtci* h);
/* QuanticsTensorCI2 accessors (:53-173) */
t4a_gpu_status t4a_gpu_qtci_evaluate(t4a_gpu_qtci* h, const size_t* grididx, size_t n_pts, double* out);
t4a_gpu_status t4a_gpu_qtci_sum(t4a_gpu_qtci* h, double* out);
t4a_gpu_status t4a_gpu_qtci_integral(t4a_gpu_qtci* h, double* out);
t4a_gpu_status t4a_gpu_qtci_n_sites(const t4a_gpu_qtci* h, size_t* n_sites, size_t* n_vars, int32_t* is_discretized);
t4a_gpu_status t4a_gpu_qtci_link_dims(const t4a_gpu_qtci* h, size_t* out /* n_sites - 1 */);
t4a_gpu_status t4a_gpu_qtci_history(const t4a_gpu_qtci* h, size_t* n_iter, size_t* ranks, double* errors);
/* tensor_train(): a new handle owning a device copy of the cores */
t4a_gpu_status t4a_gpu_qtci_tensor_train(t4a_gpu_qtci* h, t4a_gpu_tt** out);
/* tci(): pivot table of the underlying TreeTCI2 for a subtree key */
t4a_gpu_status t4a_gpu_qtci_tree_pivots(const t4a_gpu_qtci* h, const size_t* key, size_t key_len, size_t* count, size_t* out);
/* cachedata(): count entries; quantics is n_sites x count column-major (NULL to query); user_calls / user_points (may
 * be NULL) report how often and for how many points the user function was called */
t4a_gpu_status t4a_gpu_qtci_cachedata(const t4a_gpu_qtci* h, size_t* count, size_t* quantics, double* values,
                                      size_t* user_calls, size_t* user_points);
/* grid conversions: which = 0 grididx -> quantics, 1 quantics -> grididx, 2 quantics -> origcoord (out_d),
 * 3 local_dimensions, 4 grid_step (out_d) */
t4a_gpu_status t4a_gpu_qtci_grid(const t4a_gpu_qtci* h, int32_t which, const size_t* in, size_t* out_u, double* out_d);

/* quanticscrossinterpolate_batched (quanticstci/src/batched/mod.rs:50-191): vector / tensor valued f.  The callback
 * writes out[c + n_components * p] for every point p and returns n_components * n_pts; every coordinate reaches it once
 * for all components.  The result is a tensor train with one extra (component selector) site of dimension
 * prod(output_dims) (combine_component_tts, :193-318); ranks / errors (max_iter entries) are the element-wise maxima over
 * the per-component runs. */
typedef int64_t (*t4a_gpu_coord_eval_vec_fn)(void* ctx, const double* coords, size_t n_vars, size_t n_pts,
                                             size_t n_components, double* out);
t4a_gpu_status t4a_gpu_quanticscrossinterpolate_batched(const size_t* rs, size_t n_vars, const double* lower,
                                                        const double* upper, int32_t include_endpoint, int32_t grid_unfolding,
                                                        t4a_gpu_coord_eval_vec_fn f, void* ctx, const size_t* output_dims,
                                                        size_t n_output_dims, int32_t has_pivots, const size_t* initial_pivots,
                                                        size_t n_pivots, const t4a_gpu_qtci_options* options,
                                                        t4a_gpu_tt** out_tt, size_t* n_iter, size_t* ranks, double* errors,
                                                        size_t* user_points);

/* =====================================================================================
 * Dense labelled tensors: the einsum / linalg seam of tensor4all-core's dynamic-index layer
 * crates/tensor4all-core/src/defaults/{contract.rs:334-343, svd.rs:150-395, qr.rs:74-328, idx_tensor.rs:5278-5345},
 * index_ops.rs:660-696.  Tensors are column-major with one int64 label per axis (the caller's DynIndex ids); rank <= 16.
 * ===================================================================================== */
/* SvdTruncationPolicy (truncation.rs:137-147).  NULL = the default policy: relative, per value, 1e-12 (svd.rs:80-87). */
typedef struct t4a_gpu_svd_policy {
    double threshold;
    int32_t scale;    /* 0 Relative, 1 Absolute */
    int32_t measure;  /* 0 Value, 1 SquaredValue */
    int32_t rule;     /* 0 PerValue, 1 DiscardedTailSum */
} t4a_gpu_svd_policy;
/* compute_retained_rank (svd.rs:150-211) and compute_retained_rank_qr_from_dense (qr.rs:74-117; r is k x n column-major):
 * pure host functions, usable without a device. */
t4a_gpu_status t4a_gpu_svd_retained_rank(const double* s, size_t n, const t4a_gpu_svd_policy* policy, size_t* out);
t4a_gpu_status t4a_gpu_qr_retained_rank(const double* r, size_t k, size_t n, double rtol, size_t* out);
/* contract_pair(lhs, rhs): every common label is contracted; the result carries lhs's free axes then rhs's free axes, each in
 * operand order (no common label = outer product).  out may be NULL to query out_rank / out_dims / out_labels. */
t4a_gpu_status t4a_gpu_tensor_contract_f64(const double* a, const size_t* a_dims, const int64_t* a_labels, size_t a_rank,
                                           const double* b, const size_t* b_dims, const int64_t* b_labels, size_t b_rank,
                                           double* out, size_t* out_dims, int64_t* out_labels, size_t* out_rank);
/* svd_with(t, left_inds, options): unfold with the left labels first (in the given order) and the remaining axes in tensor
 * order, thin SVD, rank r = min(retained rank, max_bond_dim) >= 1 (truncate == 0: r = min(m, n)).
 * u: [left dims.., r], s: r values, v: [right dims.., r] (= V, not V^H).  Capacities: m*min(m,n), min(m,n), n*min(m,n).
 * has_max_bond_dim != 0 with max_bond_dim == 0 is an error (svd.rs:281-292). */
t4a_gpu_status t4a_gpu_tensor_svd_f64(const double* t, const size_t* dims, const int64_t* labels, size_t rank,
                                      const int64_t* left_labels, size_t n_left, int32_t truncate,
                                      const t4a_gpu_svd_policy* policy, int32_t has_max_bond_dim, size_t max_bond_dim,
                                      size_t* r, double* u, double* s, double* v);
/* qr_with(t, left_inds, options): q: [left dims.., r], r_factor: [r, right dims..]; truncate != 0 keeps the leading r rows
 * with r = number of rows of R whose norm is >= rtol * (largest row norm) (has_rtol == 0: default 1e-15). */
t4a_gpu_status t4a_gpu_tensor_qr_f64(const double* t, const size_t* dims, const int64_t* labels, size_t rank,
                                     const int64_t* left_labels, size_t n_left, int32_t truncate, int32_t has_rtol, double rtol,
                                     size_t* r, double* q, double* r_factor);

/* Device-resident labelled tensors: the same three operations without a host round trip per call (environment /
 * zip-up style chains keep their intermediates in HBM).  A handle owns its column-major payload, dims and labels. */
typedef struct t4a_gpu_tensor t4a_gpu_tensor;
t4a_gpu_status t4a_gpu_tensor_new(const double* data, const size_t* dims, const int64_t* labels, size_t rank,
                                  t4a_gpu_tensor** out);
void t4a_gpu_tensor_release(t4a_gpu_tensor* h);
t4a_gpu_status t4a_gpu_tensor_rank(const t4a_gpu_tensor* h, size_t* rank);
t4a_gpu_status t4a_gpu_tensor_dims(const t4a_gpu_tensor* h, size_t* dims, int64_t* labels);
t4a_gpu_status t4a_gpu_tensor_to_host(const t4a_gpu_tensor* h, double* out);
/* permute_indices: new axis k is the axis carrying labels[k] */
t4a_gpu_status t4a_gpu_tensor_permute(const t4a_gpu_tensor* h, const int64_t* labels, t4a_gpu_tensor** out);
/* replace label `from` by `to` (DynIndex replacement / priming on the caller's side) */
t4a_gpu_status t4a_gpu_tensor_relabel(t4a_gpu_tensor* h, int64_t from, int64_t to);
t4a_gpu_status t4a_gpu_tensor_contract(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, t4a_gpu_tensor** out);
/* N-ary contraction of ONE connected tensor network (tensor4all-core/src/defaults/contract.rs:283-298 contract /
 * contract_with_options; plan :885-941, connectivity :1167-1230): labels that occur in more than one operand are summed unless they
 * are listed in retain_labels (ContractionOptions::retain_indices: a retained shared label stays as a batch index and connects its
 * holders); the result carries the labels that occur once or are retained, in order of first appearance.  INVALID_ARGUMENT for no
 * operands, a retained label nobody has, a disconnected network ("Disconnected tensor network: k components found") or a label with
 * two dimensions.  One operand: a copy.  The network is reduced pair by pair (smallest intermediate first), every step a batched GEMM. */
t4a_gpu_status t4a_gpu_tensor_contract_many(const t4a_gpu_tensor* const* tensors, size_t n_tensors, const int64_t* retain_labels,
                                            size_t n_retain, t4a_gpu_tensor** out);
/* outer_product (contract.rs:440-447): the explicit product of operands WITHOUT a common label (INVALID_ARGUMENT otherwise) */
t4a_gpu_status t4a_gpu_tensor_outer_product(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, t4a_gpu_tensor** out);
/* tensordot (contract.rs:420-428, idx_tensor.rs:3596-3638): contraction along explicitly paired axes (labels_a[p] of a with
 * labels_b[p] of b; the labels may differ); result [free axes of a.., free axes of b..].  Errors as the reference's: no pairs, an
 * index that is not there, an axis named twice, unequal dimensions (INVALID_ARGUMENT); a common label that is not paired
 * (NOT_IMPLEMENTED: "Batch contraction is not yet implemented"). */
t4a_gpu_status t4a_gpu_tensor_tensordot(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, const int64_t* labels_a, const int64_t* labels_b,
                                        size_t n_pairs, t4a_gpu_tensor** out);
/* svd_with: u carries [left.., bond_label], s is a rank-1 tensor [bond_label] of singular values, v carries
 * [right.., bond_label_v]; qr_with: q [left.., bond_label], r [bond_label, right..] */
t4a_gpu_status t4a_gpu_tensor_svd(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t truncate,
                                  const t4a_gpu_svd_policy* policy, int32_t has_max_bond_dim, size_t max_bond_dim,
                                  int64_t bond_label, int64_t bond_label_v, t4a_gpu_tensor** u, t4a_gpu_tensor** s,
                                  t4a_gpu_tensor** v);
t4a_gpu_status t4a_gpu_tensor_qr(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t truncate,
                                 int32_t has_rtol, double rtol, int64_t bond_label, t4a_gpu_tensor** q, t4a_gpu_tensor** r);

/* factorize(t, left_inds, FactorizeOptions{alg, canonical, max_bond_dim, svd_policy, qr_rtol}) / factorize_full_rank
 * (defaults/factorize.rs:86-118, :399-425): left carries [left.., bond_label], right [bond_label, right..] and
 * left * right == t.  alg: 0 SVD (canonical 0 Left: left = U, right = S V^H; 1 Right: left = U S, right = V^H),
 * 1 QR (left = Q, right = R), 2 LU (rrLU with rel_tol 1e-14, permuted L and U, left-orthogonal for Left), 3 CI
 * (MatrixLUCI factors).  full_rank != 0: no truncation (rel_tol 0 for LU / CI).  singular_values (capacity min(m, n),
 * may be NULL) is written for SVD only; rank receives the bond dimension. */
t4a_gpu_status t4a_gpu_tensor_factorize(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t alg,
                                        int32_t canonical, int32_t full_rank, const t4a_gpu_svd_policy* policy,
                                        int32_t has_max_bond_dim, size_t max_bond_dim, int32_t has_qr_rtol, double qr_rtol,
                                        int64_t bond_label, t4a_gpu_tensor** left, t4a_gpu_tensor** right, size_t* rank,
                                        double* singular_values);

/* SimpleTensorTrain arithmetic on the device (tensor4all-simplett/src/arithmetic.rs:34-180, tensortrain.rs:264-345,
 * :449-583).  add / sub: direct sum of the cores ([A | B], block diagonal, [A; B]; sub negates the last core of b first);
 * scale = scale_mut (the last core carries the factor); reverse swaps site order and the bond legs; partial_sum(dims) sums
 * the listed sites out (all sites summed: a one-site train of dimension 1 holding the scalar). */
t4a_gpu_status t4a_gpu_tt_add(const t4a_gpu_tt* a, const t4a_gpu_tt* b, t4a_gpu_tt** out);
t4a_gpu_status t4a_gpu_tt_sub(const t4a_gpu_tt* a, const t4a_gpu_tt* b, t4a_gpu_tt** out);
t4a_gpu_status t4a_gpu_tt_scale(t4a_gpu_tt* h, double factor);
/* inner_product (simplett/src/contraction.rs:82-186): sum over all indices of a * b, two MFMA GEMMs per site */
t4a_gpu_status t4a_gpu_tt_inner_product(const t4a_gpu_tt* a, const t4a_gpu_tt* b, double* out);
t4a_gpu_status t4a_gpu_tt_reverse(const t4a_gpu_tt* h, t4a_gpu_tt** out);
t4a_gpu_status t4a_gpu_tt_partial_sum(const t4a_gpu_tt* h, const size_t* dims, size_t n_dims, t4a_gpu_tt** out);

/* Bridge between the tensor-train handles and the labelled tensors (tensor4all-treetn/src/simplett_bridge.rs).
 * _tt_to_tensors = tensor_train_to_treetn_with_names_and_site_indices (:118, :706-794) on a chain: out[s] carries the legs
 * [bond_labels[s-1], site_labels[s], bond_labels[s]], the two boundary legs of dimension 1 dropped (a single site gives
 * [site_labels[0]]); bond_labels has n_sites - 1 entries, out n_sites handles (all released again on failure).
 * _tensors_to_tt = treetn_to_tensor_train (:172-277): tensors[s] must have exactly one leg that is not shared with a chain
 * neighbour (its site index); the bond of tensors[s] and tensors[s+1] is the one label they share; legs are permuted to
 * (left bond, site, right bond) whatever their order. */
t4a_gpu_status t4a_gpu_tt_to_tensors(const t4a_gpu_tt* tt, const int64_t* site_labels, const int64_t* bond_labels,
                                     t4a_gpu_tensor** out);
t4a_gpu_status t4a_gpu_tensors_to_tt(const t4a_gpu_tensor* const* tensors, size_t n_sites, t4a_gpu_tt** out);

/* ---- tensor4all-aci: Alternating Cross Interpolation of an elementwise operator over tensor trains ----
 * (crates/tensor4all-aci/src: elementwise.rs:107-218, state.rs:24-925, local.rs:299-394, global_guard.rs:49-181).
 * Inputs are device-resident trains (t4a_gpu_tt handles); the local candidate matrices are built, pivoted (rrLU) and
 * factorised on the device. */
typedef struct t4a_gpu_aci_options { /* AciOptions (options.rs:37-168) */
    size_t max_iters;                 /* 20 */
    size_t min_iters;                 /* 2 */
    int32_t has_max_bond_dim;         /* 0 <=> None */
    size_t max_bond_dim;
    double tolerance;                 /* 1e-12 */
    int32_t scale_tolerance;          /* 1 */
    uint64_t rng_seed;                /* 0 */
    int32_t enable_global_guard;      /* 1 */
    size_t nsearch_global_pivots;     /* 5 */
    size_t max_nglobal_pivot;         /* 5 */
    size_t nsweeps_global_search;     /* 100 */
    double tol_margin_global_search;  /* 10.0 */
} t4a_gpu_aci_options;
t4a_gpu_status t4a_gpu_aci_options_default(t4a_gpu_aci_options* opts);
/* elementwise_batched's operator (batch.rs:33-217): values[input + n_inputs * point] -> out[point]; non-zero return stops
 * the sweep with T4A_GPU_CALLBACK_ERROR. */
typedef int32_t (*t4a_gpu_aci_op_fn)(void* user, const double* values, size_t n_inputs, size_t n_points, double* out);
/* op_kind: 0 callback `op`, 1 product of the inputs, 2 sum of the inputs (both fused into the candidate-matrix kernel) */
#define T4A_GPU_ACI_OP_CALLBACK 0
#define T4A_GPU_ACI_OP_PRODUCT 1
#define T4A_GPU_ACI_OP_SUM 2
/* elementwise_batched (elementwise.rs:107): initial_guess == NULL -> AciOptions::initial_guess None (random guess: the
 * reference draws ChaCha8 normals, this library a splitmix64 stream — pass the guess for a reproducible match).  ranks /
 * errors / nglobal_pivots need max_iters entries (NULL allowed); termination: 0 Converged, 1 RankLimited, 2 MaxIterations. */
t4a_gpu_status t4a_gpu_aci_elementwise(const t4a_gpu_tt* const* inputs, size_t n_inputs, int32_t op_kind, t4a_gpu_aci_op_fn op,
                                       void* user, const t4a_gpu_aci_options* options, const t4a_gpu_tt* initial_guess,
                                       t4a_gpu_tt** result, size_t* n_iters, size_t* ranks, double* errors,
                                       size_t* nglobal_pivots, int32_t* termination);
/* ElementwiseProblem (state.rs:24-109) stepped bond by bond; the inputs must outlive the problem. */
typedef struct t4a_gpu_aci_problem t4a_gpu_aci_problem;
t4a_gpu_status t4a_gpu_aci_problem_new(const t4a_gpu_tt* const* inputs, size_t n_inputs, int32_t op_kind, t4a_gpu_aci_op_fn op,
                                       void* user, const t4a_gpu_aci_options* options, const t4a_gpu_tt* initial_guess,
                                       t4a_gpu_aci_problem** out);
void t4a_gpu_aci_problem_release(t4a_gpu_aci_problem* h);
/* local_update (state.rs:729-860) */
t4a_gpu_status t4a_gpu_aci_problem_local_update(t4a_gpu_aci_problem* h, size_t bond, int32_t left_orthogonal);
/* add_global_pivots (state.rs:551-652): pivots is n_sites x n_pivots column-major; added = injected pivot count */
t4a_gpu_status t4a_gpu_aci_problem_add_global_pivots(t4a_gpu_aci_problem* h, const size_t* pivots, size_t n_pivots,
                                                     size_t* added);
/* find_global_pivots (global_guard.rs:49): out is n_sites x count column-major, capacity max_nglobal_pivot columns */
t4a_gpu_status t4a_gpu_aci_problem_find_global_pivots(t4a_gpu_aci_problem* h, uint64_t seed, size_t* count, size_t* out);
t4a_gpu_status t4a_gpu_aci_problem_solution(t4a_gpu_aci_problem* h, t4a_gpu_tt** out);
/* left (right == 0) / right frame of `input` at `site` (0..n_sites): shape (0, 0) when absent; out may be NULL */
t4a_gpu_status t4a_gpu_aci_problem_frame(t4a_gpu_aci_problem* h, int32_t right, size_t input, size_t site, size_t* rows,
                                         size_t* cols, double* out);
/* per bond: last pivot error and largest sampled operator magnitude (n_sites - 1 entries each) */
t4a_gpu_status t4a_gpu_aci_problem_errors(const t4a_gpu_aci_problem* h, double* pivot_errors, double* pivot_scales);

/* ---- tensor4all-treeaci: the edge-local step of TreeACI (crates/tensor4all-treeaci/src/local_update.rs:35-262
 * materialize_and_factor_edge; operator contract elementwise.rs:66,197 / batch.rs) from the candidate frames on: for every input k
 * row_frames[k] is bond_dims[k] x row_count and col_frames[k] is bond_dims[k] x col_count, column-major (one contiguous frame vector
 * per candidate, as InputFrameStore::candidate_frames_for_edge returns them).  value_k(row, col) = row_frame . col_frame; the operator
 * (op_kind / op / user as for t4a_gpu_aci_elementwise; callback batch layout values[input + n_inputs * (row + row_count * col)])
 * yields the local matrix; MatrixLUCI with max_bond_dim (0 = None), rel_tol = tolerance when scale_tolerance else abs_tol = tolerance.
 * A zero matrix comes back as the rank-one zero update (rank 1, indices 0, zero factors).  Capacities: row_indices / col_indices
 * min(row_count, col_count), pivot_errors min(row_count, col_count) + 1 (n_pivot_errors = entries written), left row_count x that,
 * right that x col_count (column-major, leading dimensions = row_count resp. *rank); local_values row_count x col_count (may be NULL). */
t4a_gpu_status t4a_gpu_treeaci_local_update_f64(size_t n_inputs, const size_t* bond_dims, const double* const* row_frames,
                                                const double* const* col_frames, size_t row_count, size_t col_count, int32_t op_kind,
                                                t4a_gpu_aci_op_fn op, void* user, size_t max_bond_dim, double tolerance,
                                                int32_t scale_tolerance, int32_t left_orthogonal, size_t* rank, size_t* row_indices,
                                                size_t* col_indices, double* pivot_errors, size_t* n_pivot_errors, double* left,
                                                double* right, double* sampled_scale, double* local_values);

/* ---- measurement hooks (bench.py) ---- */
/* (M, N, rank) of every bond update of the most recent 2-site half-sweep: out is 3 x (n_sites-1). */
t4a_gpu_status t4a_gpu_tci2_last_sweep_shapes(const t4a_gpu_tci2* h, size_t* out);
/* HIP-event profile accumulated since the last reset.  Slots (all in milliseconds / counts):
 *  [0] rrlu kernel ms  [1] rrlu launches  [2] pi-eval kernel ms  [3] pi launches
 *  [4] fill_site_tensors device ms  [5] fill calls  [6] factor (trsm+gemm) ms  [7] factor calls
 *  [8] total pivot steps executed by the rrlu kernel  [9] algorithmic rrlu bytes (BASELINE.md §2 model)
 *  [10] algorithmic flops (rrlu + factors + fill)  [11] function evaluations
 *  [12..15] the rrLU kernel instantiation with the largest total time: ms, launches, algorithmic bytes,
 *           code = RPT*1000 + CPT*10 + 4*row_major_ties + 2*single_workgroup + wave_uniform_columns (negative: LDS / HBM
 *           kernels; >= 100000: single-XCD kernel, see t4a_gpu_tci2_profile_variants) */
#define T4A_GPU_PROFILE_SLOTS 16
t4a_gpu_status t4a_gpu_tci2_profile_enable(t4a_gpu_tci2* h, int32_t enable);
t4a_gpu_status t4a_gpu_tci2_profile_reset(t4a_gpu_tci2* h);
t4a_gpu_status t4a_gpu_tci2_profile_get(const t4a_gpu_tci2* h, double* out /* [T4A_GPU_PROFILE_SLOTS] */);
/* One row {code, ms, launches, algorithmic bytes, pivot steps} per rrLU kernel instantiation used since the last reset
 * (query-then-fill: out == NULL returns the row count).  code >= 100000: single-XCD kernel,
 * 100000 + RPT*100 + CPT*10 + 4*row_major_ties; otherwise as slot 15 of t4a_gpu_tci2_profile_get.  Rows with
 * code >= 10000000 are SUB-aggregates of row code - 10000000: the launches of a bond chain that ran all max_bond_dim pivot
 * steps (the saturated bonds of a sweep); they are already contained in their parent row. */
t4a_gpu_status t4a_gpu_tci2_profile_variants(const t4a_gpu_tci2* h, double* out /* [cap_rows][5] */, size_t cap_rows, size_t* n_rows);

/* Device-side bond chain (the host-free half-sweep of update_pivots, tensorci2.rs:1695-1725 + :1821-2007 for built-in
 * functors): out[0] half-sweeps enqueued as one chain, out[1] bond updates inside such chains, out[2] chains that fell back to
 * the per-bond path part-way (a launch gave up), out[3] half-sweeps that were not eligible (host callback, rook search, shapes
 * beyond the device-dimension kernels) and ran bond by bond, out[4] chained half-sweeps that ran as part of a GROUP chain (one
 * launch per kernel and bond for all handles of a t4a_gpu_tci2_optimize_group call; counted in out[0] as well). */
t4a_gpu_status t4a_gpu_tci2_chain_stats(const t4a_gpu_tci2* h, uint64_t* out /* [5] */);
/* More of the same: out[0] chained sweeps (2-site half-sweeps counted in chain_stats out[0], and 1-site sweeps) that ran as ONE
 * persistent workgroup walking all bonds — every matrix of the sweep at most 64 x 32: small-rank problems and the first iterations
 * of every run (one launch per sweep instead of three per bond); out[1] 1-site sweeps (t4a_gpu_tci2_sweep1site, make_canonical,
 * the final sweep of optimize: tensorci2.rs:865-1050) that ran as a device-side chain — the independent side of every bond is the
 * index table itself, the site tensors come from the factored matrices the chain leaves behind; out[2] 1-site sweeps that were
 * not eligible and ran bond by bond; out[3] chained 1-site sweeps that fell back to the per-bond path part-way. */
t4a_gpu_status t4a_gpu_tci2_chain_stats_ext(const t4a_gpu_tci2* h, uint64_t* out /* [4] */);
/* out[0] asynchronous fill_site_tensors issued by the optimisation loop, out[1] of them replayed from the captured HIP graph, out[2]
 * graph captures (a capture happens the second time a fill with the same signature — device addresses, shapes, staging buffers —
 * is issued; handles whose cores were exported / imported through stream 0 never replay: csrc/tci2.hip issue_fill_ops). */
t4a_gpu_status t4a_gpu_tci2_fill_stats(const t4a_gpu_tci2* h, uint64_t* out /* [3] */);
/* PivotSearchStrategy::Rook on this handle: out[0] searches that ran device-resident (one launch, one host synchronisation per bond:
 * built-in functors), out[1] rows / columns they visited, out[2] searches driven from the host (callback functions: one round trip per
 * visited row / column), out[3] host synchronisations of all searches. */
t4a_gpu_status t4a_gpu_tci2_rook_stats(const t4a_gpu_tci2* h, uint64_t* out /* [4] */);
/* optimize_with_finder (tensorci2.rs:1626-1802) on up to EIGHT handles at once, driven in lock-step by the calling thread: every
 * iteration enqueues the half-sweeps of all handles — as ONE chain of launches when they line up (same number of sites, built-in
 * functors: every kernel serves all handles, handle i's rrLU runs on XCD i), otherwise one chain per handle — then completes them
 * one after the other.  This is the
 * per-GPU form of the patch farm (BASELINE.json configs[4]: eight patches per GPU; adaptive_interpolation.rs:171-330 runs the
 * patches one after the other): results on every handle are exactly those of t4a_gpu_tci2_optimize. */
t4a_gpu_status t4a_gpu_tci2_optimize_group(t4a_gpu_tci2* const* handles, size_t n_handles, const t4a_gpu_tci2_options* options,
                                           int32_t final_sweep1site);
/* fill_site_tensors (tensorci2.rs:1065-1186) on several handles at once: the fills of all handles are issued first — each on its
 * handle's own fill stream, so their kernels share the chip — and completed afterwards, instead of issue + wait handle by handle.
 * Results on every handle are exactly those of t4a_gpu_tci2_fill_site_tensors; a deferred solve error (singular pivot matrix) of
 * any handle is reported after all fills have completed. */
t4a_gpu_status t4a_gpu_tci2_fill_site_tensors_group(t4a_gpu_tci2* const* handles, size_t n_handles);
/* enable == 0: this handle runs every half-sweep bond by bond (A/B measurements, tests).  verify bit 0: after every chain the
 * device-side index tables are read back and compared with the host's I / J sets (T4A_GPU_INTERNAL_ERROR on a difference);
 * bit 1: while profiling, the rrLU launches of a chain are timed with HIP events around each launch instead of the kernels' own
 * time stamps (two more packets per bond on the stream: for calibration runs);
 * bit 2: the small-problem engine (below) is switched off for this handle; bit 3: its launch stamps its phases (diagnostic);
 * bit 4: opt-in to the relaxed guard of the captured fill_site_tensors graph — replay also on a handle whose site tensors are exported /
 * imported asynchronously, unless the legacy default stream or a blocking stream took part (default: never on such a handle);
 * bit 5: the small-problem engine keeps candidate matrices up to 32 x 32 (default 16 x 16: measured, the larger tile is slower than the
 * general path, so growing runs are handed over early). */
t4a_gpu_status t4a_gpu_tci2_set_chain(t4a_gpu_tci2* h, int32_t enable, int32_t verify);
/* Opt-in (default 1): the points of ONE candidate matrix (tensorci2.rs:1859-1893) are split into n_threads contiguous blocks and the host
 * callback is called for the blocks CONCURRENTLY from n_threads host threads (idx / out pointing into the block).  This is outside the
 * reference's contract — it calls `f` / `batched_f` sequentially from one thread on purpose (docs/design/adaptive-tci-interpolation.md:9-11)
 * — so only for callbacks that are thread safe; values, pivots and results are bitwise those of n_threads == 1.  Matrices below 16 384
 * points stay on the calling thread. */
t4a_gpu_status t4a_gpu_tci2_set_callback_threads(t4a_gpu_tci2* h, size_t n_threads);
/* The small-problem engine (round 6): optimize_with_finder (tensorci2.rs:1626-1802) of a small problem — iteration loop, the
 * update_pivots chain (:1821-2007), fill_site_tensors (:1065-1186), convergence_criterion (:1407-1437) and the final 1-site sweep
 * (:1781-1794) — as ONE launch, index sets in the LDS, every candidate matrix (up to 16 x 16; 32 x 32 on request) in the registers of one wavefront.
 * Offered every t4a_gpu_tci2_optimize / _crossinterpolate2 call on a built-in functor without global pivot search; when a set
 * outgrows 16 entries or a matrix its tile the launch hands the state at the start of that iteration back and the general path
 * continues.  Results are those of the general path (index sets, errors, ranks bit for bit).
 * out[0] calls the engine completed, [1] iterations it ran, [2] runs handed back, [3] calls that were not eligible,
 * [4..6] device time of the last launch in 100 MHz ticks (input, iterations, final sweep + results), [7] why the last launch handed back
 * (0 it did not, 1 list / matrix beyond the tile, 2 non-finite value, 3 rank beyond 16, 4 fill not representable / singular),
 * [8..15] shader cycles per phase of the last launch when the phase stamps are on (lists, evaluation, pivot steps, gather, factors,
 * fill, snapshots, convergence + rest). */
t4a_gpu_status t4a_gpu_tci2_small_stats(const t4a_gpu_tci2* h, uint64_t* out /* [16] */);

/* Evaluate a built-in function on the device for a batch of full multi-indices (parity check of the
 * workload definition itself).  idx: n_sites x n_pts column-major. */
t4a_gpu_status t4a_gpu_fn_eval(int32_t fid, int32_t n_acc, const double* params, const uint64_t* weights,
                               const size_t* local_dims, size_t n_sites, const size_t* idx, size_t n_pts,
                               double* out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* T4A_GPU_H */
