// capi.hip — extern "C" surface declared in include/t4a_gpu.h.
// No exception crosses the boundary (tensor4all-capi/src/lib.rs:139-162 convention): every entry point
// runs inside `guarded`, which stores the message in a thread-local slot and returns a status code.
#include "stdrng.hpp"
#include "smallrng.hpp"
#include <memory>
#include <initializer_list>
#include <mutex>

#include "patching.hpp"
#include "tci2.hpp"
#include "tree.hpp"
#include "quantics.hpp"
#include "tensorops.hpp"
#include "aci.hpp"
#include "globalsearch.hpp"

struct t4a_gpu_tci2 {
    t4a::Tci2 impl;
    explicit t4a_gpu_tci2(const std::vector<size_t>& d) : impl(d) {}
};

struct t4a_gpu_treetci {
    t4a::TreeTci impl;
    t4a_gpu_treetci(const std::vector<size_t>& d, const t4a::TreeGraph& g) : impl(d, g) {}
};

struct t4a_gpu_tensor {
    t4a::DevBuf<double> buf;
    std::vector<size_t> dims;
    std::vector<int64_t> labels;
    size_t size() const
    {
        size_t n = 1;
        for (size_t d : dims) n *= d;
        return n;
    }
    t4a::TensorView view() const
    {
        t4a::TensorView v;
        v.d_data = buf.get();
        v.dims = dims;
        v.labels = labels;
        return v;
    }
};

struct t4a_gpu_qtci {
    std::unique_ptr<t4a::QuanticsTci> impl;
};

struct t4a_gpu_ptt {
    std::unique_ptr<t4a::PartitionedTT> impl;
};

struct t4a_gpu_tt {
    t4a::TensorTrain impl;
    t4a_gpu_tt(const std::vector<std::array<size_t, 3>>& d, const double* data) : impl(d, data) {}
    t4a_gpu_tt(const std::vector<t4a::DevCore>& cores, hipStream_t src) : impl(cores, src) {}
};

namespace t4a {
const std::string& last_error_ref();

namespace {

template <class F> t4a_gpu_status guarded(F&& body)
{
    try {
        body();
        return T4A_GPU_SUCCESS;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return T4A_GPU_INTERNAL_ERROR;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return T4A_GPU_INTERNAL_ERROR;
    } catch (...) {
        set_last_error("unknown internal error");
        return T4A_GPU_INTERNAL_ERROR;
    }
}

#define T4A_REQUIRE_PTR(p)                                                       \
    do {                                                                         \
        if ((p) == nullptr) throw ::t4a::Error(T4A_GPU_NULL_POINTER, #p " is null"); \
    } while (0)

std::mutex g_dense_mutex;
Engine* g_dense_engine = nullptr;

// One process-global engine for the handle-less dense entry points, serialised by a mutex exactly like the
// reference's global default backend (tensorbackend/src/context.rs:318-338).  It is never destroyed: a destructor that runs
// during static destruction would call into a HIP runtime that may already be gone (under rocprofv3 that teardown does not
// return); the process exit reclaims stream and buffers.
Engine& dense_engine()
{
    if (!g_dense_engine) g_dense_engine = new Engine();
    return *g_dense_engine;
}

TCI2Options convert_options(const t4a_gpu_tci2_options* o)
{
    if (!o) throw Error(T4A_GPU_NULL_POINTER, "options is null");
    TCI2Options r;
    r.tolerance = o->tolerance;
    r.max_iter = o->max_iter;
    r.max_bond_dim = o->max_bond_dim;
    r.pivot_search = o->pivot_search;
    r.normalize_error = o->normalize_error != 0;
    r.verbosity = o->verbosity;
    r.max_nglobal_pivot = o->max_nglobal_pivot;
    r.nsearch = o->nsearch;
    r.sweep_strategy = o->sweep_strategy;
    r.ncheck_history = o->ncheck_history;
    r.strictly_nested = o->strictly_nested != 0;
    r.tol_margin_global_search = o->tol_margin_global_search;
    r.has_seed = o->has_seed != 0;
    r.seed = o->seed;
    if (r.sweep_strategy < 0 || r.sweep_strategy > 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "invalid sweep_strategy");
    if (r.pivot_search < 0 || r.pivot_search > 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "invalid pivot_search");
    return r;
}

void upload(Engine& e, double* dst, const double* src, size_t count)
{
    if (count == 0) return;
    T4A_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyHostToDevice, e.stream()));
    T4A_HIP(hipStreamSynchronize(e.stream()));
}
void download(Engine& e, double* dst, const double* src, size_t count)
{
    if (count == 0) return;
    T4A_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToHost, e.stream()));
    T4A_HIP(hipStreamSynchronize(e.stream()));
}
// The same copies in stream order WITHOUT a host synchronisation of their own, for entry points that end with a synchronising download():
// the caller's buffers stay valid until the entry point returns, and everything between is ordered on the engine's stream (round 6: an SVD
// call paid five stream synchronisations for its copies, a QR call three, a product three).
// ... and the guard that makes "until the entry point returns" true on EVERY path: an exception thrown between the first queued copy and
// the final download (a non-finite input, a HIP error) must not let the entry point return while a copy from or into the caller's memory
// is still in flight.  On the normal path the stream is already idle: one more (empty) synchronisation.  Measured (profiles/r06_svd_small.txt
// section 8): SVD calls of tiny matrices 17 - 20 us shorter (3 x 2: 125 -> 108 us), 64 x 64 and the QR calls unchanged within noise.
struct StreamSyncOnExit {
    Engine& e;
    explicit StreamSyncOnExit(Engine& eng) : e(eng) {}
    ~StreamSyncOnExit() { (void)hipStreamSynchronize(e.stream()); }
    StreamSyncOnExit(const StreamSyncOnExit&) = delete;
    StreamSyncOnExit& operator=(const StreamSyncOnExit&) = delete;
};
void upload_async(Engine& e, double* dst, const double* src, size_t count)
{
    if (count == 0) return;
    T4A_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyHostToDevice, e.stream()));
}
void download_async(Engine& e, double* dst, const double* src, size_t count)
{
    if (count == 0) return;
    T4A_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToHost, e.stream()));
}

// the kernels index with 32-bit integers: larger dimensions are refused before any narrowing cast
void require_int_dims(std::initializer_list<size_t> dims, const char* what)
{
    for (size_t d : dims)
        if (d > (size_t)std::numeric_limits<int>::max())
            throw Error(T4A_GPU_INVALID_ARGUMENT, std::string(what) + ": dimension exceeds the 32-bit index range of the device kernels");
}

size_t checked_mul(size_t a, size_t b, const char* what)
{
    if (a != 0 && b > std::numeric_limits<size_t>::max() / a)
        throw Error(T4A_GPU_INVALID_ARGUMENT, std::string(what) + " overflows usize");
    return a * b;
}

} // namespace
} // namespace t4a

using namespace t4a;

extern "C" {

t4a_gpu_status t4a_gpu_last_error_message(char* buf, size_t buf_len, size_t* required_len)
{
    const std::string& msg = last_error_ref();
    const size_t need = msg.size() + 1;
    if (required_len) *required_len = need;
    if (!buf) return required_len ? T4A_GPU_SUCCESS : T4A_GPU_NULL_POINTER;
    if (buf_len < need) return T4A_GPU_BUFFER_TOO_SMALL;
    std::memcpy(buf, msg.c_str(), need);
    return T4A_GPU_SUCCESS;
}

t4a_gpu_status t4a_gpu_stdrng_sample(uint64_t seed, const size_t* dims, size_t n, size_t* out)
{
    return guarded([&] {
        if (n == 0) return;
        T4A_REQUIRE_PTR(dims);
        T4A_REQUIRE_PTR(out);
        for (size_t i = 0; i < n; ++i)
            if (dims[i] == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "random_range(0..0): empty range");
        StdRng rng(seed);
        for (size_t i = 0; i < n; ++i) out[i] = rng.random_range(dims[i]);
    });
}

t4a_gpu_status t4a_gpu_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t* out16)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(key8);
        T4A_REQUIRE_PTR(out16);
        if (rounds <= 0 || (rounds & 1)) throw Error(T4A_GPU_INVALID_ARGUMENT, "ChaCha rounds must be a positive even number");
        StdRng::block(key8, counter, (uint32_t)stream, (uint32_t)(stream >> 32), rounds, out16);
    });
}

// ---- the other two random streams of the reference (smallrng.hpp): known-answer entry points, host only ----
t4a_gpu_status t4a_gpu_siphash(const uint8_t* msg, size_t len, uint64_t k0, uint64_t k1, int32_t c_rounds, int32_t d_rounds, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        if (len > 0) T4A_REQUIRE_PTR(msg);
        if (c_rounds == 1 && d_rounds == 3) {
            SipHasher<1, 3> h(k0, k1);
            h.write(msg, len);
            *out = h.finish();
        } else if (c_rounds == 2 && d_rounds == 4) {
            SipHasher<2, 4> h(k0, k1);
            h.write(msg, len);
            *out = h.finish();
        } else {
            throw Error(T4A_GPU_INVALID_ARGUMENT, "SipHash-1-3 or SipHash-2-4");
        }
    });
}

t4a_gpu_status t4a_gpu_smallrng_words(uint64_t seed, const uint64_t* state4, size_t n, uint64_t* out)
{
    return guarded([&] {
        if (n == 0) return;
        T4A_REQUIRE_PTR(out);
        SmallRng rng = state4 ? SmallRng::from_state(state4) : SmallRng(seed);
        for (size_t i = 0; i < n; ++i) out[i] = rng.next_u64();
    });
}

t4a_gpu_status t4a_gpu_smallrng_sample(uint64_t seed, const size_t* dims, size_t n, size_t* out)
{
    return guarded([&] {
        if (n == 0) return;
        T4A_REQUIRE_PTR(dims);
        T4A_REQUIRE_PTR(out);
        for (size_t i = 0; i < n; ++i)
            if (dims[i] == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "random_range(0..0): empty range");
        SmallRng rng(seed);
        for (size_t i = 0; i < n; ++i) out[i] = rng.random_range(dims[i]);
    });
}

t4a_gpu_status t4a_gpu_smallrng_shuffle(uint64_t seed, size_t n, size_t* out)
{
    return guarded([&] {
        if (n == 0) return;
        T4A_REQUIRE_PTR(out);
        std::vector<size_t> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = i;
        SmallRng rng(seed);
        rng.shuffle(v);
        for (size_t i = 0; i < n; ++i) out[i] = v[i];
    });
}

t4a_gpu_status t4a_gpu_tree_edge_seed(uint64_t seed, const char* tag, size_t u, size_t v, size_t history_len, size_t n_pivots_i, size_t n_pivots_j, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(tag);
        T4A_REQUIRE_PTR(out);
        DefaultHasher h;
        h.write_u64(seed);
        h.write_str(tag, std::strlen(tag));
        h.write_usize(std::min(u, v));
        h.write_usize(std::max(u, v));
        h.write_usize(history_len);
        h.write_usize(n_pivots_i);
        h.write_usize(n_pivots_j);
        *out = h.finish();
    });
}

t4a_gpu_status t4a_gpu_chacha8_standard_normal(uint64_t seed, size_t n, double* out, size_t n_words, uint32_t* out_words)
{
    return guarded([&] {
        if (n > 0) {
            T4A_REQUIRE_PTR(out);
            ChaCha8Rng rng(seed);
            for (size_t i = 0; i < n; ++i) out[i] = StandardNormal::sample(rng);
        }
        if (n_words > 0) {
            T4A_REQUIRE_PTR(out_words);
            ChaCha8Rng rng(seed);
            for (size_t i = 0; i < n_words; ++i) out_words[i] = rng.next_u32();
        }
    });
}

t4a_gpu_status t4a_gpu_device_count(int32_t* out_count)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out_count);
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *out_count = c;
    });
}

t4a_gpu_status t4a_gpu_set_device(int32_t device)
{
    return guarded([&] {
        require_device();
        T4A_HIP(hipSetDevice(device));
    });
}

const char* t4a_gpu_version(void) { return "t4a-mi355x 0.1.0 (gfx950)"; }
int32_t t4a_gpu_diag_switches_enabled(void)
{
    if (!t4a::kDiagSwitches) {
        // a known experiment switch in the environment of a production build measures the default path: say so once
        static const char* const known[] = {"T4A_NO_FUSED_PI", "T4A_NO_SMALL_FILL", "T4A_EXPORT_SYNC", "T4A_FILL_GRAPH_SHARED", "T4A_SVD_NO_PRECOND",
                                            "T4A_FILL_DEFER", "T4A_OLD_PRESIZE", "T4A_FILL_GRAPH_NO_COPY"};
        static bool warned = false;
        if (!warned)
            for (const char* k : known)
                if (std::getenv(k)) {
                    std::fprintf(stderr, "[t4a] %s is set, but this library was built without -DT4A_DIAG_SWITCHES: the switch has NO effect\n", k);
                    warned = true;
                }
    }
    return t4a::kDiagSwitches ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ dense
t4a_gpu_status t4a_gpu_rrlu_f64(double* a_inout, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                double abs_tol, int32_t left_orthogonal, size_t* row_perm, size_t* col_perm,
                                size_t* npivots, double* last_error)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(row_perm);
        T4A_REQUIRE_PTR(col_perm);
        T4A_REQUIRE_PTR(npivots);
        T4A_REQUIRE_PTR(last_error);
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count) T4A_REQUIRE_PTR(a_inout);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_a = e.pi(std::max<size_t>(count, 1));
        upload(e, d_a, a_inout, count);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        LuciResult r = e.luci(d_a, (int)m, (int)n, o, false, true);
        if (count) download(e, a_inout, e.lu_buf(), count);
        for (size_t i = 0; i < m; ++i) row_perm[i] = (size_t)r.row_perm[i];
        for (size_t j = 0; j < n; ++j) col_perm[j] = (size_t)r.col_perm[j];
        *npivots = (size_t)r.rank;
        *last_error = r.last_error;
    });
}

t4a_gpu_status t4a_gpu_luci_f64(const double* a, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                double abs_tol, int32_t left_orthogonal, size_t* rank, size_t* rows, size_t* cols,
                                double* pivot_errors, double* left, double* right)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(rank);
        T4A_REQUIRE_PTR(rows);
        T4A_REQUIRE_PTR(cols);
        T4A_REQUIRE_PTR(pivot_errors);
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count) {
            T4A_REQUIRE_PTR(a);
            T4A_REQUIRE_PTR(left);
            T4A_REQUIRE_PTR(right);
        }
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_a = e.pi(std::max<size_t>(count, 1));
        upload(e, d_a, a, count);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        LuciResult r = e.luci(d_a, (int)m, (int)n, o, true, false);
        *rank = (size_t)r.rank;
        for (int i = 0; i < r.rank; ++i) {
            rows[i] = (size_t)r.row_perm[i];
            cols[i] = (size_t)r.col_perm[i];
        }
        for (size_t i = 0; i < r.pivot_errors.size(); ++i) pivot_errors[i] = r.pivot_errors[i];
        if (r.rank > 0) {
            download(e, left, e.left(), m * (size_t)r.rank);
            download(e, right, e.right(), n * (size_t)r.rank);
        }
    });
}

t4a_gpu_status t4a_gpu_gemm_batched_f64(size_t batch, size_t m, size_t k, size_t n, const double* a, const double* b,
                                        double* c)
{
    return guarded([&] {
        const size_t na = checked_mul(checked_mul(m, k, "a shape"), batch, "a shape");
        const size_t nb = checked_mul(checked_mul(k, n, "b shape"), batch, "b shape");
        const size_t nc = checked_mul(checked_mul(m, n, "c shape"), batch, "c shape");
        require_int_dims({m, n, k, batch}, "gemm shape");
        if (na) T4A_REQUIRE_PTR(a);
        if (nb) T4A_REQUIRE_PTR(b);
        if (nc) T4A_REQUIRE_PTR(c);
        if (nc == 0) return;
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        e.d_tmp.reserve(std::max<size_t>(na + nb, 1));
        e.d_tmp2.reserve(nc);
        double* da = e.d_tmp.get();
        double* db = da + na;
        StreamSyncOnExit sync_on_exit(e);
        upload_async(e, da, a, na);
        upload_async(e, db, b, nb);
        if (k == 0) {
            fill_launch(e.d_tmp2.get(), nc, 0.0, e.stream());
        } else {
            GemmDesc g;
            g.m = (int)m;
            g.n = (int)n;
            g.k = (int)k;
            g.A = da;
            g.lda = (int)m;
            g.strideA = (long long)(m * k);
            g.transA = 0;
            g.B = db;
            g.ldb = (int)k;
            g.strideB = (long long)(k * n);
            g.transB = 0;
            g.C = e.d_tmp2.get();
            g.ldc = (int)m;
            g.strideC = (long long)(m * n);
            g.alpha = 1.0;
            g.beta = 0.0;
            g.batch = (int)batch;
            gemm_launch(g, e.stream());
        }
        T4A_HIP(hipGetLastError());
        download(e, c, e.d_tmp2.get(), nc);
    });
}

t4a_gpu_status t4a_gpu_gemm_f64(const double* a, const double* b, size_t m, size_t k, size_t n, double* c)
{
    return t4a_gpu_gemm_batched_f64(1, m, k, n, a, b, c);
}

t4a_gpu_status t4a_gpu_trsm_f64(const double* a, size_t na, const double* b, size_t bm, size_t bn, int32_t left_side,
                                int32_t lower, int32_t transpose_a, int32_t unit_diagonal, double* x)
{
    return guarded([&] {
        const size_t acount = checked_mul(na, na, "a shape");
        const size_t bcount = checked_mul(bm, bn, "b shape");
        require_int_dims({na, bm, bn}, "trsm shape");
        if (left_side ? (bm != na) : (bn != na))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "triangular_solve: dimension mismatch between A and B");
        if (acount) T4A_REQUIRE_PTR(a);
        if (bcount) {
            T4A_REQUIRE_PTR(b);
            T4A_REQUIRE_PTR(x);
        }
        if (bcount == 0) return;
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        e.d_tmp.reserve(2 * acount + 1);
        e.d_tmp2.reserve(2 * bcount + 1);
        double* dA = e.d_tmp.get();
        double* dAt = dA + acount;
        double* dB = e.d_tmp2.get();
        double* dBt = dB + bcount;
        upload(e, dA, a, acount);
        upload(e, dB, b, bcount);
        // reduce to a left-side solve with an untransposed triangular matrix T:  T Y = R
        //   left : op(A) X = B           -> T = op(A),   R = B
        //   right: X op(A) = B           -> T = op(A)^T, R = B^T, X = Y^T
        const bool need_t = left_side ? (transpose_a != 0) : (transpose_a == 0);
        const double* T = dA;
        bool low = lower != 0;
        if (need_t) {
            transpose_launch(dA, (int)na, (int)na, (int)na, dAt, (int)na, st);
            T = dAt;
            low = !low;
        }
        double* R = dB;
        int rn = (int)bm, rrhs = (int)bn;
        if (!left_side) {
            transpose_launch(dB, (int)bm, (int)bn, (int)bm, dBt, (int)bn, st);
            R = dBt;
            rn = (int)bn;
            rrhs = (int)bm;
        }
        TrsmProblem tp;
        tp.T = T;
        tp.ldt = (int)na;
        tp.n = (int)na;
        tp.B = R;
        tp.ldb = rn;
        tp.nrhs = rrhs;
        tp.lower = low ? 1 : 0;
        tp.unit_diag = unit_diagonal ? 1 : 0;
        tp.skip_flag = nullptr;
        DevBuf<TrsmProblem> dprob;
        dprob.reserve(1);
        T4A_HIP(hipMemcpyAsync(dprob.get(), &tp, sizeof(tp), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
        trsm_left_batched_launch(dprob.get(), 1, (int)na, rrhs, st);
        if (!left_side) {
            transpose_launch(dBt, (int)bn, (int)bm, (int)bn, dB, (int)bm, st);
            R = dB;
        }
        T4A_HIP(hipGetLastError());
        download(e, x, R, bcount);
    });
}

t4a_gpu_status t4a_gpu_solve_f64(const double* a, size_t n, const double* b, size_t nrhs, double* x)
{
    return guarded([&] {
        const size_t acount = checked_mul(n, n, "a shape");
        const size_t bcount = checked_mul(n, nrhs, "b shape");
        require_int_dims({n, nrhs}, "solve shape");
        if (acount) T4A_REQUIRE_PTR(a);
        if (bcount) {
            T4A_REQUIRE_PTR(b);
            T4A_REQUIRE_PTR(x);
        }
        if (bcount == 0) return;
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        e.d_tmp.reserve(acount + 1);
        e.d_tmp2.reserve(bcount + 1);
        upload(e, e.d_tmp.get(), a, acount);
        upload(e, e.d_tmp2.get(), b, bcount);
        DevBuf<int> dpiv;
        dpiv.reserve(n + 1);
        DevBuf<LuProblem> dlp;
        dlp.reserve(1);
        DevBuf<TrsmProblem> dtp;
        dtp.reserve(2);
        LuProblem lp;
        lp.A = e.d_tmp.get();
        lp.lda = (int)n;
        lp.n = (int)n;
        lp.piv = dpiv.get();
        lp.info = dpiv.get() + n;
        lp.B = e.d_tmp2.get();
        lp.ldb = (int)n;
        lp.nrhs = (int)nrhs;
        lp.pmax_bits = nullptr;
        TrsmProblem t[2];
        t[0].T = lp.A;
        t[0].ldt = lp.lda;
        t[0].n = lp.n;
        t[0].B = lp.B;
        t[0].ldb = lp.ldb;
        t[0].nrhs = lp.nrhs;
        t[0].lower = 1;
        t[0].unit_diag = 1;
        t[0].skip_flag = nullptr;
        t[1] = t[0];
        t[1].lower = 0;
        t[1].unit_diag = 0;
        T4A_HIP(hipMemcpyAsync(dlp.get(), &lp, sizeof(lp), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(dtp.get(), t, sizeof(t), hipMemcpyHostToDevice, st));
        T4A_HIP(hipStreamSynchronize(st));
        // (round 5) blocked LU + one fused launch for both triangular solves; outside its size range the two-step path
        const bool fused = lu_solve_blocked_launch(dlp.get(), 1, (int)n, (int)nrhs, st);
        const bool forward_done = fused || lu_forward_blocked_launch(dlp.get(), 1, (int)n, (int)nrhs, st);
        if (!forward_done) lu_batched_launch(dlp.get(), 1, (int)n, st);
        int info = 0;
        T4A_HIP(hipMemcpyAsync(&info, lp.info, sizeof(int), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
        if (info != 0) throw Error(T4A_GPU_SINGULAR_MATRIX, "solve: matrix is singular");
        if (!fused) {
            if (!forward_done) trsm_left_batched_launch(dtp.get(), 1, (int)n, (int)nrhs, st);
            trsm_left_batched_launch(dtp.get() + 1, 1, (int)n, (int)nrhs, st);
        }
        T4A_HIP(hipGetLastError());
        download(e, x, e.d_tmp2.get(), bcount);
    });
}

t4a_gpu_status t4a_gpu_fn_eval(int32_t fid, int32_t n_acc, const double* params, const uint64_t* weights,
                               const size_t* local_dims, size_t n_sites, const size_t* idx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(params);
        T4A_REQUIRE_PTR(weights);
        T4A_REQUIRE_PTR(local_dims);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        if (fid < 0 || fid >= T4A_FN_COUNT || n_acc < 1 || n_acc > T4A_FN_MAX_ACC)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown built-in function");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        // fold every full index into its accumulators as a "row"; a single all-zero "column"
        std::vector<size_t> off(n_sites);
        size_t total = 0;
        for (size_t s = 0; s < n_sites; ++s) {
            off[s] = total;
            total += local_dims[s];
        }
        std::vector<uint64_t> acc(n_pts * (size_t)n_acc, 0), zero((size_t)n_acc, 0);
        for (size_t p = 0; p < n_pts; ++p)
            for (int k = 0; k < n_acc; ++k) {
                uint64_t a = 0;
                for (size_t s = 0; s < n_sites; ++s) {
                    const size_t v = idx[p * n_sites + s];
                    if (v >= local_dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "index out of bounds");
                    a += weights[(size_t)k * total + off[s] + v];
                }
                acc[p * n_acc + k] = a;
            }
        DevBuf<uint64_t> dacc;
        dacc.reserve(acc.size() + zero.size());
        T4A_HIP(hipMemcpyAsync(dacc.get(), acc.data(), acc.size() * 8, hipMemcpyHostToDevice, e.stream()));
        T4A_HIP(hipMemcpyAsync(dacc.get() + acc.size(), zero.data(), zero.size() * 8, hipMemcpyHostToDevice, e.stream()));
        T4A_HIP(hipStreamSynchronize(e.stream()));
        e.d_tmp.reserve(n_pts);
        FnDevice fn;
        fn.fid = fid;
        fn.n_acc = n_acc;
        std::memcpy(fn.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
        pi_eval_launch(fn, dacc.get(), (int)n_pts, dacc.get() + acc.size(), 1, e.d_tmp.get(), (int)n_pts, false, nullptr,
                       e.stream());
        T4A_HIP(hipGetLastError());
        download(e, out, e.d_tmp.get(), n_pts);
    });
}

// ------------------------------------------------------------------------------------------------ TCI2
t4a_gpu_status t4a_gpu_tci2_options_default(t4a_gpu_tci2_options* o)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(o);
        std::memset(o, 0, sizeof(*o));
        o->tolerance = 1e-8;
        o->max_iter = 20;
        o->max_bond_dim = 0;
        o->pivot_search = 0;
        o->normalize_error = 1;
        o->verbosity = 0;
        o->max_nglobal_pivot = 5;
        o->nsearch = 5;
        o->sweep_strategy = 2;
        o->strictly_nested = 0;
        o->ncheck_history = 3;
        o->tol_margin_global_search = 10.0;
        o->has_seed = 0;
        o->seed = 0;
    });
}

t4a_gpu_status t4a_gpu_tci2_new(const size_t* local_dims, size_t n_sites, t4a_gpu_tci2** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_sites) T4A_REQUIRE_PTR(local_dims);
        std::vector<size_t> d(local_dims, local_dims + n_sites);
        if (d.size() < 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
        for (size_t s = 0; s < d.size(); ++s) {
            if (d[s] == 0)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension at site " + std::to_string(s) + " must be positive");
            if (d[s] > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension too large");
        }
        *out = new t4a_gpu_tci2(d);
    });
}

void t4a_gpu_tci2_release(t4a_gpu_tci2* h) { delete h; }

t4a_gpu_status t4a_gpu_tci2_set_builtin_function(t4a_gpu_tci2* h, int32_t fid, int32_t n_acc, const double* params,
                                                 const uint64_t* weights)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(params);
        T4A_REQUIRE_PTR(weights);
        h->impl.set_builtin(fid, n_acc, params, weights);
    });
}

t4a_gpu_status t4a_gpu_tci2_set_callback(t4a_gpu_tci2* h, t4a_gpu_batch_eval_fn cb, void* ctx)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.set_callback(cb, ctx);
    });
}

static std::vector<std::vector<uint32_t>> pivots_from(const t4a_gpu_tci2* h, const size_t* pivots, size_t n_pivots)
{
    const size_t ns = h->impl.len();
    std::vector<std::vector<uint32_t>> p(n_pivots, std::vector<uint32_t>(ns));
    for (size_t k = 0; k < n_pivots; ++k)
        for (size_t s = 0; s < ns; ++s) {
            const size_t v = pivots[s + ns * k];
            if (v > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "pivot value out of bounds");
            p[k][s] = (uint32_t)v;
        }
    return p;
}

t4a_gpu_status t4a_gpu_tci2_add_global_pivots(t4a_gpu_tci2* h, const size_t* pivots, size_t n_pivots)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pivots) T4A_REQUIRE_PTR(pivots);
        h->impl.add_global_pivots(pivots_from(h, pivots, n_pivots));
    });
}

t4a_gpu_status t4a_gpu_tci2_crossinterpolate2(t4a_gpu_tci2* h, const size_t* initial_pivots, size_t n_pivots,
                                              const t4a_gpu_tci2_options* options)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        TCI2Options o = convert_options(options);
        o.validate(); // options are validated before any callback runs (tensorci2/tests/mod.rs:7-144)
        h->impl.crossinterpolate2(pivots_from(h, initial_pivots, n_pivots), o);
    });
}

t4a_gpu_status t4a_gpu_tci2_optimize(t4a_gpu_tci2* h, const t4a_gpu_tci2_options* options, int32_t final_sweep1site)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.optimize(convert_options(options), final_sweep1site != 0);
    });
}

t4a_gpu_status t4a_gpu_tci2_optimize_group(t4a_gpu_tci2* const* handles, size_t n_handles, const t4a_gpu_tci2_options* options,
                                           int32_t final_sweep1site)
{
    return guarded([&] {
        if (n_handles) T4A_REQUIRE_PTR(handles);
        std::vector<Tci2*> hs;
        for (size_t i = 0; i < n_handles; ++i) {
            T4A_REQUIRE_PTR(handles[i]);
            hs.push_back(&handles[i]->impl);
        }
        Tci2::optimize_group(hs, convert_options(options), final_sweep1site != 0);
    });
}

t4a_gpu_status t4a_gpu_tci2_fill_site_tensors_group(t4a_gpu_tci2* const* handles, size_t n_handles)
{
    return guarded([&] {
        if (n_handles) T4A_REQUIRE_PTR(handles);
        std::vector<Tci2*> hs;
        for (size_t i = 0; i < n_handles; ++i) {
            T4A_REQUIRE_PTR(handles[i]);
            hs.push_back(&handles[i]->impl);
        }
        Tci2::fill_site_tensors_group(hs);
    });
}

t4a_gpu_status t4a_gpu_tci2_sweep2site(t4a_gpu_tci2* h, int32_t forward, const t4a_gpu_tci2_options* options)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.sweep2site(forward != 0, convert_options(options));
    });
}

t4a_gpu_status t4a_gpu_tci2_sweep1site(t4a_gpu_tci2* h, int32_t forward, double rel_tol, double abs_tol,
                                       size_t max_bond_dim, int32_t update_tensors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.sweep1site(forward != 0, rel_tol, abs_tol,
                           max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim, update_tensors != 0);
    });
}

t4a_gpu_status t4a_gpu_tci2_fill_site_tensors(t4a_gpu_tci2* h)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.fill_site_tensors();
    });
}

t4a_gpu_status t4a_gpu_tci2_make_canonical(t4a_gpu_tci2* h, double rel_tol, double abs_tol, size_t max_bond_dim)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.make_canonical(rel_tol, abs_tol, max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim);
    });
}

t4a_gpu_status t4a_gpu_tci2_len(const t4a_gpu_tci2* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.len();
    });
}
t4a_gpu_status t4a_gpu_tci2_rank(const t4a_gpu_tci2* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.rank();
    });
}
t4a_gpu_status t4a_gpu_tci2_link_dims(const t4a_gpu_tci2* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        const auto v = h->impl.link_dims();
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    });
}
t4a_gpu_status t4a_gpu_tci2_max_sample_value(const t4a_gpu_tci2* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.max_sample_value;
    });
}
t4a_gpu_status t4a_gpu_tci2_max_bond_error(const t4a_gpu_tci2* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.max_bond_error();
    });
}
t4a_gpu_status t4a_gpu_tci2_bond_errors(const t4a_gpu_tci2* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (size_t i = 0; i < h->impl.bond_errors.size(); ++i) out[i] = h->impl.bond_errors[i];
    });
}
t4a_gpu_status t4a_gpu_tci2_pivot_errors(const t4a_gpu_tci2* h, size_t* count, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        *count = h->impl.pivot_errors.size();
        if (out)
            for (size_t i = 0; i < h->impl.pivot_errors.size(); ++i) out[i] = h->impl.pivot_errors[i];
    });
}

t4a_gpu_status t4a_gpu_tci2_index_set(const t4a_gpu_tci2* h, int32_t which, size_t site, size_t* count, size_t* width,
                                      size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        T4A_REQUIRE_PTR(width);
        if (site >= h->impl.len() || which < 0 || which > 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "site/which out of range");
        if (out) const_cast<t4a_gpu_tci2*>(h)->impl.sync_digits(); // (after a device-side bond chain the digit tables are decoded on demand)
        const IndexSet& s = which == 0 ? h->impl.i_set[site] : h->impl.j_set[site];
        *count = s.count;
        *width = s.width;
        if (out)
            for (size_t k = 0; k < s.count * s.width; ++k) out[k] = s.d[k];
    });
}

t4a_gpu_status t4a_gpu_tci2_set_index_set(t4a_gpu_tci2* h, int32_t which, size_t site, size_t count, const size_t* data)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (site >= h->impl.len() || which < 0 || which > 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "site/which out of range");
        h->impl.sync_digits();
        IndexSet& s = which == 0 ? h->impl.i_set[site] : h->impl.j_set[site];
        const size_t first_site = which == 0 ? 0 : site + 1;
        if (count * s.width) T4A_REQUIRE_PTR(data);
        IndexSet n;
        n.width = s.width;
        n.count = count;
        n.d.resize(count * s.width);
        for (size_t k = 0; k < count; ++k)
            for (size_t q = 0; q < s.width; ++q) {
                const size_t v = data[k * s.width + q];
                if (v >= h->impl.local_dims[first_site + q]) throw Error(T4A_GPU_INVALID_ARGUMENT, "index out of bounds");
                n.d[k * s.width + q] = (uint32_t)v;
            }
        s = n;
        h->impl.invalidate_site_tensors();
        h->impl.mark_sets_changed();
    });
}

t4a_gpu_status t4a_gpu_tci2_set_max_sample_value(t4a_gpu_tci2* h, double value)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.max_sample_value = value;
    });
}

t4a_gpu_status t4a_gpu_tci2_clear_history(t4a_gpu_tci2* h)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.clear_history();
    });
}

t4a_gpu_status t4a_gpu_tci2_site_tensor(const t4a_gpu_tci2* h, size_t site, size_t* dims3, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(dims3);
        if (site >= h->impl.len()) throw Error(T4A_GPU_INVALID_ARGUMENT, "site out of range");
        t4a_gpu_tci2* hh = const_cast<t4a_gpu_tci2*>(h);
        hh->impl.fill_wait();
        if (!out) {
            dims3[0] = h->impl.cores[site].l;
            dims3[1] = h->impl.cores[site].s;
            dims3[2] = h->impl.cores[site].r;
            return;
        }
        std::vector<double> v = hh->impl.site_tensor_host(site, dims3);
        std::memcpy(out, v.data(), v.size() * sizeof(double));
    });
}

t4a_gpu_status t4a_gpu_tci2_site_tensor_device(const t4a_gpu_tci2* h, size_t site, void* out_device)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out_device);
        if (site >= h->impl.len()) throw Error(T4A_GPU_INVALID_ARGUMENT, "site out of range");
        t4a_gpu_tci2* hh = const_cast<t4a_gpu_tci2*>(h);
        hh->impl.fill_wait();
        const DevCore& c = h->impl.cores[site];
        if (c.size()) {
            T4A_HIP(hipMemcpyAsync(out_device, c.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice,
                                   hh->impl.eng.stream()));
            T4A_HIP(hipStreamSynchronize(hh->impl.eng.stream()));
        }
    });
}

t4a_gpu_status t4a_gpu_tci2_set_site_tensor_device(t4a_gpu_tci2* h, size_t site, const size_t* dims3,
                                                   const void* in_device)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(dims3);
        if (site >= h->impl.len()) throw Error(T4A_GPU_INVALID_ARGUMENT, "site out of range");
        if (dims3[1] != h->impl.local_dims[site]) throw Error(T4A_GPU_INVALID_ARGUMENT, "site dimension mismatch");
        h->impl.fill_wait();
        DevCore& c = h->impl.cores[site];
        const size_t count = dims3[0] * dims3[1] * dims3[2];
        c.buf.reserve(std::max<size_t>(count, 1));
        c.l = dims3[0];
        c.s = dims3[1];
        c.r = dims3[2];
        if (count) {
            T4A_REQUIRE_PTR(in_device);
            T4A_HIP(hipMemcpyAsync(c.buf.get(), in_device, count * sizeof(double), hipMemcpyDeviceToDevice,
                                   h->impl.eng.stream()));
            T4A_HIP(hipStreamSynchronize(h->impl.eng.stream()));
        }
    });
}

t4a_gpu_status t4a_gpu_tci2_export_site_tensors_async(t4a_gpu_tci2* h, void* dst_device, size_t stride,
                                                      void* consumer_stream)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(dst_device);
        h->impl.export_site_tensors_async(static_cast<double*>(dst_device), stride,
                                          static_cast<hipStream_t>(consumer_stream));
    });
}

t4a_gpu_status t4a_gpu_tci2_n_iterations(const t4a_gpu_tci2* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.errors_hist.size();
    });
}
t4a_gpu_status t4a_gpu_tci2_history(const t4a_gpu_tci2* h, size_t* ranks, double* errors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        for (size_t i = 0; i < h->impl.errors_hist.size(); ++i) {
            if (ranks) ranks[i] = h->impl.ranks_hist[i];
            if (errors) errors[i] = h->impl.errors_hist[i];
        }
    });
}
t4a_gpu_status t4a_gpu_tci2_termination(const t4a_gpu_tci2* h, int32_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.termination;
    });
}

t4a_gpu_status t4a_gpu_tci2_evaluate(t4a_gpu_tci2* h, const size_t* idx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        const size_t ns = h->impl.len();
        std::vector<uint32_t> u(n_pts * ns);
        for (size_t k = 0; k < u.size(); ++k) {
            if (idx[k] > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "evaluate: index out of bounds");
            u[k] = (uint32_t)idx[k];
        }
        std::vector<double> v = h->impl.evaluate(u.data(), n_pts);
        std::memcpy(out, v.data(), n_pts * sizeof(double));
    });
}

t4a_gpu_status t4a_gpu_tci2_sum(t4a_gpu_tci2* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.sum();
    });
}

t4a_gpu_status t4a_gpu_tci2_set_site_shard(t4a_gpu_tci2* h, size_t rank, size_t world)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (world == 0 || rank >= world) throw Error(T4A_GPU_INVALID_ARGUMENT, "invalid shard (rank, world)");
        h->impl.shard_rank = rank;
        h->impl.shard_world = world;
    });
}

t4a_gpu_status t4a_gpu_tci2_set_pi_shard(t4a_gpu_tci2* h, size_t rank, size_t world, t4a_gpu_allgather_fn gather, void* ctx)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (world == 0 || rank >= world) throw Error(T4A_GPU_INVALID_ARGUMENT, "invalid shard (rank, world)");
        if (world > 1 && !gather) throw Error(T4A_GPU_NULL_POINTER, "a column-block shard over more than one rank needs an all-gather callback");
        h->impl.pi_shard = PiShard();
        h->impl.pi_shard.rank = rank;
        h->impl.pi_shard.world = world;
        h->impl.pi_shard.gather = world > 1 ? gather : nullptr;
        h->impl.pi_shard.gather_ctx = ctx;
    });
}

t4a_gpu_status t4a_gpu_tci2_pi_shard_stats(t4a_gpu_tci2* h, size_t* out2)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out2);
        out2[0] = h->impl.pi_shard.n_gathers;
        out2[1] = h->impl.pi_shard.bytes_sent;
    });
}

t4a_gpu_status t4a_gpu_pi_shard_eval(size_t rank, size_t world, t4a_gpu_batch_eval_fn cb, void* cb_ctx, t4a_gpu_allgather_fn gather,
                                     void* gather_ctx, size_t n_sites, const uint32_t* a, size_t wa, size_t a0, size_t na,
                                     const uint32_t* b, size_t wb, size_t b0, size_t nb, double* out)
{
    return guarded([&] { // host only: no device is touched
        if (!cb) throw Error(T4A_GPU_NULL_POINTER, "batch callback is NULL");
        if (world == 0 || rank >= world) throw Error(T4A_GPU_INVALID_ARGUMENT, "invalid shard (rank, world)");
        if (na == 0 || nb == 0) return;
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        if (wa + wb != n_sites || a0 + wa > n_sites || b0 + wb > n_sites) throw Error(T4A_GPU_INVALID_ARGUMENT, "index halves do not cover all sites");
        if (world > 1) {
            if (!gather) throw Error(T4A_GPU_NULL_POINTER, "a column-block shard over more than one rank needs an all-gather callback");
            PiShard ps;
            ps.rank = rank;
            ps.world = world;
            ps.gather = gather;
            ps.gather_ctx = gather_ctx;
            pi_shard_evaluate(ps, cb, cb_ctx, n_sites, a, wa, a0, na, b, wb, b0, nb, out);
            return;
        }
        std::vector<uint32_t> idx(na * nb * n_sites);
        for (size_t ia = 0; ia < na; ++ia)
            for (size_t ib = 0; ib < nb; ++ib) {
                uint32_t* dst = idx.data() + (ia * nb + ib) * n_sites;
                std::memcpy(dst + a0, a + ia * wa, wa * sizeof(uint32_t));
                std::memcpy(dst + b0, b + ib * wb, wb * sizeof(uint32_t));
            }
        const int64_t got = cb(cb_ctx, idx.data(), n_sites, na * nb, out);
        if (got < 0 || (size_t)got != na * nb) throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned a wrong number of values");
    });
}

t4a_gpu_status t4a_gpu_tci2_export_site_shard_async(t4a_gpu_tci2* h, void* dst_device, size_t stride, void* consumer_stream)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(dst_device);
        h->impl.export_site_shard_async(static_cast<double*>(dst_device), stride, static_cast<hipStream_t>(consumer_stream));
    });
}

t4a_gpu_status t4a_gpu_tci2_import_site_shard_async(t4a_gpu_tci2* h, const void* src_device, size_t stride, size_t per_rank,
                                                    void* producer_stream)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(src_device);
        h->impl.import_site_shard_async(static_cast<const double*>(src_device), stride, per_rank, static_cast<hipStream_t>(producer_stream));
    });
}

t4a_gpu_status t4a_gpu_tci2_set_keep_site_tensors(t4a_gpu_tci2* h, int32_t keep)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.keep_site_tensors = keep != 0;
    });
}

t4a_gpu_status t4a_gpu_tci2_last_sweep_shapes(const t4a_gpu_tci2* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (size_t b = 0; b < h->impl.last_sweep_shapes.size(); ++b)
            for (int k = 0; k < 3; ++k) out[3 * b + k] = h->impl.last_sweep_shapes[b][k];
    });
}

t4a_gpu_status t4a_gpu_tci2_profile_enable(t4a_gpu_tci2* h, int32_t enable)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.eng.prof.enabled = enable != 0;
    });
}
t4a_gpu_status t4a_gpu_tci2_profile_reset(t4a_gpu_tci2* h)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        for (double& v : h->impl.eng.prof.v) v = 0.0;
        h->impl.eng.variant_stats_.clear();
    });
}
t4a_gpu_status t4a_gpu_tci2_profile_get(const t4a_gpu_tci2* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (int i = 0; i < T4A_GPU_PROFILE_SLOTS; ++i) out[i] = h->impl.eng.prof.v[i];
        // slots 12..15: the rrLU kernel instantiation with the largest total time
        double best = -1.0;
        for (const auto& kv : h->impl.eng.variant_stats_)
            if (kv.second[0] > best) {
                best = kv.second[0];
                out[12] = kv.second[0];
                out[13] = kv.second[1];
                out[14] = kv.second[2];
                out[15] = (double)kv.first;
            }
    });
}
t4a_gpu_status t4a_gpu_tci2_profile_variants(const t4a_gpu_tci2* h, double* out, size_t cap_rows, size_t* n_rows)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(n_rows);
        const auto& vs = h->impl.eng.variant_stats_;
        *n_rows = vs.size();
        if (out == nullptr) return; // query
        if (cap_rows < vs.size()) throw Error(T4A_GPU_BUFFER_TOO_SMALL, "profile_variants: buffer too small");
        size_t i = 0;
        for (const auto& kv : vs) {
            out[5 * i + 0] = (double)kv.first;
            for (int j = 0; j < 4; ++j) out[5 * i + 1 + j] = kv.second[j];
            ++i;
        }
    });
}


t4a_gpu_status t4a_gpu_tci2_chain_stats(const t4a_gpu_tci2* h, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (int k = 0; k < 5; ++k) out[k] = h->impl.chain_stats[k];
    });
}

t4a_gpu_status t4a_gpu_tci2_chain_stats_ext(const t4a_gpu_tci2* h, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (int k = 0; k < 4; ++k) out[k] = h->impl.chain_stats_ext[k];
    });
}

t4a_gpu_status t4a_gpu_tci2_rook_stats(const t4a_gpu_tci2* h, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        const RookWork& w = h->impl.rook_work();
        out[0] = w.n_device_searches;
        out[1] = w.n_device_visits;
        out[2] = w.n_host_searches;
        out[3] = w.n_host_syncs;
    });
}

t4a_gpu_status t4a_gpu_tci2_fill_stats(const t4a_gpu_tci2* h, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (int k = 0; k < 3; ++k) out[k] = h->impl.fill_stats()[k];
    });
}

t4a_gpu_status t4a_gpu_tci2_set_chain(t4a_gpu_tci2* h, int32_t enable, int32_t verify)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.chain_enabled = enable != 0;
        h->impl.chain_verify = (verify & 1) != 0;
        h->impl.chain_event_timing = (verify & 2) != 0;
        h->impl.small_enabled = (verify & 4) == 0;
        h->impl.small_stamps = (verify & 8) != 0;
        h->impl.fill_graph_relaxed = (verify & 16) != 0;
        h->impl.small_tile_max = (verify & 32) ? 32 : 16;
    });
}

t4a_gpu_status t4a_gpu_tci2_set_callback_threads(t4a_gpu_tci2* h, size_t n_threads)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_threads < 1 || n_threads > 256) throw Error(T4A_GPU_INVALID_ARGUMENT, "callback threads: between 1 and 256");
        h->impl.callback_threads = n_threads;
    });
}

t4a_gpu_status t4a_gpu_tci2_small_stats(const t4a_gpu_tci2* h, uint64_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        for (int k = 0; k < 4; ++k) out[k] = h->impl.small_stats[k];
        for (int k = 0; k < 3; ++k) out[4 + k] = h->impl.small_last_clocks_[k];
        out[7] = (uint64_t)h->impl.small_last_reason_;
        for (int k = 0; k < 8; ++k) out[8 + k] = h->impl.small_last_clocks_[3 + k];
    });
}

// ------------------------------------------------------------------------------------------------ lazy block-rook LUCI
extern "C++" {
static void rook_outputs(Engine& e, const LuciResult& r, size_t m, size_t n, size_t* rank, size_t* rows, size_t* cols,
                         double* pivot_errors, double* left, double* right)
{
    *rank = (size_t)r.rank;
    for (int i = 0; i < r.rank; ++i) {
        rows[i] = (size_t)r.row_perm[i];
        cols[i] = (size_t)r.col_perm[i];
    }
    for (size_t i = 0; i < r.pivot_errors.size(); ++i) pivot_errors[i] = r.pivot_errors[i];
    if (r.rank > 0) {
        download(e, left, e.left(), m * (size_t)r.rank);
        download(e, right, e.right(), n * (size_t)r.rank);
    }
}
static RookWork& dense_rook_work()
{
    static RookWork w; // guarded by g_dense_mutex like the dense engine itself
    return w;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_luci_blocks_f64(size_t m, size_t n, t4a_gpu_fill_block_fn fill_block, void* ctx,
                                       size_t max_bond_dim, double rel_tol, double abs_tol, int32_t left_orthogonal,
                                       size_t* rank, size_t* rows, size_t* cols, double* pivot_errors, double* left,
                                       double* right)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(rank);
        T4A_REQUIRE_PTR(rows);
        T4A_REQUIRE_PTR(cols);
        T4A_REQUIRE_PTR(pivot_errors);
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count) {
            T4A_REQUIRE_PTR(fill_block);
            T4A_REQUIRE_PTR(left);
            T4A_REQUIRE_PTR(right);
        }
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "luci: dimensions above 65535 are not supported");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        std::vector<size_t> all_rows(m), all_cols(n);
        for (size_t i = 0; i < m; ++i) all_rows[i] = i;
        for (size_t j = 0; j < n; ++j) all_cols[j] = j;
        std::vector<double> hbuf(std::max(m, n));
        RookSource src;
        src.M = (int)m;
        src.N = (int)n;
        src.column = [&](int c, double* d_out) {
            const size_t cc = (size_t)c;
            fill_block(ctx, all_rows.data(), m, &cc, 1, hbuf.data());
            T4A_HIP(hipMemcpyAsync(d_out, hbuf.data(), m * sizeof(double), hipMemcpyHostToDevice, st));
            T4A_HIP(hipStreamSynchronize(st));
        };
        src.row = [&](int r, double* d_out) {
            const size_t rr = (size_t)r;
            fill_block(ctx, &rr, 1, all_cols.data(), n, hbuf.data());
            T4A_HIP(hipMemcpyAsync(d_out, hbuf.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
            T4A_HIP(hipStreamSynchronize(st));
        };
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        LuciResult r = rook_luci(e, dense_rook_work(), src, o, nullptr, nullptr);
        rook_outputs(e, r, m, n, rank, rows, cols, pivot_errors, left, right);
    });
}

t4a_gpu_status t4a_gpu_luci_rook_f64(const double* a, size_t m, size_t n, size_t max_bond_dim, double rel_tol,
                                     double abs_tol, int32_t left_orthogonal, size_t* rank, size_t* rows, size_t* cols,
                                     double* pivot_errors, double* left, double* right)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(rank);
        T4A_REQUIRE_PTR(rows);
        T4A_REQUIRE_PTR(cols);
        T4A_REQUIRE_PTR(pivot_errors);
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count) {
            T4A_REQUIRE_PTR(a);
            T4A_REQUIRE_PTR(left);
            T4A_REQUIRE_PTR(right);
        }
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "luci: dimensions above 65535 are not supported");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        // the dense source lives on the device: A (m x n) and its transpose (rows contiguous)
        e.d_tmp2.reserve(2 * std::max<size_t>(count, 1));
        double* d_a = e.d_tmp2.get();
        double* d_at = d_a + count;
        upload(e, d_a, a, count);
        if (count) transpose_launch(d_a, (int)m, (int)n, (int)m, d_at, (int)n, st);
        RookSource src;
        src.M = (int)m;
        src.N = (int)n;
        src.column = [&](int c, double* d_out) {
            T4A_HIP(hipMemcpyAsync(d_out, d_a + (size_t)c * m, m * sizeof(double), hipMemcpyDeviceToDevice, st));
        };
        src.row = [&](int r, double* d_out) {
            T4A_HIP(hipMemcpyAsync(d_out, d_at + (size_t)r * n, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        };
        static const bool host_driven = std::getenv("T4A_ROOK_HOST") != nullptr; // (A/B and parity of the two search drivers)
        if (!host_driven)
            src.full = [&](double* d_out) { T4A_HIP(hipMemcpyAsync(d_out, d_a, count * sizeof(double), hipMemcpyDeviceToDevice, st)); };
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = left_orthogonal != 0;
        LuciResult r = rook_luci(e, dense_rook_work(), src, o, nullptr, nullptr);
        rook_outputs(e, r, m, n, rank, rows, cols, pivot_errors, left, right);
    });
}

// ------------------------------------------------------------------------------------------------ svd / qr / full-piv LU
t4a_gpu_status t4a_gpu_svd_f64(const double* a, size_t m, size_t n, double* u, double* s, double* vt)
{
    return guarded([&] {
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD of an empty matrix");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "svd: dimensions above 65535 are not supported");
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(u);
        T4A_REQUIRE_PTR(s);
        T4A_REQUIRE_PTR(vt);
        const size_t k = std::min(m, n);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_a = e.pi(count);
        e.d_tmp.reserve(m * k + k + k * n); // (before the copy is queued: a growing buffer is freed and allocated, which synchronises)
        StreamSyncOnExit sync_on_exit(e);
        upload_async(e, d_a, a, count);
        double* d_u = e.d_tmp.get();
        double* d_s = d_u + m * k;
        double* d_vt = d_s + k;
        e.svd(d_a, (int)m, (int)n, d_u, d_s, d_vt);
        download_async(e, u, d_u, m * k);
        download_async(e, s, d_s, k);
        download(e, vt, d_vt, k * n);
    });
}

t4a_gpu_status t4a_gpu_rsvd_f64(const double* a, size_t m, size_t n, size_t k, size_t oversample, size_t power_iters, uint64_t seed,
                                double* u, double* s, double* vt)
{
    return guarded([&] {
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD of an empty matrix");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "svd: dimensions above 65535 are not supported");
        if (k == 0 || k > std::min(m, n)) throw Error(T4A_GPU_INVALID_ARGUMENT, "rsvd: the rank must be between 1 and min(m, n)");
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(u);
        T4A_REQUIRE_PTR(s);
        T4A_REQUIRE_PTR(vt);
        const size_t l = std::min(std::min(m, n), k + oversample); // sketch width
        // Omega (n x l): standard normals by Box-Muller on the library's StdRng stream
        std::vector<double> omega(n * l);
        {
            StdRng rng(seed);
            for (size_t i = 0; i < omega.size(); i += 2) {
                const double u1 = ((double)(rng.next_u64() >> 11) + 1.0) * (1.0 / 9007199254740992.0);
                const double u2 = (double)(rng.next_u64() >> 11) * (1.0 / 9007199254740992.0);
                const double rad = std::sqrt(-2.0 * std::log(u1));
                omega[i] = rad * std::cos(6.283185307179586 * u2);
                if (i + 1 < omega.size()) omega[i + 1] = rad * std::sin(6.283185307179586 * u2);
            }
        }
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        DevBuf<double> dA, dO, dY, dQ, dR, dZ, dB, dUb, dSb, dVt, dU;
        dA.reserve(count);
        dO.reserve(n * l);
        dY.reserve(m * l);
        dQ.reserve(m * l);
        dR.reserve(l * std::max(l, n));
        dZ.reserve(n * l);
        dB.reserve(l * n);
        dUb.reserve(l * l);
        dSb.reserve(l);
        dVt.reserve(l * n);
        dU.reserve(m * l);
        upload(e, dA.get(), a, count);
        upload(e, dO.get(), omega.data(), omega.size());
        auto gemm = [&](const double* A_, int lda, bool ta, const double* B_, int ldb, double* C_, int ldc, size_t M_, size_t N_, size_t K_) {
            GemmDesc g{};
            g.m = (int)M_;
            g.n = (int)N_;
            g.k = (int)K_;
            g.A = A_;
            g.lda = lda;
            g.transA = ta ? 1 : 0;
            g.B = B_;
            g.ldb = ldb;
            g.transB = 0;
            g.C = C_;
            g.ldc = ldc;
            g.alpha = 1.0;
            g.beta = 0.0;
            g.batch = 1;
            gemm_launch(g, st);
        };
        gemm(dA.get(), (int)m, false, dO.get(), (int)n, dY.get(), (int)m, m, l, n);      // Y = A Omega
        e.qr(dY.get(), (int)m, (int)l, dQ.get(), dR.get());                             // Q (m x l)
        for (size_t it = 0; it < power_iters; ++it) {
            gemm(dA.get(), (int)m, true, dQ.get(), (int)m, dZ.get(), (int)n, n, l, m);  // Z = A^T Q (n x l)
            e.qr(dZ.get(), (int)n, (int)l, dO.get(), dR.get());                         // orthonormal basis of Z in dO (n x l)
            gemm(dA.get(), (int)m, false, dO.get(), (int)n, dY.get(), (int)m, m, l, n); // Y = A Z
            e.qr(dY.get(), (int)m, (int)l, dQ.get(), dR.get());
        }
        gemm(dQ.get(), (int)m, true, dA.get(), (int)m, dB.get(), (int)l, l, n, m);      // B = Q^T A (l x n)
        e.svd(dB.get(), (int)l, (int)n, dUb.get(), dSb.get(), dVt.get());               // B = Ub S Vt, Ub l x l, Vt l x n
        gemm(dQ.get(), (int)m, false, dUb.get(), (int)l, dU.get(), (int)m, m, l, l);    // U = Q Ub
        T4A_HIP(hipGetLastError());
        download(e, u, dU.get(), m * k);  // the first k columns
        download(e, s, dSb.get(), k);
        // the first k rows of Vt (l x n, column-major: strided) -> k x n
        std::vector<double> hv(l * n);
        download(e, hv.data(), dVt.get(), l * n);
        for (size_t j = 0; j < n; ++j)
            for (size_t i = 0; i < k; ++i) vt[i + k * j] = hv[i + l * j];
    });
}

t4a_gpu_status t4a_gpu_qr_f64(const double* a, size_t m, size_t n, double* q, double* r)
{
    return guarded([&] {
        const size_t count = checked_mul(m, n, "matrix shape");
        require_int_dims({m, n}, "matrix shape");
        if (count == 0) return;
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "qr: dimensions above 65535 are not supported");
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(q);
        T4A_REQUIRE_PTR(r);
        const size_t k = std::min(m, n);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_a = e.pi(count);
        e.d_tmp.reserve(m * k + k * n);
        StreamSyncOnExit sync_on_exit(e);
        upload_async(e, d_a, a, count);
        double* d_q = e.d_tmp.get();
        double* d_r = d_q + m * k;
        e.qr(d_a, (int)m, (int)n, d_q, d_r);
        download_async(e, q, d_q, m * k);
        download(e, r, d_r, k * n);
    });
}

t4a_gpu_status t4a_gpu_full_piv_lu_f64(const double* a, size_t n, double* p, double* l, double* u, double* q)
{
    return guarded([&] {
        const size_t count = checked_mul(n, n, "matrix shape");
        require_int_dims({n}, "matrix shape");
        if (count == 0) return;
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(p);
        T4A_REQUIRE_PTR(l);
        T4A_REQUIRE_PTR(u);
        T4A_REQUIRE_PTR(q);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_a = e.pi(count);
        upload(e, d_a, a, count);
        RrLUOptions o;
        o.rel_tol = 0.0;
        o.abs_tol = 0.0;
        o.left_orthogonal = true;
        LuciResult r = e.luci(d_a, (int)n, (int)n, o, false, true);
        // square factors: L = [lower trapezoid of the first rank columns | identity columns], U = [first rank rows ; 0]
        e.d_tmp.reserve(2 * count);
        double* d_l = e.d_tmp.get();
        double* d_u = d_l + count;
        hipStream_t st = e.stream();
        set_identity_launch(d_l, (int)n, (int)n, (int)n, st);
        fill_launch(d_u, count, 0.0, st);
        if (r.rank > 0) {
            tri_extract_launch(e.lu_buf(), (int)n, (int)n, r.rank, 1, 1, d_l, (int)n, st);
            tri_extract_launch(e.lu_buf(), (int)n, r.rank, (int)n, 0, 0, d_u, (int)n, st);
        }
        T4A_HIP(hipGetLastError());
        download(e, l, d_l, count);
        download(e, u, d_u, count);
        std::memset(p, 0, count * sizeof(double));
        std::memset(q, 0, count * sizeof(double));
        for (size_t k = 0; k < n; ++k) {
            p[k + n * (size_t)r.row_perm[k]] = 1.0;
            q[k + n * (size_t)r.col_perm[k]] = 1.0;
        }
    });
}

// ------------------------------------------------------------------------------------------------ SimpleTensorTrain
extern "C++" {
static std::vector<uint32_t> narrow_indices(const size_t* idx, size_t count)
{
    std::vector<uint32_t> u(count);
    for (size_t k = 0; k < count; ++k) {
        if (idx[k] > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "index out of bounds");
        u[k] = (uint32_t)idx[k];
    }
    return u;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_tt_new(const size_t* dims3, size_t n_sites, const double* cores, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_sites) T4A_REQUIRE_PTR(dims3);
        require_device();
        std::vector<std::array<size_t, 3>> d(n_sites);
        for (size_t s = 0; s < n_sites; ++s) {
            d[s] = {dims3[3 * s], dims3[3 * s + 1], dims3[3 * s + 2]};
            checked_mul(checked_mul(d[s][0], d[s][1], "tensor shape"), d[s][2], "tensor shape");
        }
        *out = new t4a_gpu_tt(d, cores);
    });
}

void t4a_gpu_tt_release(t4a_gpu_tt* h) { delete h; }

t4a_gpu_status t4a_gpu_tt_clone(const t4a_gpu_tt* h, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = new t4a_gpu_tt(h->impl.cores, h->impl.eng.stream());
    });
}

t4a_gpu_status t4a_gpu_tt_len(const t4a_gpu_tt* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.len();
    });
}

t4a_gpu_status t4a_gpu_tt_dims(const t4a_gpu_tt* h, size_t* dims3)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (h->impl.len()) T4A_REQUIRE_PTR(dims3);
        for (size_t s = 0; s < h->impl.len(); ++s) {
            dims3[3 * s] = h->impl.cores[s].l;
            dims3[3 * s + 1] = h->impl.cores[s].s;
            dims3[3 * s + 2] = h->impl.cores[s].r;
        }
    });
}

t4a_gpu_status t4a_gpu_tt_site_tensor(const t4a_gpu_tt* h, size_t site, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        std::vector<double> v = const_cast<t4a_gpu_tt*>(h)->impl.site_tensor_host(site);
        if (!v.empty()) {
            T4A_REQUIRE_PTR(out);
            std::memcpy(out, v.data(), v.size() * sizeof(double));
        }
    });
}

t4a_gpu_status t4a_gpu_tt_evaluate(t4a_gpu_tt* h, const size_t* idx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        std::vector<uint32_t> u = narrow_indices(idx, checked_mul(n_pts, h->impl.len(), "index buffer"));
        std::vector<double> v = h->impl.evaluate(u.data(), n_pts);
        std::memcpy(out, v.data(), n_pts * sizeof(double));
    });
}

t4a_gpu_status t4a_gpu_tt_sum(t4a_gpu_tt* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.sum();
    });
}

t4a_gpu_status t4a_gpu_tt_norm2(t4a_gpu_tt* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.norm2();
    });
}

t4a_gpu_status t4a_gpu_tt_compress(t4a_gpu_tt* h, int32_t method, double tolerance, size_t max_bond_dim,
                                   int32_t normalize_error)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (method < 0 || method > 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown compression method");
        CompressionOptions o;
        o.method = (CompressionMethod)method;
        o.tolerance = tolerance;
        o.max_bond_dim = max_bond_dim;
        o.normalize_error = normalize_error != 0;
        h->impl.compress(o);
    });
}

t4a_gpu_status t4a_gpu_tt_evaluate_many(t4a_gpu_tt* h, const size_t* idx, size_t n_pts, size_t split, double* out,
                                        size_t* used_split)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (used_split) *used_split = split;
        if (n_pts == 0) return; // cache.rs:563-565
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        std::vector<uint32_t> u = narrow_indices(idx, checked_mul(n_pts, h->impl.len(), "index buffer"));
        const size_t s = h->impl.evaluate_many(u.data(), n_pts, split, out);
        if (used_split) *used_split = s;
    });
}

// ---- estimate_true_error / floating_zone / opt_first_pivot (tensorci/src/globalsearch.rs, optfirstpivot.rs) ----
extern "C++" {
static t4a::SearchFn wrap_search_fn(t4a_gpu_batch_eval_fn f, void* ctx)
{
    return [f, ctx](const uint32_t* idx, size_t n_sites, size_t n_pts, double* out) {
        const int64_t got = f(ctx, idx, n_sites, n_pts, out);
        if (got != (int64_t)n_pts)
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback returned " + std::to_string(got) + " values for " + std::to_string(n_pts) + " points");
    };
}
static std::vector<size_t> dims_vec(const size_t* local_dims, size_t n_sites)
{
    return std::vector<size_t>(local_dims, local_dims + n_sites);
}
} // extern "C++"

t4a_gpu_status t4a_gpu_tt_floating_zone(t4a_gpu_tt* tt, t4a_gpu_batch_eval_fn f, void* ctx, const size_t* local_dims, size_t n_sites,
                                        const size_t* init_p, uint64_t seed, double early_stop_tol, size_t* pivot_out, double* error_out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(tt);
        T4A_REQUIRE_PTR(f);
        T4A_REQUIRE_PTR(pivot_out);
        T4A_REQUIRE_PTR(error_out);
        if (n_sites) T4A_REQUIRE_PTR(local_dims);
        std::vector<uint32_t> init;
        if (init_p) init = narrow_indices(init_p, n_sites);
        auto r = t4a::floating_zone(tt->impl, wrap_search_fn(f, ctx), dims_vec(local_dims, n_sites), init_p ? &init : nullptr, seed, early_stop_tol);
        for (size_t s = 0; s < r.first.size(); ++s) pivot_out[s] = r.first[s];
        *error_out = r.second;
    });
}

t4a_gpu_status t4a_gpu_tt_estimate_true_error(t4a_gpu_tt* tt, t4a_gpu_batch_eval_fn f, void* ctx, size_t nsearch, const size_t* initial_points,
                                              size_t n_initial, uint64_t seed, size_t* pivots_out, double* errors_out, size_t capacity,
                                              size_t* n_out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(tt);
        T4A_REQUIRE_PTR(f);
        T4A_REQUIRE_PTR(n_out);
        const size_t n = tt->impl.len();
        std::vector<std::vector<uint32_t>> init;
        if (initial_points) {
            const std::vector<uint32_t> u = narrow_indices(initial_points, checked_mul(n, n_initial, "initial points"));
            for (size_t k = 0; k < n_initial; ++k) init.emplace_back(u.begin() + k * n, u.begin() + (k + 1) * n);
        }
        const auto res = t4a::estimate_true_error(tt->impl, wrap_search_fn(f, ctx), nsearch, initial_points ? &init : nullptr, seed);
        *n_out = res.size();
        if (res.size() > capacity) throw Error(T4A_GPU_BUFFER_TOO_SMALL, "estimate_true_error: output capacity " + std::to_string(capacity) +
                                                                              " < " + std::to_string(res.size()) + " results");
        if (!res.empty()) {
            T4A_REQUIRE_PTR(pivots_out);
            T4A_REQUIRE_PTR(errors_out);
        }
        for (size_t k = 0; k < res.size(); ++k) {
            for (size_t s = 0; s < n; ++s) pivots_out[s + n * k] = res[k].first[s];
            errors_out[k] = res[k].second;
        }
    });
}

t4a_gpu_status t4a_gpu_opt_first_pivot(t4a_gpu_batch_eval_fn f, void* ctx, const size_t* local_dims, size_t n_sites, const size_t* first_pivot,
                                       size_t max_sweep, size_t* pivot_out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(f);
        T4A_REQUIRE_PTR(pivot_out);
        if (n_sites) {
            T4A_REQUIRE_PTR(local_dims);
            T4A_REQUIRE_PTR(first_pivot);
        }
        const std::vector<uint32_t> fp = narrow_indices(first_pivot, n_sites);
        const std::vector<uint32_t> r = t4a::opt_first_pivot(wrap_search_fn(f, ctx), dims_vec(local_dims, n_sites), fp, max_sweep);
        for (size_t s = 0; s < r.size(); ++s) pivot_out[s] = r[s];
    });
}

t4a_gpu_status t4a_gpu_tci2_to_tensor_train(t4a_gpu_tci2* h, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        h->impl.fill_wait();
        *out = new t4a_gpu_tt(h->impl.cores, h->impl.eng.stream());
    });
}

t4a_gpu_status t4a_gpu_tci2_from_tensor_train(const t4a_gpu_tt* tt, double tolerance, size_t max_bond_dim,
                                              size_t max_iter, t4a_gpu_tci2** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(tt);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (tt->impl.len() < 2)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion requires at least 2 tensor-train sites");
        FromTensorTrainOptions o;
        o.tolerance = tolerance;
        o.max_bond_dim = max_bond_dim;
        o.max_iter = max_iter;
        std::unique_ptr<t4a_gpu_tci2> h(new t4a_gpu_tci2(tt->impl.site_dims()));
        h->impl.assign_from_tensor_train(tt->impl, o);
        *out = h.release();
    });
}


// ------------------------------------------------------------------------------------------------ adaptive patching
extern "C++" {
static t4a_gpu_ptt* run_adaptive(const size_t* local_dims, size_t n_sites, const FullFunction& f,
                                 const size_t* initial_pivots, size_t n_pivots, const t4a_gpu_tci2_options* tci_options,
                                 const size_t* patch_order, size_t n_initial_pivots, int32_t recycle_pivots)
{
    if (n_sites) T4A_REQUIRE_PTR(local_dims);
    if (n_pivots) T4A_REQUIRE_PTR(initial_pivots);
    AdaptiveOptions ao;
    ao.tci = convert_options(tci_options);
    ao.n_initial_pivots = n_initial_pivots;
    ao.recycle_pivots = recycle_pivots != 0;
    if (patch_order) ao.patch_order.assign(patch_order, patch_order + n_sites);
    std::vector<size_t> dims(local_dims, local_dims + n_sites);
    std::vector<std::vector<uint32_t>> piv(n_pivots, std::vector<uint32_t>(n_sites));
    for (size_t k = 0; k < n_pivots; ++k)
        for (size_t s = 0; s < n_sites; ++s) {
            const size_t v = initial_pivots[s + n_sites * k];
            // out-of-range values are reported by the driver's own validation; keep them representable
            piv[k][s] = v > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)v;
        }
    require_device();
    std::unique_ptr<t4a_gpu_ptt> h(new t4a_gpu_ptt);
    h->impl = adaptive_interpolate(dims, f, piv, ao);
    return h.release();
}
} // extern "C++"

t4a_gpu_status t4a_gpu_adaptive_interpolate_builtin(const size_t* local_dims, size_t n_sites, int32_t fid, int32_t n_acc,
                                                    const double* params, const uint64_t* weights,
                                                    const size_t* initial_pivots, size_t n_pivots,
                                                    const t4a_gpu_tci2_options* tci_options, const size_t* patch_order,
                                                    size_t n_initial_pivots, int32_t recycle_pivots, t4a_gpu_ptt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        T4A_REQUIRE_PTR(params);
        T4A_REQUIRE_PTR(weights);
        if (fid < 0 || fid >= T4A_FN_COUNT) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown built-in function id");
        if (n_acc < 1 || n_acc > T4A_FN_MAX_ACC) throw Error(T4A_GPU_INVALID_ARGUMENT, "n_acc out of range");
        FullFunction f;
        f.builtin = true;
        f.fid = fid;
        f.n_acc = n_acc;
        std::memcpy(f.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
        size_t total = 0;
        for (size_t s = 0; s < n_sites; ++s) total += local_dims ? local_dims[s] : 0;
        f.weights.assign(weights, weights + (size_t)n_acc * total);
        *out = run_adaptive(local_dims, n_sites, f, initial_pivots, n_pivots, tci_options, patch_order, n_initial_pivots,
                            recycle_pivots);
    });
}

t4a_gpu_status t4a_gpu_adaptive_interpolate_callback(const size_t* local_dims, size_t n_sites, t4a_gpu_batch_eval_fn cb,
                                                     void* ctx, const size_t* initial_pivots, size_t n_pivots,
                                                     const t4a_gpu_tci2_options* tci_options, const size_t* patch_order,
                                                     size_t n_initial_pivots, int32_t recycle_pivots, t4a_gpu_ptt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        T4A_REQUIRE_PTR(cb);
        FullFunction f;
        f.builtin = false;
        f.cb = cb;
        f.ctx = ctx;
        *out = run_adaptive(local_dims, n_sites, f, initial_pivots, n_pivots, tci_options, patch_order, n_initial_pivots,
                            recycle_pivots);
    });
}

void t4a_gpu_ptt_release(t4a_gpu_ptt* h) { delete h; }

t4a_gpu_status t4a_gpu_ptt_len(const t4a_gpu_ptt* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl->patches.size();
    });
}

t4a_gpu_status t4a_gpu_ptt_projector(const t4a_gpu_ptt* h, size_t k, size_t* count, size_t* positions, size_t* values)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        if (k >= h->impl->patches.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, "patch index out of range");
        const auto& pr = h->impl->patches[k].projector;
        *count = pr.size();
        size_t i = 0;
        for (const auto& kv : pr) {
            if (positions) positions[i] = kv.first;
            if (values) values[i] = kv.second;
            ++i;
        }
    });
}

t4a_gpu_status t4a_gpu_ptt_patch_tt(const t4a_gpu_ptt* h, size_t k, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (k >= h->impl->patches.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, "patch index out of range");
        *out = new t4a_gpu_tt(h->impl->patches[k].cores, h->impl->eng.stream());
    });
}

t4a_gpu_status t4a_gpu_ptt_evaluate(t4a_gpu_ptt* h, const size_t* idx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        std::vector<uint32_t> u = narrow_indices(idx, checked_mul(n_pts, h->impl->dims.size(), "index buffer"));
        std::vector<double> v = h->impl->evaluate(u.data(), n_pts);
        std::memcpy(out, v.data(), n_pts * sizeof(double));
    });
}

// ------------------------------------------------------------------------------------------------ TreeTCI
t4a_gpu_status t4a_gpu_treetci_options_default(t4a_gpu_treetci_options* o)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(o);
        std::memset(o, 0, sizeof(*o));
        o->tolerance = 1e-8;
        o->max_iter = 20;
        o->max_bond_dim = 0;
        o->normalize_error = 1;
        o->enable_global_pivots = 1;
        o->nsearch = 5;
        o->max_nglobal_pivot = 5;
        o->tol_margin_global_search = 10.0;
        o->has_seed = 0;
        o->seed = 0;
    });
}

extern "C++" {
static TreeTciOptions convert_tree_options(const t4a_gpu_treetci_options* o)
{
    TreeTciOptions r;
    if (!o) return r;
    r.tolerance = o->tolerance;
    r.max_iter = o->max_iter;
    r.max_bond_dim = o->max_bond_dim;
    r.normalize_error = o->normalize_error != 0;
    r.enable_global_pivots = o->enable_global_pivots != 0;
    r.nsearch = o->nsearch;
    r.max_nglobal_pivot = o->max_nglobal_pivot;
    r.tol_margin_global_search = o->tol_margin_global_search;
    r.has_seed = o->has_seed != 0;
    r.seed = o->seed;
    return r;
}
static void write_index_set(const IndexSet& s, size_t* count, size_t* out)
{
    *count = s.count;
    if (out)
        for (size_t k = 0; k < s.count * s.width; ++k) out[k] = s.d[k];
}
static void write_history(const t4a_gpu_treetci* h, size_t* n_iter, size_t* ranks, double* errors)
{
    if (n_iter) *n_iter = h->impl.ranks_hist.size();
    for (size_t k = 0; k < h->impl.ranks_hist.size(); ++k) {
        if (ranks) ranks[k] = h->impl.ranks_hist[k];
        if (errors) errors[k] = h->impl.errors_hist[k];
    }
}
} // extern "C++"

t4a_gpu_status t4a_gpu_treetci_new(const size_t* local_dims, size_t n_sites, const size_t* edges, size_t n_edges,
                                   t4a_gpu_treetci** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_sites) T4A_REQUIRE_PTR(local_dims);
        if (n_edges) T4A_REQUIRE_PTR(edges);
        std::vector<TreeEdge> es;
        for (size_t k = 0; k < n_edges; ++k) es.emplace_back(edges[2 * k], edges[2 * k + 1]);
        TreeGraph g(n_sites, es); // graph errors are reported before the device is touched
        std::vector<size_t> d(local_dims, local_dims + n_sites);
        if (!(d.size() > 1)) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
        for (size_t s = 0; s < d.size(); ++s)
            if (d[s] == 0)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension at site " + std::to_string(s) + " must be positive");
        *out = new t4a_gpu_treetci(d, g);
    });
}

void t4a_gpu_treetci_release(t4a_gpu_treetci* h) { delete h; }

t4a_gpu_status t4a_gpu_treetci_set_builtin_function(t4a_gpu_treetci* h, int32_t fid, int32_t n_acc, const double* params,
                                                    const uint64_t* weights)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(params);
        T4A_REQUIRE_PTR(weights);
        h->impl.set_builtin(fid, n_acc, params, weights);
    });
}

t4a_gpu_status t4a_gpu_treetci_set_callback(t4a_gpu_treetci* h, t4a_gpu_batch_eval_fn cb, void* ctx)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.set_callback(cb, ctx);
    });
}

extern "C++" {
static std::vector<std::vector<uint32_t>> tree_pivots_from(const t4a_gpu_treetci* h, const size_t* pivots, size_t n_pivots)
{
    const size_t ns = h->impl.local_dims.size();
    std::vector<std::vector<uint32_t>> p(n_pivots, std::vector<uint32_t>(ns));
    for (size_t k = 0; k < n_pivots; ++k)
        for (size_t s = 0; s < ns; ++s) {
            const size_t v = pivots[s + ns * k];
            if (v > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "pivot value out of bounds");
            p[k][s] = (uint32_t)v;
        }
    return p;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_treetci_add_global_pivots(t4a_gpu_treetci* h, const size_t* pivots, size_t n_pivots)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pivots) T4A_REQUIRE_PTR(pivots);
        h->impl.add_global_pivots(tree_pivots_from(h, pivots, n_pivots));
    });
}

t4a_gpu_status t4a_gpu_treetci_set_proposer(t4a_gpu_treetci* h, int32_t kind, uint64_t seed)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (kind < 0 || kind > 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown proposer");
        h->impl.proposer = kind;
        h->impl.proposer_seed = seed;
    });
}

t4a_gpu_status t4a_gpu_treetci_subregion_vertices(const t4a_gpu_treetci* h, size_t u, size_t v, size_t* n_left,
                                                  size_t* left, size_t* n_right, size_t* right)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(n_left);
        T4A_REQUIRE_PTR(n_right);
        const auto keys = h->impl.graph.subregion_vertices(TreeEdge(u, v));
        *n_left = keys.first.size();
        *n_right = keys.second.size();
        if (left) std::copy(keys.first.begin(), keys.first.end(), left);
        if (right) std::copy(keys.second.begin(), keys.second.end(), right);
    });
}

t4a_gpu_status t4a_gpu_treetci_candidates(const t4a_gpu_treetci* h, size_t u, size_t v, size_t* n_left, size_t* left,
                                          size_t* n_right, size_t* right)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(n_left);
        T4A_REQUIRE_PTR(n_right);
        IndexSet l, r;
        h->impl.candidates(TreeEdge(u, v), l, r);
        write_index_set(l, n_left, left);
        write_index_set(r, n_right, right);
    });
}

t4a_gpu_status t4a_gpu_treetci_update_edge(t4a_gpu_treetci* h, size_t u, size_t v, size_t max_bond_dim, double rel_tol,
                                           double abs_tol, size_t* rank, size_t* rows, size_t* cols, double* pivot_errors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        RrLUOptions o;
        o.max_bond_dim = max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : max_bond_dim;
        o.rel_tol = rel_tol;
        o.abs_tol = abs_tol;
        o.left_orthogonal = true;
        const EdgeSelection sel = h->impl.update_edge(TreeEdge(u, v), o);
        if (rank) *rank = sel.rank;
        for (size_t k = 0; k < sel.rank; ++k) {
            if (rows) rows[k] = sel.row_indices[k];
            if (cols) cols[k] = sel.col_indices[k];
        }
        if (pivot_errors)
            for (size_t k = 0; k < sel.pivot_errors.size(); ++k) pivot_errors[k] = sel.pivot_errors[k];
    });
}

t4a_gpu_status t4a_gpu_treetci_optimize(t4a_gpu_treetci* h, const t4a_gpu_treetci_options* options, size_t* n_iter,
                                        size_t* ranks, double* errors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        const TreeTciOptions o = convert_tree_options(options);
        o.validate(); // before any callback runs (optimize/tests.rs:10-75)
        h->impl.optimize(o);
        write_history(h, n_iter, ranks, errors);
    });
}

t4a_gpu_status t4a_gpu_treetci_crossinterpolate2(t4a_gpu_treetci* h, const size_t* initial_pivots, size_t n_pivots,
                                                 const t4a_gpu_treetci_options* options, size_t* n_iter, size_t* ranks,
                                                 double* errors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        const TreeTciOptions o = convert_tree_options(options);
        o.validate();
        h->impl.crossinterpolate2(tree_pivots_from(h, initial_pivots, n_pivots), o);
        write_history(h, n_iter, ranks, errors);
    });
}

t4a_gpu_status t4a_gpu_treetci_find_global_pivots(t4a_gpu_treetci* h, size_t nsearch, size_t max_nglobal_pivot,
                                                  double tol_margin, double abs_tol, uint64_t seed, size_t* count,
                                                  size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        const auto pv = h->impl.find_global_pivots(nsearch, max_nglobal_pivot, tol_margin, abs_tol, seed);
        *count = pv.size();
        const size_t ns = h->impl.local_dims.size();
        if (out)
            for (size_t k = 0; k < pv.size(); ++k)
                for (size_t s = 0; s < ns; ++s) out[s + ns * k] = pv[k][s];
    });
}

t4a_gpu_status t4a_gpu_treetci_pivots(const t4a_gpu_treetci* h, const size_t* key, size_t key_len, size_t* count,
                                      size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        if (key_len) T4A_REQUIRE_PTR(key);
        SubtreeKey k(key, key + key_len);
        std::sort(k.begin(), k.end());
        k.erase(std::unique(k.begin(), k.end()), k.end());
        write_index_set(h->impl.pivots_of(k), count, out);
    });
}

t4a_gpu_status t4a_gpu_treetci_bond_errors(const t4a_gpu_treetci* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        size_t k = 0;
        for (const auto& kv : h->impl.bond_errors) out[k++] = kv.second;
    });
}

t4a_gpu_status t4a_gpu_treetci_pivot_errors(const t4a_gpu_treetci* h, size_t* count, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        *count = h->impl.pivot_errors.size();
        if (out) std::copy(h->impl.pivot_errors.begin(), h->impl.pivot_errors.end(), out);
    });
}

t4a_gpu_status t4a_gpu_treetci_flush_pivot_errors(t4a_gpu_treetci* h)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.flush_pivot_errors();
    });
}

t4a_gpu_status t4a_gpu_treetci_max_sample_value(const t4a_gpu_treetci* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.max_sample_value;
    });
}

t4a_gpu_status t4a_gpu_treetci_set_max_sample_value(t4a_gpu_treetci* h, double value)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.max_sample_value = value;
    });
}

t4a_gpu_status t4a_gpu_treetci_max_bond_error(const t4a_gpu_treetci* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.max_bond_error();
    });
}

t4a_gpu_status t4a_gpu_treetci_max_bond_dim(const t4a_gpu_treetci* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl.max_bond_dim();
    });
}

t4a_gpu_status t4a_gpu_treetci_materialize(t4a_gpu_treetci* h, size_t center_site)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.materialize(center_site);
    });
}

t4a_gpu_status t4a_gpu_treetci_site_tensor(t4a_gpu_treetci* h, size_t site, size_t* ndims, size_t* dims, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(ndims);
        std::vector<size_t> d;
        const std::vector<double> v = h->impl.site_tensor_host(site, d);
        *ndims = d.size();
        if (dims) std::copy(d.begin(), d.end(), dims);
        if (out && !v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(double));
    });
}

t4a_gpu_status t4a_gpu_treetci_evaluate(t4a_gpu_treetci* h, const size_t* idx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(idx);
        T4A_REQUIRE_PTR(out);
        std::vector<uint32_t> u = narrow_indices(idx, checked_mul(n_pts, h->impl.local_dims.size(), "index buffer"));
        std::vector<double> v = h->impl.evaluate(u.data(), n_pts);
        std::memcpy(out, v.data(), n_pts * sizeof(double));
    });
}

// ------------------------------------------------------------------------------------------------ quantics front end
t4a_gpu_status t4a_gpu_qtci_options_default(t4a_gpu_qtci_options* o)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(o);
        std::memset(o, 0, sizeof(*o));
        o->tolerance = 1e-8;
        o->max_bond_dim = 0;
        o->max_iter = 200;
        o->n_random_init_pivot = 5;
        o->unfolding_scheme = 0;
        o->normalize_error = 1;
        o->has_seed = 0;
        o->seed = 0;
    });
}

extern "C++" {
static QtciOptions convert_qtci_options(const t4a_gpu_qtci_options* o)
{
    QtciOptions r;
    if (!o) return r;
    r.tolerance = o->tolerance;
    r.max_bond_dim = o->max_bond_dim;
    r.max_iter = o->max_iter;
    r.n_random_init_pivot = o->n_random_init_pivot;
    if (o->unfolding_scheme != 0 && o->unfolding_scheme != 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown unfolding scheme");
    r.unfolding = o->unfolding_scheme ? Unfolding::Fused : Unfolding::Interleaved;
    r.normalize_error = o->normalize_error != 0;
    r.has_seed = o->has_seed != 0;
    r.seed = o->seed;
    r.to_treetci_options().validate();
    return r;
}
static std::vector<std::vector<size_t>> qtci_pivots(const size_t* pivots, size_t n_pivots, size_t n_vars)
{
    std::vector<std::vector<size_t>> p;
    for (size_t k = 0; k < n_pivots; ++k) p.emplace_back(pivots + k * n_vars, pivots + (k + 1) * n_vars);
    return p;
}
static void qtci_run(t4a_gpu_qtci** out, std::unique_ptr<QuanticsTci> q, int32_t has_pivots, const size_t* pivots,
                     size_t n_pivots, const QtciOptions& o)
{
    const auto pv = qtci_pivots(pivots, has_pivots ? n_pivots : 0, q->grid.n_vars());
    q->run(has_pivots ? &pv : nullptr, o);
    auto* h = new t4a_gpu_qtci();
    h->impl = std::move(q);
    *out = h;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_quanticscrossinterpolate(const size_t* rs, size_t n_vars, const double* lower, const double* upper,
                                                int32_t include_endpoint, int32_t grid_unfolding, t4a_gpu_coord_eval_fn f,
                                                void* ctx, int32_t has_pivots, const size_t* initial_pivots, size_t n_pivots,
                                                const t4a_gpu_qtci_options* options, t4a_gpu_qtci** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        T4A_REQUIRE_PTR(rs);
        T4A_REQUIRE_PTR(f);
        if (has_pivots && n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        const QtciOptions o = convert_qtci_options(options);
        QuanticsGrid grid(std::vector<size_t>(rs, rs + n_vars), grid_unfolding ? Unfolding::Fused : Unfolding::Interleaved, true,
                          lower ? std::vector<double>(lower, lower + n_vars) : std::vector<double>(),
                          upper ? std::vector<double>(upper, upper + n_vars) : std::vector<double>(), include_endpoint != 0);
        qtci_run(out, std::unique_ptr<QuanticsTci>(new QuanticsTci(grid, f, nullptr, ctx, {})), has_pivots, initial_pivots,
                 n_pivots, o);
    });
}

t4a_gpu_status t4a_gpu_quanticscrossinterpolate_discrete(const size_t* sizes, size_t n_vars, t4a_gpu_grididx_eval_fn f,
                                                         void* ctx, int32_t has_pivots, const size_t* initial_pivots,
                                                         size_t n_pivots, const t4a_gpu_qtci_options* options,
                                                         t4a_gpu_qtci** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        T4A_REQUIRE_PTR(f);
        if (n_vars == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "this method requires at least one grid dimension, got an empty size");
        T4A_REQUIRE_PTR(sizes);
        if (has_pivots && n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        const QtciOptions o = convert_qtci_options(options);
        const std::vector<size_t> sz(sizes, sizes + n_vars);
        for (size_t s : sz)
            if (s == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "this method only supports grid sizes that are powers of 2");
        qtci_check_sizes(sz);
        const size_t r = (size_t)std::log2((double)sz[0]);
        QuanticsGrid grid(std::vector<size_t>(n_vars, r), o.unfolding, false);
        qtci_run(out, std::unique_ptr<QuanticsTci>(new QuanticsTci(grid, nullptr, f, ctx, {})), has_pivots, initial_pivots,
                 n_pivots, o);
    });
}

t4a_gpu_status t4a_gpu_quanticscrossinterpolate_from_arrays(const double* xvals, const size_t* sizes, size_t n_vars,
                                                            t4a_gpu_coord_eval_fn f, void* ctx, int32_t has_pivots,
                                                            const size_t* initial_pivots, size_t n_pivots,
                                                            const t4a_gpu_qtci_options* options, t4a_gpu_qtci** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        T4A_REQUIRE_PTR(f);
        if (n_vars == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "xvals must not be empty");
        T4A_REQUIRE_PTR(sizes);
        if (has_pivots && n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        const QtciOptions o = convert_qtci_options(options);
        std::vector<std::vector<double>> xv;
        size_t off = 0;
        for (size_t d = 0; d < n_vars; ++d) {
            if (sizes[d]) T4A_REQUIRE_PTR(xvals);
            xv.emplace_back(xvals + off, xvals + off + sizes[d]);
            off += sizes[d];
        }
        const bool uniform = qtci_check_xvals_uniform(xv);
        std::vector<size_t> rs;
        for (const auto& x : xv) rs.push_back((size_t)std::log2((double)x.size()));
        if (uniform) {
            std::vector<double> lo, up;
            for (const auto& x : xv) {
                lo.push_back(x.front());
                up.push_back(x.back());
            }
            QuanticsGrid grid(rs, o.unfolding, true, lo, up, true);
            qtci_run(out, std::unique_ptr<QuanticsTci>(new QuanticsTci(grid, f, nullptr, ctx, {})), has_pivots, initial_pivots,
                     n_pivots, o);
        } else {
            QuanticsGrid grid(rs, o.unfolding, false);
            qtci_run(out, std::unique_ptr<QuanticsTci>(new QuanticsTci(grid, f, nullptr, ctx, xv)), has_pivots, initial_pivots,
                     n_pivots, o);
        }
    });
}

void t4a_gpu_qtci_release(t4a_gpu_qtci* h) { delete h; }

t4a_gpu_status t4a_gpu_qtci_evaluate(t4a_gpu_qtci* h, const size_t* grididx, size_t n_pts, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pts == 0) return;
        T4A_REQUIRE_PTR(grididx);
        T4A_REQUIRE_PTR(out);
        const std::vector<double> v = h->impl->evaluate(grididx, n_pts);
        std::memcpy(out, v.data(), n_pts * sizeof(double));
    });
}

t4a_gpu_status t4a_gpu_qtci_sum(t4a_gpu_qtci* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl->sum();
    });
}

t4a_gpu_status t4a_gpu_qtci_integral(t4a_gpu_qtci* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = h->impl->integral();
    });
}

t4a_gpu_status t4a_gpu_qtci_n_sites(const t4a_gpu_qtci* h, size_t* n_sites, size_t* n_vars, int32_t* is_discretized)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_sites) *n_sites = h->impl->grid.n_sites();
        if (n_vars) *n_vars = h->impl->grid.n_vars();
        if (is_discretized) *is_discretized = h->impl->grid.discretized ? 1 : 0;
    });
}

t4a_gpu_status t4a_gpu_qtci_link_dims(const t4a_gpu_qtci* h, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        const auto ld = h->impl->tt->link_dims();
        if (!ld.empty()) T4A_REQUIRE_PTR(out);
        std::copy(ld.begin(), ld.end(), out);
    });
}

t4a_gpu_status t4a_gpu_qtci_history(const t4a_gpu_qtci* h, size_t* n_iter, size_t* ranks, double* errors)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        const auto& t = *h->impl->tci;
        if (n_iter) *n_iter = t.ranks_hist.size();
        for (size_t k = 0; k < t.ranks_hist.size(); ++k) {
            if (ranks) ranks[k] = t.ranks_hist[k];
            if (errors) errors[k] = t.errors_hist[k];
        }
    });
}

t4a_gpu_status t4a_gpu_qtci_tensor_train(t4a_gpu_qtci* h, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        *out = new t4a_gpu_tt(h->impl->tt->cores, h->impl->tt->eng.stream());
    });
}

t4a_gpu_status t4a_gpu_qtci_tree_pivots(const t4a_gpu_qtci* h, const size_t* key, size_t key_len, size_t* count, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        if (key_len) T4A_REQUIRE_PTR(key);
        SubtreeKey k(key, key + key_len);
        std::sort(k.begin(), k.end());
        write_index_set(h->impl->tci->pivots_of(k), count, out);
    });
}

t4a_gpu_status t4a_gpu_qtci_cachedata(const t4a_gpu_qtci* h, size_t* count, size_t* quantics, double* values,
                                      size_t* user_calls, size_t* user_points)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        *count = h->impl->cache.size();
        if (user_calls) *user_calls = h->impl->n_user_calls;
        if (user_points) *user_points = h->impl->n_user_points;
        if (!quantics && !values) return;
        size_t k = 0;
        const size_t ns = h->impl->grid.n_sites();
        for (const auto& kv : h->impl->cache) {
            if (quantics)
                for (size_t s = 0; s < ns; ++s) quantics[k * ns + s] = kv.first[s];
            if (values) values[k] = kv.second;
            ++k;
        }
    });
}

t4a_gpu_status t4a_gpu_qtci_grid(const t4a_gpu_qtci* h, int32_t which, const size_t* in, size_t* out_u, double* out_d)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        const QuanticsGrid& g = h->impl->grid;
        const size_t ns = g.n_sites(), nv = g.n_vars();
        if (which == 0) {
            T4A_REQUIRE_PTR(in);
            T4A_REQUIRE_PTR(out_u);
            std::vector<uint32_t> q(ns);
            g.grididx_to_quantics(in, q.data());
            std::copy(q.begin(), q.end(), out_u);
        } else if (which == 1 || which == 2) {
            T4A_REQUIRE_PTR(in);
            std::vector<uint32_t> q = narrow_indices(in, ns);
            if (which == 1) {
                T4A_REQUIRE_PTR(out_u);
                g.quantics_to_grididx(q.data(), out_u);
            } else {
                T4A_REQUIRE_PTR(out_d);
                // cachedata_origcoord (quantics_tci.rs:156-173): only discretized grids carry coordinates
                if (!g.discretized)
                    throw Error(T4A_GPU_INVALID_ARGUMENT, "original coordinates are only available for discretized grids");
                g.quantics_to_origcoord(q.data(), out_d);
            }
        } else if (which == 3) {
            T4A_REQUIRE_PTR(out_u);
            const auto d = g.local_dimensions();
            std::copy(d.begin(), d.end(), out_u);
        } else if (which == 4) {
            T4A_REQUIRE_PTR(out_d);
            const auto st = g.grid_step();
            std::copy(st.begin(), st.end(), out_d);
        } else {
            throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown grid query");
        }
        (void)nv;
    });
}

t4a_gpu_status t4a_gpu_quanticscrossinterpolate_batched(const size_t* rs, size_t n_vars, const double* lower,
                                                        const double* upper, int32_t include_endpoint, int32_t grid_unfolding,
                                                        t4a_gpu_coord_eval_vec_fn f, void* ctx, const size_t* output_dims,
                                                        size_t n_output_dims, int32_t has_pivots, const size_t* initial_pivots,
                                                        size_t n_pivots, const t4a_gpu_qtci_options* options,
                                                        t4a_gpu_tt** out_tt, size_t* n_iter, size_t* ranks, double* errors,
                                                        size_t* user_points)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out_tt);
        *out_tt = nullptr;
        T4A_REQUIRE_PTR(rs);
        T4A_REQUIRE_PTR(f);
        if (n_output_dims) T4A_REQUIRE_PTR(output_dims);
        if (has_pivots && n_pivots) T4A_REQUIRE_PTR(initial_pivots);
        const QtciOptions o = convert_qtci_options(options);
        QuanticsGrid grid(std::vector<size_t>(rs, rs + n_vars), grid_unfolding ? Unfolding::Fused : Unfolding::Interleaved, true,
                          lower ? std::vector<double>(lower, lower + n_vars) : std::vector<double>(),
                          upper ? std::vector<double>(upper, upper + n_vars) : std::vector<double>(), include_endpoint != 0);
        const auto pv = qtci_pivots(initial_pivots, has_pivots ? n_pivots : 0, n_vars);
        QuanticsBatchedResult r = quantics_batched(grid, f, ctx, std::vector<size_t>(output_dims, output_dims + n_output_dims),
                                                   has_pivots ? &pv : nullptr, o);
        if (n_iter) *n_iter = r.ranks.size();
        for (size_t k = 0; k < r.ranks.size(); ++k) {
            if (ranks) ranks[k] = r.ranks[k];
            if (errors) errors[k] = r.errors[k];
        }
        if (user_points) *user_points = r.n_user_points;
        *out_tt = new t4a_gpu_tt(r.tt->cores, r.tt->eng.stream());
    });
}

// ------------------------------------------------------------------------------------------------ dense labelled tensors
extern "C++" {
static SvdPolicy convert_policy(const t4a_gpu_svd_policy* p)
{
    SvdPolicy r;
    if (!p) return r;
    r.threshold = p->threshold;
    r.scale = p->scale;
    r.measure = p->measure;
    r.rule = p->rule;
    if (r.scale < 0 || r.scale > 1 || r.measure < 0 || r.measure > 1 || r.rule < 0 || r.rule > 1)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown SVD truncation policy field");
    return r;
}
static TensorView host_view(const size_t* dims, const int64_t* labels, size_t rank)
{
    TensorView v;
    v.d_data = nullptr;
    if (rank) {
        T4A_REQUIRE_PTR(dims);
        T4A_REQUIRE_PTR(labels);
    }
    v.dims.assign(dims, dims + rank);
    v.labels.assign(labels, labels + rank);
    size_t n = 1;
    for (size_t d : v.dims) n = checked_mul(n, d, "tensor shape");
    return v;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_svd_retained_rank(const double* s, size_t n, const t4a_gpu_svd_policy* policy, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        if (n) T4A_REQUIRE_PTR(s);
        *out = svd_retained_rank(s, n, convert_policy(policy));
    });
}

t4a_gpu_status t4a_gpu_qr_retained_rank(const double* r, size_t k, size_t n, double rtol, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        if (checked_mul(k, n, "R shape")) T4A_REQUIRE_PTR(r);
        *out = qr_retained_rank(r, k, n, rtol);
    });
}

t4a_gpu_status t4a_gpu_tensor_contract_f64(const double* a, const size_t* a_dims, const int64_t* a_labels, size_t a_rank,
                                           const double* b, const size_t* b_dims, const int64_t* b_labels, size_t b_rank,
                                           double* out, size_t* out_dims, int64_t* out_labels, size_t* out_rank)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out_rank);
        TensorView va = host_view(a_dims, a_labels, a_rank), vb = host_view(b_dims, b_labels, b_rank);
        const ContractPlan plan = plan_contract_pair(va, vb); // index errors before the device is touched
        *out_rank = plan.out_dims.size();
        for (size_t k = 0; k < plan.out_dims.size(); ++k) {
            if (out_dims) out_dims[k] = plan.out_dims[k];
            if (out_labels) out_labels[k] = plan.out_labels[k];
        }
        if (!out) return;
        const size_t na = va.size(), nb = vb.size(), nc = checked_mul(plan.M, plan.N, "result shape");
        if (nc == 0) return;
        if (na) T4A_REQUIRE_PTR(a);
        if (nb) T4A_REQUIRE_PTR(b);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_in = e.pi(na + nb + nc);
        upload(e, d_in, a, na);
        upload(e, d_in + na, b, nb);
        va.d_data = d_in;
        vb.d_data = d_in + na;
        tensor_contract_pair(e, va, vb, plan, d_in + na + nb);
        download(e, out, d_in + na + nb, nc);
    });
}

t4a_gpu_status t4a_gpu_tensor_svd_f64(const double* t, const size_t* dims, const int64_t* labels, size_t rank,
                                      const int64_t* left_labels, size_t n_left, int32_t truncate,
                                      const t4a_gpu_svd_policy* policy, int32_t has_max_bond_dim, size_t max_bond_dim,
                                      size_t* r, double* u, double* s, double* v)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(r);
        TensorView tv = host_view(dims, labels, rank);
        if (n_left) T4A_REQUIRE_PTR(left_labels);
        const UnfoldPlan un = plan_unfold_split(tv, std::vector<int64_t>(left_labels, left_labels + n_left));
        SvdPolicy pol = convert_policy(policy);
        if (truncate) { // validated before any linear algebra (svd/tests/mod.rs:120-160)
            if (has_max_bond_dim && max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be positive when specified");
            if (!std::isfinite(pol.threshold) || pol.threshold < 0.0)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid SVD truncation threshold: threshold must be finite and non-negative");
        }
        const size_t m = un.m, n = un.n, k = std::min(m, n), count = tv.size();
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD of an empty tensor");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "svd: unfolded dimensions above 65535 are not supported");
        T4A_REQUIRE_PTR(t);
        T4A_REQUIRE_PTR(u);
        T4A_REQUIRE_PTR(s);
        T4A_REQUIRE_PTR(v);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_t = e.pi(2 * count);
        upload(e, d_t, t, count);
        tv.d_data = d_t;
        double* d_mat = d_t + count;
        tensor_permute(e, tv, un.perm, d_mat);
        e.d_tmp.reserve(m * k + k + k * n + n * k);
        double* d_u = e.d_tmp.get();
        double* d_s = d_u + m * k;
        double* d_vt = d_s + k;
        double* d_v = d_vt + k * n;
        e.svd(d_mat, (int)m, (int)n, d_u, d_s, d_vt);
        std::vector<double> hs(k);
        download(e, hs.data(), d_s, k);
        size_t keep = k;
        if (truncate) {
            keep = svd_retained_rank(hs.data(), k, pol);
            if (has_max_bond_dim) keep = std::min(keep, max_bond_dim);
            keep = std::max<size_t>(keep, 1);
        } else {
            keep = std::max<size_t>(k, 1);
        }
        keep = std::min(keep, k);
        *r = keep;
        std::copy(hs.begin(), hs.begin() + keep, s);
        download(e, u, d_u, m * keep);
        // V [n x keep] = (first `keep` rows of V^T)^T
        transpose_launch(d_vt, (int)keep, (int)n, (int)k, d_v, (int)n, e.stream());
        T4A_HIP(hipGetLastError());
        download(e, v, d_v, n * keep);
    });
}

t4a_gpu_status t4a_gpu_tensor_qr_f64(const double* t, const size_t* dims, const int64_t* labels, size_t rank,
                                     const int64_t* left_labels, size_t n_left, int32_t truncate, int32_t has_rtol, double rtol,
                                     size_t* r, double* q, double* r_factor)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(r);
        TensorView tv = host_view(dims, labels, rank);
        if (n_left) T4A_REQUIRE_PTR(left_labels);
        const UnfoldPlan un = plan_unfold_split(tv, std::vector<int64_t>(left_labels, left_labels + n_left));
        const double tol = has_rtol ? rtol : 1e-15; // default_qr_rtol (qr.rs:62-66)
        if (truncate && (!std::isfinite(tol) || tol < 0.0))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid rtol value: rtol must be finite and non-negative");
        const size_t m = un.m, n = un.n, k = std::min(m, n), count = tv.size();
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "QR of an empty tensor");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "qr: unfolded dimensions above 65535 are not supported");
        T4A_REQUIRE_PTR(t);
        T4A_REQUIRE_PTR(q);
        T4A_REQUIRE_PTR(r_factor);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_t = e.pi(2 * count);
        upload(e, d_t, t, count);
        tv.d_data = d_t;
        double* d_mat = d_t + count;
        tensor_permute(e, tv, un.perm, d_mat);
        e.d_tmp.reserve(m * k + 2 * k * n);
        double* d_q = e.d_tmp.get();
        double* d_r = d_q + m * k;
        double* d_rk = d_r + k * n;
        e.qr(d_mat, (int)m, (int)n, d_q, d_r);
        std::vector<double> hr(k * n);
        download(e, hr.data(), d_r, k * n);
        size_t keep = k;
        if (truncate) keep = std::min(qr_retained_rank(hr.data(), k, n, tol), k);
        *r = keep;
        download(e, q, d_q, m * keep);
        // leading `keep` rows of R with leading dimension keep
        gather_launch(d_r, (int)k, nullptr, (int)keep, nullptr, (int)n, d_rk, (int)keep, e.stream());
        T4A_HIP(hipGetLastError());
        download(e, r_factor, d_rk, keep * n);
    });
}

// ------------------------------------------------------------------------------------------------ device-resident labelled tensors
extern "C++" {
static std::unique_ptr<t4a_gpu_tensor> make_tensor(const std::vector<size_t>& dims, const std::vector<int64_t>& labels)
{
    std::unique_ptr<t4a_gpu_tensor> t(new t4a_gpu_tensor());
    t->dims = dims;
    t->labels = labels;
    t->buf.reserve(std::max<size_t>(t->size(), 1));
    return t;
}
} // extern "C++"

t4a_gpu_status t4a_gpu_tensor_new(const double* data, const size_t* dims, const int64_t* labels, size_t rank,
                                  t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        TensorView hv = host_view(dims, labels, rank);
        if (rank > (size_t)TENSOR_MAX_RANK) throw Error(T4A_GPU_NOT_IMPLEMENTED, "tensor rank above 16");
        for (size_t a = 0; a < rank; ++a)
            for (size_t b = a + 1; b < rank; ++b)
                if (labels[a] == labels[b]) throw Error(T4A_GPU_INVALID_ARGUMENT, "duplicate index in tensor");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        auto t = make_tensor(hv.dims, hv.labels);
        if (t->size()) {
            T4A_REQUIRE_PTR(data);
            upload(e, t->buf.get(), data, t->size());
        }
        *out = t.release();
    });
}

void t4a_gpu_tensor_release(t4a_gpu_tensor* h) { delete h; }

t4a_gpu_status t4a_gpu_tensor_rank(const t4a_gpu_tensor* h, size_t* rank)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(rank);
        *rank = h->dims.size();
    });
}

t4a_gpu_status t4a_gpu_tensor_dims(const t4a_gpu_tensor* h, size_t* dims, int64_t* labels)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (dims) std::copy(h->dims.begin(), h->dims.end(), dims);
        if (labels) std::copy(h->labels.begin(), h->labels.end(), labels);
    });
}

t4a_gpu_status t4a_gpu_tensor_to_host(const t4a_gpu_tensor* h, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (h->size() == 0) return;
        T4A_REQUIRE_PTR(out);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        download(dense_engine(), out, h->buf.get(), h->size());
    });
}

t4a_gpu_status t4a_gpu_tensor_permute(const t4a_gpu_tensor* h, const int64_t* labels, t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        const size_t r = h->dims.size();
        if (r) T4A_REQUIRE_PTR(labels);
        std::vector<size_t> perm(r);
        std::vector<char> used(r, 0);
        for (size_t k = 0; k < r; ++k) {
            auto it = std::find(h->labels.begin(), h->labels.end(), labels[k]);
            if (it == h->labels.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "permute: index not found in tensor");
            perm[k] = (size_t)(it - h->labels.begin());
            if (used[perm[k]]) throw Error(T4A_GPU_INVALID_ARGUMENT, "permute: duplicate index");
            used[perm[k]] = 1;
        }
        std::vector<size_t> nd(r);
        std::vector<int64_t> nl(r);
        for (size_t k = 0; k < r; ++k) {
            nd[k] = h->dims[perm[k]];
            nl[k] = h->labels[perm[k]];
        }
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        auto t = make_tensor(nd, nl);
        tensor_permute(e, h->view(), perm, t->buf.get());
        e.sync();
        *out = t.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_relabel(t4a_gpu_tensor* h, int64_t from, int64_t to)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        auto it = std::find(h->labels.begin(), h->labels.end(), from);
        if (it == h->labels.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "relabel: index not found in tensor");
        if (from != to && std::find(h->labels.begin(), h->labels.end(), to) != h->labels.end())
            throw Error(T4A_GPU_INVALID_ARGUMENT, "relabel: the new label is already present");
        *it = to;
    });
}

t4a_gpu_status t4a_gpu_tensor_contract(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        const TensorView va = a->view(), vb = b->view();
        const ContractPlan plan = plan_contract_pair(va, vb);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        auto t = make_tensor(plan.out_dims, plan.out_labels);
        tensor_contract_pair(e, va, vb, plan, t->buf.get());
        e.sync();
        *out = t.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_contract_many(const t4a_gpu_tensor* const* tensors, size_t n_tensors, const int64_t* retain_labels,
                                            size_t n_retain, t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_tensors) T4A_REQUIRE_PTR(tensors);
        if (n_retain) T4A_REQUIRE_PTR(retain_labels);
        std::vector<TensorView> views;
        for (size_t i = 0; i < n_tensors; ++i) {
            T4A_REQUIRE_PTR(tensors[i]);
            views.push_back(tensors[i]->view());
        }
        const std::vector<int64_t> retain(retain_labels, retain_labels + n_retain);
        (void)plan_contract_network(views, retain); // (argument errors before the device is touched)
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        OwnedTensor r = tensor_contract_network(dense_engine(), views, retain);
        auto t = std::make_unique<t4a_gpu_tensor>();
        t->buf = std::move(r.buf);
        t->dims = std::move(r.dims);
        t->labels = std::move(r.labels);
        *out = t.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_outer_product(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        const TensorView va = a->view(), vb = b->view();
        for (int64_t l : va.labels)
            if (std::find(vb.labels.begin(), vb.labels.end(), l) != vb.labels.end())
                throw Error(T4A_GPU_INVALID_ARGUMENT, "outer_product: the operands share an index; use contract for a contraction");
        const ContractPlan plan = plan_contract_pair(va, vb); // (no common label: M x 1 times 1 x N)
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        auto t = make_tensor(plan.out_dims, plan.out_labels);
        tensor_contract_pair(e, va, vb, plan, t->buf.get());
        e.sync();
        *out = t.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_tensordot(const t4a_gpu_tensor* a, const t4a_gpu_tensor* b, const int64_t* labels_a, const int64_t* labels_b,
                                        size_t n_pairs, t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_pairs == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: No pairs specified for contraction");
        T4A_REQUIRE_PTR(labels_a);
        T4A_REQUIRE_PTR(labels_b);
        const TensorView va = a->view();
        TensorView vb = b->view();
        // prepare_contraction_pairs (index_ops.rs): every pair names one axis of each operand, no axis twice, equal dimensions; a label the
        // operands have in common that is not paired would be a batch contraction (not implemented in the reference either)
        std::vector<size_t> ax_a, ax_b;
        for (size_t p = 0; p < n_pairs; ++p) {
            const auto ia = std::find(va.labels.begin(), va.labels.end(), labels_a[p]);
            const auto ib = std::find(vb.labels.begin(), vb.labels.end(), labels_b[p]);
            if (ia == va.labels.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: Index not found in self tensor");
            if (ib == vb.labels.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: Index not found in other tensor");
            const size_t pa = (size_t)(ia - va.labels.begin()), pb = (size_t)(ib - vb.labels.begin());
            if (std::find(ax_a.begin(), ax_a.end(), pa) != ax_a.end())
                throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: Duplicate axis " + std::to_string(pa) + " in self tensor");
            if (std::find(ax_b.begin(), ax_b.end(), pb) != ax_b.end())
                throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: Duplicate axis " + std::to_string(pb) + " in other tensor");
            if (va.dims[pa] != vb.dims[pb])
                throw Error(T4A_GPU_INVALID_ARGUMENT, "tensordot: Dimension mismatch: self[" + std::to_string(pa) + "]=" + std::to_string(va.dims[pa]) +
                                                          " != other[" + std::to_string(pb) + "]=" + std::to_string(vb.dims[pb]));
            ax_a.push_back(pa);
            ax_b.push_back(pb);
        }
        for (size_t i = 0; i < va.labels.size(); ++i)
            for (size_t j = 0; j < vb.labels.size(); ++j)
                if (va.labels[i] == vb.labels[j]) {
                    bool paired = false;
                    for (size_t p = 0; p < n_pairs; ++p) paired = paired || (ax_a[p] == i && ax_b[p] == j);
                    if (!paired)
                        throw Error(T4A_GPU_NOT_IMPLEMENTED,
                                    "tensordot: Common index found but not in contraction pairs. Batch contraction is not yet implemented.");
                }
        // the paired axes of `b` take the labels of their partners; unpaired axes keep theirs (none is shared, checked above)
        for (size_t p = 0; p < n_pairs; ++p) vb.labels[ax_b[p]] = va.labels[ax_a[p]];
        const ContractPlan plan = plan_contract_pair(va, vb);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        auto t = make_tensor(plan.out_dims, plan.out_labels);
        tensor_contract_pair(e, va, vb, plan, t->buf.get());
        e.sync();
        *out = t.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_svd(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t truncate,
                                  const t4a_gpu_svd_policy* policy, int32_t has_max_bond_dim, size_t max_bond_dim,
                                  int64_t bond_label, int64_t bond_label_v, t4a_gpu_tensor** u, t4a_gpu_tensor** s,
                                  t4a_gpu_tensor** v)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(t);
        T4A_REQUIRE_PTR(u);
        T4A_REQUIRE_PTR(s);
        T4A_REQUIRE_PTR(v);
        *u = *s = *v = nullptr;
        if (n_left) T4A_REQUIRE_PTR(left_labels);
        const TensorView tv = t->view();
        const std::vector<int64_t> left(left_labels, left_labels + n_left);
        const UnfoldPlan un = plan_unfold_split(tv, left);
        SvdPolicy pol = convert_policy(policy);
        if (truncate) {
            if (has_max_bond_dim && max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be positive when specified");
            if (!std::isfinite(pol.threshold) || pol.threshold < 0.0)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid SVD truncation threshold: threshold must be finite and non-negative");
        }
        const size_t m = un.m, n = un.n, k = std::min(m, n), count = tv.size();
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "SVD of an empty tensor");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "svd: unfolded dimensions above 65535 are not supported");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_mat = e.pi(count);
        tensor_permute(e, tv, un.perm, d_mat);
        e.d_tmp.reserve(m * k + k + k * n);
        double* d_u = e.d_tmp.get();
        double* d_s = d_u + m * k;
        double* d_vt = d_s + k;
        e.svd(d_mat, (int)m, (int)n, d_u, d_s, d_vt);
        std::vector<double> hs(k);
        download(e, hs.data(), d_s, k);
        size_t keep = k;
        if (truncate) {
            keep = svd_retained_rank(hs.data(), k, pol);
            if (has_max_bond_dim) keep = std::min(keep, max_bond_dim);
        }
        keep = std::min(std::max<size_t>(keep, 1), k);
        std::vector<size_t> ud = un.left_dims, vd = un.right_dims;
        std::vector<int64_t> ul = left, vl;
        for (size_t a = n_left; a < un.perm.size(); ++a) vl.push_back(tv.labels[un.perm[a]]);
        ud.push_back(keep);
        ul.push_back(bond_label);
        vd.push_back(keep);
        vl.push_back(bond_label_v);
        auto tu = make_tensor(ud, ul), ts = make_tensor({keep}, {bond_label}), tvv = make_tensor(vd, vl);
        T4A_HIP(hipMemcpyAsync(tu->buf.get(), d_u, m * keep * sizeof(double), hipMemcpyDeviceToDevice, e.stream()));
        T4A_HIP(hipMemcpyAsync(ts->buf.get(), d_s, keep * sizeof(double), hipMemcpyDeviceToDevice, e.stream()));
        transpose_launch(d_vt, (int)keep, (int)n, (int)k, tvv->buf.get(), (int)n, e.stream());
        T4A_HIP(hipGetLastError());
        e.sync();
        *u = tu.release();
        *s = ts.release();
        *v = tvv.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_qr(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t truncate,
                                 int32_t has_rtol, double rtol, int64_t bond_label, t4a_gpu_tensor** q, t4a_gpu_tensor** r)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(t);
        T4A_REQUIRE_PTR(q);
        T4A_REQUIRE_PTR(r);
        *q = *r = nullptr;
        if (n_left) T4A_REQUIRE_PTR(left_labels);
        const TensorView tv = t->view();
        const std::vector<int64_t> left(left_labels, left_labels + n_left);
        const UnfoldPlan un = plan_unfold_split(tv, left);
        const double tol = has_rtol ? rtol : 1e-15;
        if (truncate && (!std::isfinite(tol) || tol < 0.0))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid rtol value: rtol must be finite and non-negative");
        const size_t m = un.m, n = un.n, k = std::min(m, n), count = tv.size();
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "QR of an empty tensor");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "qr: unfolded dimensions above 65535 are not supported");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        double* d_mat = e.pi(count);
        tensor_permute(e, tv, un.perm, d_mat);
        e.d_tmp.reserve(m * k + k * n);
        double* d_q = e.d_tmp.get();
        double* d_r = d_q + m * k;
        e.qr(d_mat, (int)m, (int)n, d_q, d_r);
        size_t keep = k;
        if (truncate) {
            std::vector<double> hr(k * n);
            download(e, hr.data(), d_r, k * n);
            keep = std::min(qr_retained_rank(hr.data(), k, n, tol), k);
        }
        std::vector<size_t> qd = un.left_dims, rd{keep};
        std::vector<int64_t> ql = left, rl{bond_label};
        qd.push_back(keep);
        ql.push_back(bond_label);
        for (size_t a = n_left; a < un.perm.size(); ++a) {
            rd.push_back(tv.dims[un.perm[a]]);
            rl.push_back(tv.labels[un.perm[a]]);
        }
        auto tq = make_tensor(qd, ql), tr = make_tensor(rd, rl);
        T4A_HIP(hipMemcpyAsync(tq->buf.get(), d_q, m * keep * sizeof(double), hipMemcpyDeviceToDevice, e.stream()));
        gather_launch(d_r, (int)k, nullptr, (int)keep, nullptr, (int)n, tr->buf.get(), (int)keep, e.stream());
        T4A_HIP(hipGetLastError());
        e.sync();
        *q = tq.release();
        *r = tr.release();
    });
}

t4a_gpu_status t4a_gpu_tensor_factorize(const t4a_gpu_tensor* t, const int64_t* left_labels, size_t n_left, int32_t alg,
                                        int32_t canonical, int32_t full_rank, const t4a_gpu_svd_policy* policy,
                                        int32_t has_max_bond_dim, size_t max_bond_dim, int32_t has_qr_rtol, double qr_rtol,
                                        int64_t bond_label, t4a_gpu_tensor** left, t4a_gpu_tensor** right, size_t* rank,
                                        double* singular_values)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(t);
        T4A_REQUIRE_PTR(left);
        T4A_REQUIRE_PTR(right);
        *left = *right = nullptr;
        if (n_left) T4A_REQUIRE_PTR(left_labels);
        if (alg < 0 || alg > 3) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown factorization algorithm");
        if (canonical < 0 || canonical > 1) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown canonical direction");
        const TensorView tv = t->view();
        const std::vector<int64_t> lf(left_labels, left_labels + n_left);
        const UnfoldPlan un = plan_unfold_split(tv, lf);
        const bool trunc = full_rank == 0;
        SvdPolicy pol = convert_policy(policy);
        if (alg == 0 && trunc) {
            if (has_max_bond_dim && max_bond_dim == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_bond_dim must be positive when specified");
            if (!std::isfinite(pol.threshold) || pol.threshold < 0.0)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid SVD truncation threshold: threshold must be finite and non-negative");
        }
        const double tol = has_qr_rtol ? qr_rtol : 1e-15;
        if (alg == 1 && trunc && (!std::isfinite(tol) || tol < 0.0))
            throw Error(T4A_GPU_INVALID_ARGUMENT, "Invalid rtol value: rtol must be finite and non-negative");
        const size_t m = un.m, n = un.n, k = std::min(m, n), count = tv.size();
        if (count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "factorization of an empty tensor");
        if (m > 65535 || n > 65535) throw Error(T4A_GPU_NOT_IMPLEMENTED, "factorize: unfolded dimensions above 65535 are not supported");
        std::vector<int64_t> right_labels;
        for (size_t a = n_left; a < un.perm.size(); ++a) right_labels.push_back(tv.labels[un.perm[a]]);
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        hipStream_t st = e.stream();
        double* d_mat = e.pi(count);
        tensor_permute(e, tv, un.perm, d_mat);
        size_t keep = 0;
        auto finish = [&](const double* d_l, int ldl, const double* d_r, int ldr) {
            // d_l: m x keep (ld ldl), d_r: keep x n (ld ldr) -> handles
            std::vector<size_t> ld = un.left_dims, rd{keep};
            std::vector<int64_t> ll = lf, rl{bond_label};
            ld.push_back(keep);
            ll.push_back(bond_label);
            rd.insert(rd.end(), un.right_dims.begin(), un.right_dims.end());
            rl.insert(rl.end(), right_labels.begin(), right_labels.end());
            auto tl = make_tensor(ld, ll), tr = make_tensor(rd, rl);
            gather_launch(d_l, ldl, nullptr, (int)m, nullptr, (int)keep, tl->buf.get(), (int)m, st);
            gather_launch(d_r, ldr, nullptr, (int)keep, nullptr, (int)n, tr->buf.get(), (int)keep, st);
            T4A_HIP(hipGetLastError());
            e.sync();
            if (rank) *rank = keep;
            *left = tl.release();
            *right = tr.release();
        };
        if (alg == 0) { // SVD
            e.d_tmp.reserve(m * k + k + k * n);
            e.d_tmp2.reserve(std::max(m, n) * k);
            double* d_u = e.d_tmp.get();
            double* d_s = d_u + m * k;
            double* d_vt = d_s + k;
            e.svd(d_mat, (int)m, (int)n, d_u, d_s, d_vt);
            std::vector<double> hs(k);
            download(e, hs.data(), d_s, k);
            keep = k;
            if (trunc) {
                keep = svd_retained_rank(hs.data(), k, pol);
                if (has_max_bond_dim) keep = std::min(keep, max_bond_dim);
            }
            keep = std::min(std::max<size_t>(keep, 1), k);
            if (singular_values) std::copy(hs.begin(), hs.begin() + keep, singular_values);
            if (canonical == 0) { // right = S V^H
                diag_scale_launch(d_vt, (int)k, (int)keep, (int)n, d_s, true, e.d_tmp2.get(), (int)keep, st);
                finish(d_u, (int)m, e.d_tmp2.get(), (int)keep);
            } else { // left = U S
                diag_scale_launch(d_u, (int)m, (int)m, (int)keep, d_s, false, e.d_tmp2.get(), (int)m, st);
                finish(e.d_tmp2.get(), (int)m, d_vt, (int)k);
            }
        } else if (alg == 1) { // QR
            e.d_tmp.reserve(m * k + k * n);
            double* d_q = e.d_tmp.get();
            double* d_r = d_q + m * k;
            e.qr(d_mat, (int)m, (int)n, d_q, d_r);
            keep = k;
            if (trunc) {
                std::vector<double> hr(k * n);
                download(e, hr.data(), d_r, k * n);
                keep = std::min(qr_retained_rank(hr.data(), k, n, tol), k);
            }
            finish(d_q, (int)m, d_r, (int)k);
        } else { // LU / CI on the rrLU kernels
            RrLUOptions lo;
            lo.max_bond_dim = (!trunc || !has_max_bond_dim) ? std::numeric_limits<size_t>::max() : max_bond_dim;
            lo.rel_tol = trunc ? 1e-14 : 0.0;
            lo.abs_tol = 0.0;
            lo.left_orthogonal = canonical == 0;
            if (alg == 2) {
                LuciResult r = e.luci(d_mat, (int)m, (int)n, lo, false, true);
                e.lu_permuted_factors(r, lo.left_orthogonal);
                keep = (size_t)r.rank;
            } else {
                LuciResult r = e.luci(d_mat, (int)m, (int)n, lo, true, false);
                keep = (size_t)r.rank;
            }
            if (keep == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "Failed to create bond index: dimension 0");
            finish(e.left(), (int)m, e.right(), (int)keep);
        }
    });
}

// ---- SimpleTensorTrain arithmetic ----
extern "C++" {
static t4a_gpu_tt* wrap_tt(std::unique_ptr<TensorTrain> r)
{
    r->eng.sync();
    return new t4a_gpu_tt(r->cores, r->eng.stream());
}
} // extern "C++"

t4a_gpu_status t4a_gpu_tt_add(const t4a_gpu_tt* a, const t4a_gpu_tt* b, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        *out = wrap_tt(const_cast<t4a_gpu_tt*>(a)->impl.add(const_cast<t4a_gpu_tt*>(b)->impl, false));
    });
}

t4a_gpu_status t4a_gpu_tt_sub(const t4a_gpu_tt* a, const t4a_gpu_tt* b, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        *out = wrap_tt(const_cast<t4a_gpu_tt*>(a)->impl.add(const_cast<t4a_gpu_tt*>(b)->impl, true));
    });
}

t4a_gpu_status t4a_gpu_tt_scale(t4a_gpu_tt* h, double factor)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl.scale(factor);
    });
}

t4a_gpu_status t4a_gpu_tt_inner_product(const t4a_gpu_tt* a, const t4a_gpu_tt* b, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(a);
        T4A_REQUIRE_PTR(b);
        T4A_REQUIRE_PTR(out);
        *out = const_cast<t4a_gpu_tt*>(a)->impl.inner_product(const_cast<t4a_gpu_tt*>(b)->impl);
    });
}

t4a_gpu_status t4a_gpu_tt_reverse(const t4a_gpu_tt* h, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        *out = wrap_tt(const_cast<t4a_gpu_tt*>(h)->impl.reverse());
    });
}

t4a_gpu_status t4a_gpu_tt_partial_sum(const t4a_gpu_tt* h, const size_t* dims, size_t n_dims, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        if (n_dims) T4A_REQUIRE_PTR(dims);
        *out = wrap_tt(const_cast<t4a_gpu_tt*>(h)->impl.partial_sum(std::vector<size_t>(dims, dims + n_dims)));
    });
}

// ---- simplett_bridge.rs: chain of labelled tensors <-> tensor train ----
t4a_gpu_status t4a_gpu_tt_to_tensors(const t4a_gpu_tt* tt, const int64_t* site_labels, const int64_t* bond_labels,
                                     t4a_gpu_tensor** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(tt);
        const size_t n = tt->impl.len();
        if (n == 0) return;
        T4A_REQUIRE_PTR(site_labels);
        T4A_REQUIRE_PTR(out);
        if (n > 1) T4A_REQUIRE_PTR(bond_labels);
        for (size_t s = 0; s < n; ++s) out[s] = nullptr;
        std::vector<int64_t> all(site_labels, site_labels + n);
        if (n > 1) all.insert(all.end(), bond_labels, bond_labels + n - 1);
        std::sort(all.begin(), all.end());
        if (std::adjacent_find(all.begin(), all.end()) != all.end())
            throw Error(T4A_GPU_INVALID_ARGUMENT, "tensor_train_to_treetn: site and bond indices must be distinct");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        const_cast<t4a_gpu_tt*>(tt)->impl.eng.sync();
        std::vector<std::unique_ptr<t4a_gpu_tensor>> made;
        for (size_t s = 0; s < n; ++s) {
            const DevCore& c = tt->impl.cores[s];
            std::vector<size_t> dims;
            std::vector<int64_t> labels;
            if (s > 0) {
                dims.push_back(c.l);
                labels.push_back(bond_labels[s - 1]);
            }
            dims.push_back(c.s);
            labels.push_back(site_labels[s]);
            if (s + 1 < n) {
                dims.push_back(c.r);
                labels.push_back(bond_labels[s]);
            }
            auto t = make_tensor(dims, labels); // boundary legs have dimension 1: the column-major data is unchanged
            if (c.size())
                T4A_HIP(hipMemcpyAsync(t->buf.get(), c.buf.get(), c.size() * sizeof(double), hipMemcpyDeviceToDevice, e.stream()));
            made.push_back(std::move(t));
        }
        e.sync();
        for (size_t s = 0; s < n; ++s) out[s] = made[s].release();
    });
}

t4a_gpu_status t4a_gpu_tensors_to_tt(const t4a_gpu_tensor* const* tensors, size_t n_sites, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        require_device();
        if (n_sites == 0) {
            *out = new t4a_gpu_tt(std::vector<std::array<size_t, 3>>{}, nullptr);
            return;
        }
        T4A_REQUIRE_PTR(tensors);
        for (size_t s = 0; s < n_sites; ++s) T4A_REQUIRE_PTR(tensors[s]);
        auto shared = [&](size_t a, size_t b) { // the one label two neighbours share
            std::vector<int64_t> common;
            for (int64_t l : tensors[a]->labels)
                if (std::find(tensors[b]->labels.begin(), tensors[b]->labels.end(), l) != tensors[b]->labels.end()) common.push_back(l);
            if (common.size() != 1)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "treetn_to_tensor_train: missing chain edge between nodes " + std::to_string(a) + " and " +
                                                          std::to_string(b));
            return common[0];
        };
        std::vector<int64_t> bonds(n_sites > 0 ? n_sites - 1 : 0);
        for (size_t s = 0; s + 1 < n_sites; ++s) bonds[s] = shared(s, s + 1);
        for (size_t a = 0; a < n_sites; ++a) // a chain: no label may connect nodes that are not neighbours
            for (size_t b = a + 2; b < n_sites; ++b)
                for (int64_t l : tensors[a]->labels)
                    if (std::find(tensors[b]->labels.begin(), tensors[b]->labels.end(), l) != tensors[b]->labels.end())
                        throw Error(T4A_GPU_INVALID_ARGUMENT, "treetn_to_tensor_train: expected a chain with " + std::to_string(n_sites - 1) + " edges");
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        Engine& e = dense_engine();
        std::vector<DevCore> cores(n_sites);
        for (size_t s = 0; s < n_sites; ++s) {
            const t4a_gpu_tensor* t = tensors[s];
            const size_t want = 1 + (s > 0 ? 1 : 0) + (s + 1 < n_sites ? 1 : 0);
            if (t->labels.size() != want)
                throw Error(T4A_GPU_INVALID_ARGUMENT, "treetn_to_tensor_train: node " + std::to_string(s) + " must have exactly one site index");
            std::vector<size_t> perm;
            auto pos = [&](int64_t l) { return (size_t)(std::find(t->labels.begin(), t->labels.end(), l) - t->labels.begin()); };
            size_t site_axis = 0;
            for (size_t a = 0; a < t->labels.size(); ++a)
                if ((s == 0 || t->labels[a] != bonds[s - 1]) && (s + 1 >= n_sites || t->labels[a] != bonds[s])) site_axis = a;
            if (s > 0) perm.push_back(pos(bonds[s - 1]));
            perm.push_back(site_axis);
            if (s + 1 < n_sites) perm.push_back(pos(bonds[s]));
            DevCore& c = cores[s];
            c.l = s > 0 ? t->dims[perm[0]] : 1;
            c.s = t->dims[site_axis];
            c.r = s + 1 < n_sites ? t->dims[perm.back()] : 1;
            if (c.l == 0 || c.s == 0 || c.r == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "treetn_to_tensor_train: a resulting core has a zero dimension");
            c.buf.reserve(c.size());
            tensor_permute(e, t->view(), perm, c.buf.get());
        }
        T4A_HIP(hipGetLastError());
        e.sync();
        *out = new t4a_gpu_tt(cores, e.stream());
    });
}

// ---- tensor4all-aci ----
t4a_gpu_status t4a_gpu_aci_options_default(t4a_gpu_aci_options* o)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(o);
        const AciOptions d;
        o->max_iters = d.max_iters;
        o->min_iters = d.min_iters;
        o->has_max_bond_dim = 0;
        o->max_bond_dim = 0;
        o->tolerance = d.tolerance;
        o->scale_tolerance = 1;
        o->rng_seed = 0;
        o->enable_global_guard = 1;
        o->nsearch_global_pivots = d.nsearch_global_pivots;
        o->max_nglobal_pivot = d.max_nglobal_pivot;
        o->nsweeps_global_search = d.nsweeps_global_search;
        o->tol_margin_global_search = d.tol_margin_global_search;
    });
}

} // extern "C"

struct t4a_gpu_aci_problem {
    std::unique_ptr<t4a::AciProblem> impl;
};

namespace {
AciOptions convert_aci(const t4a_gpu_aci_options* o)
{
    AciOptions a;
    if (o) {
        a.max_iters = o->max_iters;
        a.min_iters = o->min_iters;
        a.has_max_bond_dim = o->has_max_bond_dim != 0;
        a.max_bond_dim = o->max_bond_dim;
        a.tolerance = o->tolerance;
        a.scale_tolerance = o->scale_tolerance != 0;
        a.rng_seed = o->rng_seed;
        a.enable_global_guard = o->enable_global_guard != 0;
        a.nsearch_global_pivots = o->nsearch_global_pivots;
        a.max_nglobal_pivot = o->max_nglobal_pivot;
        a.nsweeps_global_search = o->nsweeps_global_search;
        a.tol_margin_global_search = o->tol_margin_global_search;
    }
    return a;
}
AciHostOp make_aci_op(int32_t kind, t4a_gpu_aci_op_fn op, void* user)
{
    if (kind < 0 || kind > 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown ACI operator kind");
    if (kind != 0) return {};
    if (!op) throw Error(T4A_GPU_NULL_POINTER, "operator callback is null");
    return [op, user](const double* v, size_t K, size_t np, double* out) {
        if (op(user, v, K, np, out) != 0) throw Error(T4A_GPU_CALLBACK_ERROR, "operator callback reported an error");
    };
}
std::vector<TensorTrain*> aci_inputs(const t4a_gpu_tt* const* inputs, size_t n)
{
    if (n == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "inputs must not be empty");
    if (!inputs) throw Error(T4A_GPU_NULL_POINTER, "inputs is null");
    std::vector<TensorTrain*> v(n);
    for (size_t k = 0; k < n; ++k) {
        if (!inputs[k]) throw Error(T4A_GPU_NULL_POINTER, "input tensor train is null");
        v[k] = const_cast<TensorTrain*>(&inputs[k]->impl);
    }
    return v;
}
} // namespace

extern "C" {

t4a_gpu_status t4a_gpu_aci_elementwise(const t4a_gpu_tt* const* inputs, size_t n_inputs, int32_t op_kind, t4a_gpu_aci_op_fn op,
                                       void* user, const t4a_gpu_aci_options* options, const t4a_gpu_tt* initial_guess,
                                       t4a_gpu_tt** result, size_t* n_iters, size_t* ranks, double* errors,
                                       size_t* nglobal_pivots, int32_t* termination)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(result);
        *result = nullptr;
        const AciOptions o = convert_aci(options);
        o.validate();
        std::vector<TensorTrain*> in = aci_inputs(inputs, n_inputs);
        AciHostOp hop = make_aci_op(op_kind, op, user);
        require_device();
        if (n_iters) *n_iters = 0;
        if (in[0] && in[0]->len() == 1) {
            std::unique_ptr<TensorTrain> tt = aci_one_site(in, (AciOpKind)op_kind, hop);
            *result = new t4a_gpu_tt(tt->cores, tt->eng.stream());
            if (termination) *termination = 0;
            return;
        }
        AciProblem p(in, initial_guess ? &initial_guess->impl : nullptr, o, (AciOpKind)op_kind, hop);
        p.run();
        std::unique_ptr<TensorTrain> tt = p.solution_tt();
        tt->eng.sync();
        *result = new t4a_gpu_tt(tt->cores, tt->eng.stream());
        if (n_iters) *n_iters = p.ranks.size();
        for (size_t i = 0; i < p.ranks.size(); ++i) {
            if (ranks) ranks[i] = p.ranks[i];
            if (errors) errors[i] = p.errors[i];
            if (nglobal_pivots) nglobal_pivots[i] = p.nglobal_pivots[i];
        }
        if (termination) *termination = (int32_t)p.termination;
    });
}

t4a_gpu_status t4a_gpu_aci_problem_new(const t4a_gpu_tt* const* inputs, size_t n_inputs, int32_t op_kind, t4a_gpu_aci_op_fn op,
                                       void* user, const t4a_gpu_aci_options* options, const t4a_gpu_tt* initial_guess,
                                       t4a_gpu_aci_problem** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(out);
        *out = nullptr;
        const AciOptions o = convert_aci(options);
        o.validate();
        std::vector<TensorTrain*> in = aci_inputs(inputs, n_inputs);
        AciHostOp hop = make_aci_op(op_kind, op, user);
        require_device();
        auto h = std::make_unique<t4a_gpu_aci_problem>();
        h->impl = std::make_unique<AciProblem>(in, initial_guess ? &initial_guess->impl : nullptr, o, (AciOpKind)op_kind, hop);
        *out = h.release();
    });
}

void t4a_gpu_aci_problem_release(t4a_gpu_aci_problem* h) { delete h; }

t4a_gpu_status t4a_gpu_treeaci_local_update_f64(size_t n_inputs, const size_t* bond_dims, const double* const* row_frames,
                                                const double* const* col_frames, size_t row_count, size_t col_count, int32_t op_kind,
                                                t4a_gpu_aci_op_fn op, void* user, size_t max_bond_dim, double tolerance,
                                                int32_t scale_tolerance, int32_t left_orthogonal, size_t* rank, size_t* row_indices,
                                                size_t* col_indices, double* pivot_errors, size_t* n_pivot_errors, double* left,
                                                double* right, double* sampled_scale, double* local_values)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(bond_dims);
        T4A_REQUIRE_PTR(row_frames);
        T4A_REQUIRE_PTR(col_frames);
        T4A_REQUIRE_PTR(rank);
        T4A_REQUIRE_PTR(row_indices);
        T4A_REQUIRE_PTR(col_indices);
        T4A_REQUIRE_PTR(pivot_errors);
        T4A_REQUIRE_PTR(n_pivot_errors);
        T4A_REQUIRE_PTR(left);
        T4A_REQUIRE_PTR(right);
        T4A_REQUIRE_PTR(sampled_scale);
        if (op_kind < 0 || op_kind > 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown operator kind");
        checked_mul(row_count, col_count, "local matrix elements");
        require_int_dims({row_count, col_count}, "local matrix shape");
        std::vector<size_t> bd(bond_dims, bond_dims + n_inputs);
        std::vector<const double*> rf(n_inputs), cf(n_inputs);
        for (size_t k = 0; k < n_inputs; ++k) {
            if (bd[k] && (!row_frames[k] || !col_frames[k])) throw Error(T4A_GPU_NULL_POINTER, "frame block is null");
            rf[k] = row_frames[k];
            cf[k] = col_frames[k];
        }
        AciHostOp host;
        if (op_kind == T4A_GPU_ACI_OP_CALLBACK) {
            T4A_REQUIRE_PTR(op);
            host = [op, user](const double* values, size_t ni, size_t npts, double* out) {
                if (op(user, values, ni, npts, out) != 0) throw Error(T4A_GPU_CALLBACK_ERROR, "elementwise operator callback failed");
            };
        }
        std::lock_guard<std::mutex> lock(g_dense_mutex);
        TreeAciLocalResult r = treeaci_local_update(dense_engine(), bd, rf, cf, row_count, col_count, (AciOpKind)op_kind, host, max_bond_dim,
                                                    tolerance, scale_tolerance != 0, left_orthogonal != 0);
        *rank = r.rank;
        for (size_t i = 0; i < r.rank; ++i) {
            row_indices[i] = r.row_indices[i];
            col_indices[i] = r.col_indices[i];
        }
        *n_pivot_errors = r.pivot_errors.size();
        for (size_t i = 0; i < r.pivot_errors.size(); ++i) pivot_errors[i] = r.pivot_errors[i];
        std::copy(r.left.begin(), r.left.end(), left);
        std::copy(r.right.begin(), r.right.end(), right);
        *sampled_scale = r.sampled_scale;
        if (local_values) std::copy(r.local_values.begin(), r.local_values.end(), local_values);
    });
}

t4a_gpu_status t4a_gpu_aci_problem_local_update(t4a_gpu_aci_problem* h, size_t bond, int32_t left_orthogonal)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        h->impl->local_update(bond, left_orthogonal != 0);
    });
}

t4a_gpu_status t4a_gpu_aci_problem_add_global_pivots(t4a_gpu_aci_problem* h, const size_t* pivots, size_t n_pivots, size_t* added)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (n_pivots) T4A_REQUIRE_PTR(pivots);
        const size_t n = h->impl->len();
        std::vector<std::vector<uint32_t>> pv(n_pivots, std::vector<uint32_t>(n));
        for (size_t p = 0; p < n_pivots; ++p)
            for (size_t s = 0; s < n; ++s) {
                if (pivots[s + n * p] > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot index out of bounds");
                pv[p][s] = (uint32_t)pivots[s + n * p];
            }
        const size_t a = h->impl->add_global_pivots(pv);
        if (added) *added = a;
    });
}

t4a_gpu_status t4a_gpu_aci_problem_find_global_pivots(t4a_gpu_aci_problem* h, uint64_t seed, size_t* count, size_t* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(count);
        const auto pv = h->impl->find_global_pivots(seed);
        *count = pv.size();
        const size_t n = h->impl->len();
        if (out)
            for (size_t p = 0; p < pv.size(); ++p)
                for (size_t s = 0; s < n; ++s) out[s + n * p] = pv[p][s];
    });
}

t4a_gpu_status t4a_gpu_aci_problem_solution(t4a_gpu_aci_problem* h, t4a_gpu_tt** out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(out);
        std::unique_ptr<TensorTrain> tt = h->impl->solution_tt();
        tt->eng.sync();
        *out = new t4a_gpu_tt(tt->cores, tt->eng.stream());
    });
}

t4a_gpu_status t4a_gpu_aci_problem_frame(t4a_gpu_aci_problem* h, int32_t right, size_t input, size_t site, size_t* rows,
                                         size_t* cols, double* out)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        T4A_REQUIRE_PTR(rows);
        T4A_REQUIRE_PTR(cols);
        const std::vector<double> f = h->impl->frame_host(right != 0, input, site, rows, cols);
        if (out) std::copy(f.begin(), f.end(), out);
    });
}

t4a_gpu_status t4a_gpu_aci_problem_errors(const t4a_gpu_aci_problem* h, double* pivot_errors, double* pivot_scales)
{
    return guarded([&] {
        T4A_REQUIRE_PTR(h);
        if (pivot_errors) std::copy(h->impl->pivot_errors.begin(), h->impl->pivot_errors.end(), pivot_errors);
        if (pivot_scales) std::copy(h->impl->pivot_scales.begin(), h->impl->pivot_scales.end(), pivot_scales);
    });
}

} // extern "C"
