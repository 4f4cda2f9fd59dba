// tensorops.hip — see tensorops.hpp.
#include "tensorops.hpp"

#include <algorithm>
#include <cmath>

namespace t4a {

size_t svd_retained_rank(const double* s, size_t n, const SvdPolicy& p)
{
    if (n == 0) return 1;
    std::vector<double> m(n);
    bool all_zero = true;
    for (size_t k = 0; k < n; ++k) {
        m[k] = p.measure == 0 ? s[k] : s[k] * s[k];
        if (m[k] != 0.0) all_zero = false;
    }
    if (all_zero) return 1;
    size_t keep = 0;
    if (p.rule == 0) {
        if (p.scale == 0) {
            double ref = 0.0;
            for (double v : m) ref = std::max(ref, v);
            while (keep < n && ref > 0.0 && m[keep] / ref > p.threshold) ++keep;
        } else {
            while (keep < n && m[keep] > p.threshold) ++keep;
        }
    } else {
        double total = 0.0;
        for (double v : m) total += v;
        if (p.scale == 0 && total == 0.0) return 1;
        double discarded = 0.0;
        keep = n;
        for (size_t i = n; i-- > 0;) {
            const bool ok = p.scale == 0 ? (discarded + m[i]) / total <= p.threshold : discarded + m[i] <= p.threshold;
            if (!ok) break;
            discarded += m[i];
            keep = i;
        }
    }
    return std::max<size_t>(keep, 1);
}

size_t qr_retained_rank(const double* r, size_t k, size_t n, double rtol)
{
    if (k == 0 || n == 0) return 1;
    const size_t md = std::min(k, n);
    std::vector<double> norms(md);
    double mx = 0.0;
    for (size_t i = 0; i < md; ++i) {
        double sq = 0.0;
        for (size_t j = i; j < n; ++j) {
            const double v = std::fabs(r[i + j * k]);
            sq += v * v;
        }
        norms[i] = std::sqrt(sq);
        mx = std::max(mx, norms[i]);
    }
    if (mx == 0.0) return 1;
    const double thr = rtol * mx;
    size_t cnt = 0;
    for (double v : norms)
        if (v >= thr) ++cnt;
    return std::max<size_t>(cnt, 1);
}

namespace {

struct PermuteArgs {
    int rank;
    unsigned long long total;
    unsigned long long out_dims[TENSOR_MAX_RANK];
    unsigned long long src_stride[TENSOR_MAX_RANK]; // stride in the input of output axis k
};

__global__ void __launch_bounds__(256) permute_kernel(const double* __restrict__ in, double* __restrict__ out, PermuteArgs a)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.total; e += stride) {
        unsigned long long rem = e, src = 0;
        for (int k = 0; k < a.rank; ++k) {
            const unsigned long long d = a.out_dims[k];
            const unsigned long long q = rem / d;
            src += (rem - q * d) * a.src_stride[k];
            rem = q;
        }
        out[e] = in[src];
    }
}

__global__ void __launch_bounds__(256) diag_scale_kernel(const double* __restrict__ in, int ldi, int rows, int cols,
                                                         const double* __restrict__ s, int by_row, double* __restrict__ out, int ldo)
{
    const size_t total = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (size_t)rows), j = (int)(e / (size_t)rows);
        out[i + (size_t)ldo * j] = in[i + (size_t)ldi * j] * (by_row ? s[i] : s[j]);
    }
}

void validate(const TensorView& t, const char* who)
{
    if (t.dims.size() != t.labels.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, std::string(who) + ": dims / labels length mismatch");
    if (t.dims.size() > (size_t)TENSOR_MAX_RANK)
        throw Error(T4A_GPU_NOT_IMPLEMENTED, std::string(who) + ": tensors of rank above " + std::to_string(TENSOR_MAX_RANK) + " are not supported");
    for (size_t a = 0; a < t.labels.size(); ++a)
        for (size_t b = a + 1; b < t.labels.size(); ++b)
            if (t.labels[a] == t.labels[b]) throw Error(T4A_GPU_INVALID_ARGUMENT, std::string(who) + ": duplicate index in tensor");
}

} // namespace

void tensor_permute(Engine& e, const TensorView& t, const std::vector<size_t>& perm, double* d_out)
{
    const size_t r = t.dims.size();
    const size_t total = t.size();
    if (total == 0) return;
    bool identity = true;
    for (size_t k = 0; k < r; ++k) identity = identity && perm[k] == k;
    if (identity) {
        T4A_HIP(hipMemcpyAsync(d_out, t.d_data, total * sizeof(double), hipMemcpyDeviceToDevice, e.stream()));
        return;
    }
    std::vector<unsigned long long> in_stride(r, 1);
    for (size_t k = 1; k < r; ++k) in_stride[k] = in_stride[k - 1] * t.dims[k - 1];
    PermuteArgs a{};
    a.rank = (int)r;
    a.total = total;
    for (size_t k = 0; k < r; ++k) {
        a.out_dims[k] = t.dims[perm[k]];
        a.src_stride[k] = in_stride[perm[k]];
    }
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(permute_kernel, dim3(blocks), dim3(256), 0, e.stream(), t.d_data, d_out, a);
    T4A_HIP(hipGetLastError());
}

ContractPlan plan_contract_pair(const TensorView& a, const TensorView& b) // index_ops.rs:660-696
{
    validate(a, "contract_pair lhs");
    validate(b, "contract_pair rhs");
    std::vector<size_t> axes_a, axes_b;
    for (size_t i = 0; i < a.labels.size(); ++i)
        for (size_t j = 0; j < b.labels.size(); ++j)
            if (a.labels[i] == b.labels[j]) {
                if (a.dims[i] != b.dims[j])
                    throw Error(T4A_GPU_INVALID_ARGUMENT, "contraction dimension mismatch: lhs axis " + std::to_string(i) + " has " +
                                                              std::to_string(a.dims[i]) + ", rhs axis " + std::to_string(j) + " has " +
                                                              std::to_string(b.dims[j]));
                axes_a.push_back(i);
                axes_b.push_back(j);
            }
    ContractPlan p;
    for (size_t i = 0; i < a.labels.size(); ++i)
        if (std::find(axes_a.begin(), axes_a.end(), i) == axes_a.end()) {
            p.perm_a.push_back(i);
            p.M *= a.dims[i];
            p.out_dims.push_back(a.dims[i]);
            p.out_labels.push_back(a.labels[i]);
        }
    for (size_t i : axes_a) {
        p.perm_a.push_back(i);
        p.K *= a.dims[i];
    }
    for (size_t j : axes_b) p.perm_b.push_back(j);
    for (size_t j = 0; j < b.labels.size(); ++j)
        if (std::find(axes_b.begin(), axes_b.end(), j) == axes_b.end()) {
            p.perm_b.push_back(j);
            p.N *= b.dims[j];
            p.out_dims.push_back(b.dims[j]);
            p.out_labels.push_back(b.labels[j]);
        }
    if (p.out_dims.size() > (size_t)TENSOR_MAX_RANK) throw Error(T4A_GPU_NOT_IMPLEMENTED, "contract_pair: result rank too large");
    if (p.M > 0x7FFFFFFFull || p.N > 0x7FFFFFFFull || p.K > 0x7FFFFFFFull)
        throw Error(T4A_GPU_NOT_IMPLEMENTED, "contract_pair: fused dimension above 2^31");
    return p;
}

void tensor_contract_pair(Engine& e, const TensorView& a, const TensorView& b, const ContractPlan& p, double* d_out)
{
    if (p.M * p.N == 0) return;
    if (p.K == 0) {
        fill_launch(d_out, p.M * p.N, 0.0, e.stream());
        return;
    }
    e.d_tmp.reserve(std::max<size_t>(a.size(), 1));
    e.d_tmp2.reserve(std::max<size_t>(b.size(), 1));
    tensor_permute(e, a, p.perm_a, e.d_tmp.get());
    tensor_permute(e, b, p.perm_b, e.d_tmp2.get());
    GemmDesc g{};
    g.m = (int)p.M;
    g.n = (int)p.N;
    g.k = (int)p.K;
    g.A = e.d_tmp.get();
    g.lda = (int)p.M;
    g.strideA = 0;
    g.transA = 0;
    g.B = e.d_tmp2.get();
    g.ldb = (int)p.K;
    g.strideB = 0;
    g.transB = 0;
    g.C = d_out;
    g.ldc = (int)p.M;
    g.strideC = 0;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.batch = 1;
    gemm_launch(g, e.stream());
    T4A_HIP(hipGetLastError());
}

UnfoldPlan plan_unfold_split(const TensorView& t, const std::vector<int64_t>& left) // idx_tensor.rs:5278-5345
{
    validate(t, "unfold_split");
    const size_t rank = t.dims.size();
    if (!(rank >= 2)) throw Error(T4A_GPU_INVALID_ARGUMENT, "Tensor must have rank >= 2, got rank " + std::to_string(rank));
    if (!(left.size() > 0 && left.size() < rank))
        throw Error(T4A_GPU_INVALID_ARGUMENT, "Left indices must be a non-empty proper subset of tensor indices (0 < left_len < rank), got left_len=" +
                                                  std::to_string(left.size()) + ", rank=" + std::to_string(rank));
    UnfoldPlan p;
    for (size_t a = 0; a < left.size(); ++a) {
        auto it = std::find(t.labels.begin(), t.labels.end(), left[a]);
        if (it == t.labels.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "Index in left_inds not found in tensor");
        for (size_t b = 0; b < a; ++b)
            if (left[a] == left[b]) throw Error(T4A_GPU_INVALID_ARGUMENT, "Duplicate index in left_inds");
        p.perm.push_back((size_t)(it - t.labels.begin()));
    }
    const size_t nl = left.size();
    for (size_t k = 0; k < rank; ++k)
        if (std::find(p.perm.begin(), p.perm.begin() + nl, k) == p.perm.begin() + nl) p.perm.push_back(k);
    for (size_t k = 0; k < rank; ++k) {
        const size_t d = t.dims[p.perm[k]];
        if (k < nl) {
            p.left_dims.push_back(d);
            p.m *= d;
        } else {
            p.right_dims.push_back(d);
            p.n *= d;
        }
    }
    return p;
}

void diag_scale_launch(const double* in, int ldi, int rows, int cols, const double* sv, bool by_row, double* out, int ldo,
                       hipStream_t stream)
{
    const size_t total = (size_t)rows * cols;
    if (total == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(diag_scale_kernel, dim3(blocks), dim3(256), 0, stream, in, ldi, rows, cols, sv, by_row ? 1 : 0, out, ldo);
}

} // namespace t4a
