// tensorops.hip — see tensorops.hpp.
#include "tensorops.hpp"

#include <memory>

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

NetworkPlan plan_contract_network(const std::vector<TensorView>& ts, const std::vector<int64_t>& retain) // contract.rs:530-572, :885-941
{
    if (ts.empty()) throw Error(T4A_GPU_INVALID_ARGUMENT, "No tensors to contract");
    for (const TensorView& t : ts) validate(t, "contract operand");
    auto has = [](const TensorView& t, int64_t l) { return std::find(t.labels.begin(), t.labels.end(), l) != t.labels.end(); };
    for (int64_t r : retain) { // validate_retained_indices_exist :943-959
        bool found = false;
        for (const TensorView& t : ts) found = found || has(t, r);
        if (!found) throw Error(T4A_GPU_INVALID_ARGUMENT, "Retained index " + std::to_string(r) + " does not appear in the input tensors");
    }
    NetworkPlan p;
    if (ts.size() == 1) { // (:539-541: a single operand is returned as it is)
        p.out_labels = ts[0].labels;
        p.out_dims = ts[0].dims;
        return p;
    }
    // connected components: operands that share a label (contractable or retained) are joined (:1167-1230)
    const size_t n = ts.size();
    std::vector<size_t> parent(n);
    for (size_t i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](size_t x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
            for (int64_t l : ts[i].labels)
                if (has(ts[j], l)) parent[find(i)] = find(j);
    size_t components = 0;
    for (size_t i = 0; i < n; ++i) components += find(i) == i ? 1 : 0;
    if (components > 1)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "Disconnected tensor network: " + std::to_string(components) + " components found");
    // sizes and occurrence counts per label (:728-750, :892-897)
    std::vector<int64_t> seen;
    std::vector<size_t> seen_dim, count;
    for (const TensorView& t : ts)
        for (size_t a = 0; a < t.labels.size(); ++a) {
            const auto it = std::find(seen.begin(), seen.end(), t.labels[a]);
            if (it == seen.end()) {
                seen.push_back(t.labels[a]);
                seen_dim.push_back(t.dims[a]);
                count.push_back(1);
            } else {
                const size_t k = (size_t)(it - seen.begin());
                if (seen_dim[k] != t.dims[a])
                    throw Error(T4A_GPU_INVALID_ARGUMENT, "Internal label shape mismatch: label " + std::to_string(t.labels[a]) + " has dimensions " +
                                                              std::to_string(seen_dim[k]) + " and " + std::to_string(t.dims[a]));
                ++count[k];
            }
        }
    for (size_t k = 0; k < seen.size(); ++k) { // first appearance order is the order of `seen`
        const bool retained = std::find(retain.begin(), retain.end(), seen[k]) != retain.end();
        if (count[k] == 1 || retained) {
            p.out_labels.push_back(seen[k]);
            p.out_dims.push_back(seen_dim[k]);
        }
    }
    if (p.out_labels.size() > (size_t)TENSOR_MAX_RANK) throw Error(T4A_GPU_NOT_IMPLEMENTED, "contract: result rank too large");
    return p;
}

OwnedTensor tensor_contract_network(Engine& e, const std::vector<TensorView>& ts, const std::vector<int64_t>& retain)
{
    const NetworkPlan plan = plan_contract_network(ts, retain);
    hipStream_t st = e.stream();
    struct Node {
        TensorView v;
        std::shared_ptr<DevBuf<double>> own; // null: an input operand
    };
    std::vector<Node> work;
    for (const TensorView& t : ts) work.push_back(Node{t, nullptr});
    auto has = [](const TensorView& t, int64_t l) { return std::find(t.labels.begin(), t.labels.end(), l) != t.labels.end(); };
    auto in_out = [&](int64_t l) { return std::find(plan.out_labels.begin(), plan.out_labels.end(), l) != plan.out_labels.end(); };
    while (work.size() > 1) {
        // the connected pair with the smallest result (ties: the first such pair)
        size_t bi = 0, bj = 0;
        double best = -1.0;
        for (size_t i = 0; i < work.size(); ++i)
            for (size_t j = i + 1; j < work.size(); ++j) {
                bool shares = false;
                for (int64_t l : work[i].v.labels) shares = shares || has(work[j].v, l);
                if (!shares) continue;
                double sz = 1.0;
                auto needed = [&](int64_t l) {
                    if (in_out(l)) return true;
                    for (size_t q = 0; q < work.size(); ++q)
                        if (q != i && q != j && has(work[q].v, l)) return true;
                    return false;
                };
                for (size_t a = 0; a < work[i].v.labels.size(); ++a) {
                    const int64_t l = work[i].v.labels[a];
                    if (!has(work[j].v, l) || needed(l)) sz *= (double)work[i].v.dims[a];
                }
                for (size_t a = 0; a < work[j].v.labels.size(); ++a)
                    if (!has(work[i].v, work[j].v.labels[a])) sz *= (double)work[j].v.dims[a];
                if (best < 0.0 || sz < best) {
                    best = sz;
                    bi = i;
                    bj = j;
                }
            }
        if (best < 0.0) throw Error(T4A_GPU_INTERNAL_ERROR, "contract: no connected pair left in a connected network");
        const TensorView& A = work[bi].v;
        const TensorView& B = work[bj].v;
        auto needed = [&](int64_t l) {
            if (in_out(l)) return true;
            for (size_t q = 0; q < work.size(); ++q)
                if (q != bi && q != bj && has(work[q].v, l)) return true;
            return false;
        };
        // A -> [free A, summed, batch], B -> [summed, free B, batch]; result [free A, free B, batch]
        std::vector<size_t> pa_free, pa_sum, pa_batch, pb_sum, pb_free, pb_batch;
        size_t M = 1, K = 1, N = 1, Bt = 1;
        for (size_t a = 0; a < A.labels.size(); ++a) {
            const int64_t l = A.labels[a];
            if (!has(B, l)) {
                pa_free.push_back(a);
                M *= A.dims[a];
            } else if (needed(l)) {
                pa_batch.push_back(a);
                Bt *= A.dims[a];
            } else {
                pa_sum.push_back(a);
                K *= A.dims[a];
            }
        }
        auto pos_in_b = [&](int64_t l) { return (size_t)(std::find(B.labels.begin(), B.labels.end(), l) - B.labels.begin()); };
        for (size_t a : pa_sum) pb_sum.push_back(pos_in_b(A.labels[a]));
        for (size_t a : pa_batch) pb_batch.push_back(pos_in_b(A.labels[a]));
        for (size_t b = 0; b < B.labels.size(); ++b)
            if (!has(A, B.labels[b])) {
                pb_free.push_back(b);
                N *= B.dims[b];
            }
        if (M > 0x7FFFFFFFull || N > 0x7FFFFFFFull || K > 0x7FFFFFFFull || Bt > 0x7FFFFFFFull)
            throw Error(T4A_GPU_NOT_IMPLEMENTED, "contract: fused dimension above 2^31");
        std::vector<size_t> perm_a, perm_b;
        perm_a.insert(perm_a.end(), pa_free.begin(), pa_free.end());
        perm_a.insert(perm_a.end(), pa_sum.begin(), pa_sum.end());
        perm_a.insert(perm_a.end(), pa_batch.begin(), pa_batch.end());
        perm_b.insert(perm_b.end(), pb_sum.begin(), pb_sum.end());
        perm_b.insert(perm_b.end(), pb_free.begin(), pb_free.end());
        perm_b.insert(perm_b.end(), pb_batch.begin(), pb_batch.end());
        Node res;
        for (size_t a : pa_free) {
            res.v.dims.push_back(A.dims[a]);
            res.v.labels.push_back(A.labels[a]);
        }
        for (size_t b : pb_free) {
            res.v.dims.push_back(B.dims[b]);
            res.v.labels.push_back(B.labels[b]);
        }
        for (size_t a : pa_batch) {
            res.v.dims.push_back(A.dims[a]);
            res.v.labels.push_back(A.labels[a]);
        }
        if (res.v.dims.size() > (size_t)TENSOR_MAX_RANK) throw Error(T4A_GPU_NOT_IMPLEMENTED, "contract: intermediate rank too large");
        res.own = std::make_shared<DevBuf<double>>();
        res.own->reserve(std::max<size_t>(M * N * Bt, 1));
        res.v.d_data = res.own->get();
        if (M * N * Bt > 0) {
            e.d_tmp.reserve(std::max<size_t>(A.size(), 1));
            e.d_tmp2.reserve(std::max<size_t>(B.size(), 1));
            tensor_permute(e, A, perm_a, e.d_tmp.get());
            tensor_permute(e, B, perm_b, e.d_tmp2.get());
            GemmDesc g{};
            g.m = (int)M;
            g.n = (int)N;
            g.k = (int)K;
            g.A = e.d_tmp.get();
            g.lda = (int)M;
            g.strideA = (long long)(M * K);
            g.transA = 0;
            g.B = e.d_tmp2.get();
            g.ldb = (int)K;
            g.strideB = (long long)(K * N);
            g.transB = 0;
            g.C = res.own->get();
            g.ldc = (int)M;
            g.strideC = (long long)(M * N);
            g.alpha = 1.0;
            g.beta = 0.0;
            g.batch = (int)Bt;
            gemm_launch(g, st);
            T4A_HIP(hipGetLastError());
            e.sync(); // (the permutation scratch is reused by the next step)
        }
        work.erase(work.begin() + (long)bj);
        work.erase(work.begin() + (long)bi);
        work.push_back(std::move(res));
    }
    // into the reference's index order
    const TensorView& F = work[0].v;
    OwnedTensor out;
    out.dims = plan.out_dims;
    out.labels = plan.out_labels;
    std::vector<size_t> perm;
    for (int64_t l : plan.out_labels) {
        const auto it = std::find(F.labels.begin(), F.labels.end(), l);
        if (it == F.labels.end()) throw Error(T4A_GPU_INTERNAL_ERROR, "contract: a result index was lost");
        perm.push_back((size_t)(it - F.labels.begin()));
    }
    if (perm.size() != F.labels.size()) throw Error(T4A_GPU_INTERNAL_ERROR, "contract: an index was left uncontracted");
    out.buf.reserve(std::max<size_t>(F.size(), 1));
    if (F.dims.empty()) T4A_HIP(hipMemcpyAsync(out.buf.get(), F.d_data, sizeof(double), hipMemcpyDeviceToDevice, st)); // (a scalar)
    else if (F.size() > 0) tensor_permute(e, F, perm, out.buf.get());
    e.sync();
    return out;
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
