// tensorops.hpp — dense part of the dynamic-index tensor layer of tensor4all-core on the gfx950 engine:
// unfold_split (defaults/idx_tensor.rs:5278-5345), contract_pair for dense operands (defaults/contract.rs:334-343 with
// prepare_contraction, index_ops.rs:660-696), svd_with (defaults/svd.rs:255-395) and qr_with (defaults/qr.rs:206-328)
// with the reference's rank rules.  Indices are integer labels; prime levels, tags, structured storage and AD stay with
// the caller.  Permutations run as one gather kernel, contractions on the f64-MFMA GEMM, factorisations on the
// Jacobi SVD / Householder QR of kernels_linalg.hip.
#pragma once

#include "engine.hpp"

namespace t4a {

constexpr int TENSOR_MAX_RANK = 16;

struct SvdPolicy { // truncation.rs:137-147; default relative / per value / 1e-12 (svd.rs:80-87)
    double threshold = 1e-12;
    int scale = 0;   // 0 Relative, 1 Absolute
    int measure = 0; // 0 Value, 1 SquaredValue
    int rule = 0;    // 0 PerValue, 1 DiscardedTailSum
};

// host-side rank rules (no device needed)
size_t svd_retained_rank(const double* s, size_t n, const SvdPolicy& policy);   // svd.rs:150-211
size_t qr_retained_rank(const double* r, size_t k, size_t n, double rtol);      // qr.rs:74-117, r is k x n column-major

struct TensorView {
    const double* d_data; // device, column-major
    std::vector<size_t> dims;
    std::vector<int64_t> labels;
    size_t size() const
    {
        size_t n = 1;
        for (size_t d : dims) n *= d;
        return n;
    }
};

// out index k takes input index perm[k]; d_out must hold t.size() doubles
void tensor_permute(Engine& e, const TensorView& t, const std::vector<size_t>& perm, double* d_out);

struct ContractPlan {
    std::vector<size_t> perm_a, perm_b;
    size_t M = 1, K = 1, N = 1;
    std::vector<size_t> out_dims;
    std::vector<int64_t> out_labels;
};
ContractPlan plan_contract_pair(const TensorView& a, const TensorView& b);
// d_out: M x N doubles = the result tensor [free a.., free b..]; uses e.d_tmp / e.d_tmp2 as permutation scratch
void tensor_contract_pair(Engine& e, const TensorView& a, const TensorView& b, const ContractPlan& plan, double* d_out);

// N-ary contraction of a connected tensor network (defaults/contract.rs:283-298 contract / contract_with_options, plan :885-941,
// connectivity :1167-1230): every label that occurs in more than one operand is summed unless it is retained; the result carries the
// labels that occur once, or are retained, in order of first appearance (operands in order, axes in order).  Errors like the
// reference's: no operands, a retained label that no operand has, operands that fall into several connected components (a retained
// label connects its holders), a label with two different dimensions.  The network is reduced pair by pair — at every step the
// connected pair with the smallest result — each step one (batched) GEMM on the f64 matrix cores: a shared label that another
// operand or the result still needs stays as a batch axis of that step (the reference hands the whole network to tenferro's einsum;
// summation order is the backend's there as here: values agree to rounding).
struct OwnedTensor {
    DevBuf<double> buf;
    std::vector<size_t> dims;
    std::vector<int64_t> labels;
};
struct NetworkPlan {
    std::vector<int64_t> out_labels;
    std::vector<size_t> out_dims;
};
NetworkPlan plan_contract_network(const std::vector<TensorView>& ts, const std::vector<int64_t>& retain); // validation + result indices (host only)
OwnedTensor tensor_contract_network(Engine& e, const std::vector<TensorView>& ts, const std::vector<int64_t>& retain);

struct UnfoldPlan {
    std::vector<size_t> perm;
    std::vector<size_t> left_dims, right_dims;
    size_t m = 1, n = 1;
};
UnfoldPlan plan_unfold_split(const TensorView& t, const std::vector<int64_t>& left);

// out[i + ldo*j] = in[i + ldi*j] * (by_row ? s[i] : s[j])   (S absorbed into a factor, factorize.rs:519-557)
void diag_scale_launch(const double* in, int ldi, int rows, int cols, const double* s, bool by_row, double* out, int ldo,
                       hipStream_t stream);

} // namespace t4a
