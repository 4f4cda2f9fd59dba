// tt.hpp — device-resident mirror of tensor4all-simplett's SimpleTensorTrain<f64>
// (crates/tensor4all-simplett/src/tensortrain.rs:97, traits.rs:146-355, compression.rs:375-507,
//  cache.rs:558-744).  Site tensors live in HBM as column-major (left, site, right) blocks.
#pragma once

#include <array>
#include <memory>
#include <vector>

#include "engine.hpp"

namespace t4a {

struct DevCore {
    DevBuf<double> buf;
    size_t l = 0, s = 0, r = 0;
    size_t size() const { return l * s * r; }
};

enum class CompressionMethod : int { LU = 0, CI = 1, SVD = 2 }; // compression.rs:40-52

struct CompressionOptions { // compression.rs:75-125
    CompressionMethod method = CompressionMethod::LU;
    double tolerance = 1e-12;
    size_t max_bond_dim = 0; // 0 == None
    bool normalize_error = true;
};

class TensorTrain {
public:
    // dims3: (l, s, r) per site; host_data: the cores concatenated, each column-major.
    TensorTrain(const std::vector<std::array<size_t, 3>>& dims3, const double* host_data);
    // adopt device cores (copied device-to-device on this object's stream)
    TensorTrain(const std::vector<DevCore>& cores, hipStream_t src_stream);

    size_t len() const { return cores.size(); }
    std::vector<size_t> link_dims() const;
    std::vector<size_t> site_dims() const;
    size_t rank() const;
    std::vector<double> site_tensor_host(size_t site);

    std::vector<double> evaluate(const uint32_t* idx, size_t n_pts); // idx n_sites x n_pts col-major
    double sum();
    double norm2();
    void compress(const CompressionOptions& options);
    // arithmetic.rs:34-180, tensortrain.rs:264-345, :449-583 — results are new device-resident trains
    std::unique_ptr<TensorTrain> add(TensorTrain& other, bool subtract);
    void scale(double factor); // scale_mut: the last core carries the factor
    double inner_product(TensorTrain& other); // contraction.rs:82-186, two MFMA GEMMs per site
    std::unique_ptr<TensorTrain> reverse();
    std::unique_ptr<TensorTrain> partial_sum(const std::vector<size_t>& dims);
    // TTCache::evaluate_many; split == 0 -> find_split_heuristic.  Returns the split that was used.
    size_t evaluate_many(const uint32_t* idx, size_t n_pts, size_t split, double* out);
    size_t find_split_heuristic(const uint32_t* idx, size_t n_pts) const;

    std::vector<DevCore> cores;
    Engine eng;

private:
    void upload_descs();
    size_t max_bond() const;
    // factorize (compression.rs:165-227): left() (M x rank) and right() (rank x N) of `eng` afterwards
    size_t factorize(const double* d_mat, int M, int N, CompressionMethod method, double tolerance,
                     bool normalize_error, size_t max_bond_dim, bool left_orthogonal);
    DevBuf<TtCoreDesc> d_desc_;
    DevBuf<uint32_t> d_idx_, d_il_, d_ir_;
    DevBuf<double> d_vals_, d_envl_, d_envr_, d_m1_, d_m2_, d_svdu_, d_svds_, d_svdvt_;
};

} // namespace t4a
