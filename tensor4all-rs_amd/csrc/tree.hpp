// tree.hpp — host-side mirror of tensor4all-treetci (crates/tensor4all-treetci/src) on top of the gfx950 engine:
// TreeTciGraph (graph.rs), TreeTCI2 state (state.rs), DefaultProposer (proposer.rs:57-88), update_edge (update.rs:22-115),
// optimize_with_proposer (optimize.rs:95-220), to_treetn site tensors (materialize.rs:17-166) and the global pivot
// search (globalpivot.rs:24-172).  Pivot tables are the master copy on the host (flat digit tables keyed by the sorted
// site list of a subtree); candidate matrices, their rrLU, the site tensors and the tree contraction live on the device.
#pragma once

#include <map>
#include <set>
#include <vector>

#include "engine.hpp"
#include "tci2.hpp"

namespace t4a {

using SubtreeKey = std::vector<size_t>; // sorted site list (key.rs)

struct TreeEdge { // graph.rs:17-41
    size_t u = 0, v = 0;
    TreeEdge() = default;
    TreeEdge(size_t a, size_t b) : u(a <= b ? a : b), v(a <= b ? b : a) {}
    bool operator<(const TreeEdge& o) const { return u != o.u ? u < o.u : v < o.v; }
    bool operator==(const TreeEdge& o) const { return u == o.u && v == o.v; }
};

class TreeGraph {
public:
    TreeGraph() = default;
    TreeGraph(size_t n_sites, const std::vector<TreeEdge>& edges);
    size_t n_sites() const { return n_; }
    bool has_edge(const TreeEdge& e) const { return edges_.count(e) != 0; }
    void require_edge(const TreeEdge& e) const;
    SubtreeKey subtree_vertices(size_t parent, size_t child) const;                  // :120-147 (one child)
    std::pair<SubtreeKey, SubtreeKey> subregion_vertices(const TreeEdge& e) const;   // :150-156
    std::vector<TreeEdge> adjacent_edges(size_t site, const TreeEdge* excluded) const; // :159-174
    std::vector<TreeEdge> edges() const { return std::vector<TreeEdge>(edges_.begin(), edges_.end()); }
    void bfs_tree(size_t root, std::vector<size_t>& parents, std::vector<size_t>& distances) const; // :237-261
    // key of the subtree hanging off `site` across every given edge (:264-289)
    std::vector<SubtreeKey> edge_in_ij_keys(size_t site, const std::vector<TreeEdge>& es) const;

private:
    size_t n_ = 0;
    std::set<TreeEdge> edges_;
    std::vector<std::vector<size_t>> adj_;
};

struct TreeTciOptions { // optimize.rs:13-76
    double tolerance = 1e-8;
    size_t max_iter = 20;
    size_t max_bond_dim = 0; // 0 == None
    bool normalize_error = true;
    bool enable_global_pivots = true;
    size_t nsearch = 5;
    size_t max_nglobal_pivot = 5;
    double tol_margin_global_search = 10.0;
    bool has_seed = false;
    uint64_t seed = 0;
    void validate() const;
};

struct EdgeSelection { // the part of MatrixLuciFactors update_edge's callers read
    size_t rank = 0;
    std::vector<size_t> row_indices, col_indices;
    std::vector<double> pivot_errors;
};

constexpr int TREE_MAX_INCOMING = 8;
struct TreeSiteDesc {
    const double* data;
    int site, d, n_in, out_dim;
    int in_off[TREE_MAX_INCOMING]; // message offset of every child
    int in_dim[TREE_MAX_INCOMING];
    int msg_off; // offset of this site's message inside one point's message block
    int pad_;
};

class TreeTci {
public:
    TreeTci(const std::vector<size_t>& local_dims, const TreeGraph& graph);

    void set_builtin(int fid, int n_acc, const double* params, const uint64_t* weights);
    void set_callback(t4a_gpu_batch_eval_fn cb, void* ctx);

    void add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots);
    // PivotCandidateProposer::candidates of the selected proposer (proposer.rs:57-249)
    void candidates(const TreeEdge& edge, IndexSet& left, IndexSet& right) const;
    void default_candidates(const TreeEdge& edge, IndexSet& left, IndexSet& right) const; // DefaultProposer :57-88
    int proposer = 0;           // 0 DefaultProposer, 1 SimpleProposer, 2 TruncatedDefaultProposer
    uint64_t proposer_seed = 0; // ::seeded(seed); rand SmallRng / DefaultHasher streams of the reference are "parity unpinned"
    EdgeSelection update_edge(const TreeEdge& edge, const RrLUOptions& options);
    void optimize(const TreeTciOptions& options);
    void crossinterpolate2(std::vector<std::vector<uint32_t>> pivots, const TreeTciOptions& options);
    std::vector<std::vector<uint32_t>> find_global_pivots(size_t nsearch, size_t max_nglobal_pivot, double tol_margin,
                                                          double abs_tol, uint64_t seed);

    void materialize(size_t center_site); // to_treetn: the site tensors stay on the device
    std::vector<double> evaluate(const uint32_t* idx, size_t n_pts); // idx n_sites x n_pts col-major
    std::vector<double> site_tensor_host(size_t site, std::vector<size_t>& dims);
    // treetn_to_tensor_train bridge for a linear chain materialised around site 0 (quanticstci/src/quantics_tci.rs:284-292):
    // site tensors [d, bond to k+1, bond to k-1] -> cores (l, s, r), device to device
    std::vector<DevCore> chain_cores();

    void flush_pivot_errors() { pivot_errors.clear(); }
    double max_bond_error() const;
    size_t max_bond_dim() const;
    const IndexSet& pivots_of(const SubtreeKey& key) const;

    // state (public like the fields of the reference struct, state.rs:41-58)
    std::vector<size_t> local_dims;
    TreeGraph graph;
    std::map<SubtreeKey, IndexSet> ijset;
    std::vector<std::map<SubtreeKey, IndexSet>> ijset_history;
    std::map<TreeEdge, double> bond_errors;
    std::vector<double> pivot_errors;
    double max_sample_value = 0.0;
    std::vector<size_t> ranks_hist;
    std::vector<double> errors_hist;
    Engine eng;

private:
    enum class FnKind { None, Builtin, Callback };
    FnKind fn_kind_ = FnKind::None;
    FnDevice fn_dev_{};
    std::vector<uint64_t> weights_;
    std::vector<size_t> offset_;
    size_t total_ = 0;
    t4a_gpu_batch_eval_fn cb_ = nullptr;
    void* cb_ctx_ = nullptr;
    void require_fn() const;

    // accumulators of a list of partial multi-indices living on `sites` (entry k, digit s <-> site sites[s])
    void accumulate(const IndexSet& set, const std::vector<size_t>& sites, std::vector<uint64_t>& acc) const;
    // d_out (rows.count x cols.count, column-major, or its transpose) = f(rows[i] on row_sites, cols[j] on col_sites);
    // d_maxbits (optional) receives bits(max sqrt(v*v)).  The two site lists together must cover every site once.
    void eval_matrix(const IndexSet& rows, const std::vector<size_t>& row_sites, const IndexSet& cols,
                     const std::vector<size_t>& col_sites, double* d_out, bool transposed, unsigned long long* d_maxbits);
    std::vector<double> eval_points(const std::vector<uint32_t>& idx, size_t n_pts);
    void update_pivot_errors(const std::vector<double>& e);
    // rows of a site tensor: (site value fastest, then the pivots of in_keys[0], in_keys[1], ...) over `sites`
    void site_rows(size_t site, const std::vector<SubtreeKey>& in_keys, IndexSet& rows, std::vector<size_t>& sites) const;

    // materialised network
    struct SiteTensor {
        std::vector<size_t> dims;
        std::vector<size_t> in_neighbors;
        bool has_parent = false;
        DevBuf<double> data;
        size_t count = 0;
    };
    bool has_net_ = false;
    size_t net_root_ = 0;
    std::vector<size_t> net_order_;
    std::vector<SiteTensor> net_;
    DevBuf<TreeSiteDesc> d_desc_;
    int msg_total_ = 0;

    DevBuf<uint64_t> d_acc_;
    PinBuf<uint64_t> h_acc_; // pinned staging of the row / column accumulators of a candidate matrix (one copy instead of two pageable ones)
    DevBuf<unsigned long long> d_maxbits_;
    DevBuf<double> d_vals_, d_a_, d_b_, d_c_, d_msg_;
    DevBuf<uint32_t> d_idx_;
    DevBuf<int> d_perm_;
    DevBuf<TrsmProblem> d_trsm_;
};

void tree_evaluate_launch(const TreeSiteDesc* d_desc, int n_sites, int msg_total, const uint32_t* d_idx, int n_pts,
                          double* d_msg, double* d_out, hipStream_t stream);

} // namespace t4a
