// tree.hip — TreeTCI driver on the gfx950 engine (see tree.hpp).  Reference: crates/tensor4all-treetci/src.
#include "tree.hpp"
#include "stdrng.hpp"
#include "smallrng.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <limits>

#include "../../include/t4a_testfunctions.h"

namespace t4a {

// ================================================================================================= graph.rs
TreeGraph::TreeGraph(size_t n_sites, const std::vector<TreeEdge>& edges) : n_(n_sites), adj_(n_sites) // :51-106
{
    if (n_sites == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "TreeTCI graph must contain at least one site");
    for (const TreeEdge& e : edges) {
        if (e.u == e.v) throw Error(T4A_GPU_INVALID_ARGUMENT, "self-loops are not allowed in TreeTCI graphs");
        if (e.v >= n_sites)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "edge endpoint " + std::to_string(e.v) + " is out of bounds for " +
                                                      std::to_string(n_sites) + " sites");
        if (!edges_.insert(e).second)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "duplicate edge (" + std::to_string(e.u) + ", " + std::to_string(e.v) + ")");
        adj_[e.u].push_back(e.v);
        adj_[e.v].push_back(e.u);
    }
    if (edges_.size() + 1 != n_sites)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "TreeTCI graph must be a tree: expected " + std::to_string(n_sites - 1) +
                                                  " edges for " + std::to_string(n_sites) + " sites, got " +
                                                  std::to_string(edges_.size()));
    for (auto& a : adj_) std::sort(a.begin(), a.end());
    std::vector<char> seen(n_sites, 0);
    std::vector<size_t> stack{0};
    seen[0] = 1;
    size_t reached = 1;
    while (!stack.empty()) {
        const size_t c = stack.back();
        stack.pop_back();
        for (size_t w : adj_[c])
            if (!seen[w]) {
                seen[w] = 1;
                ++reached;
                stack.push_back(w);
            }
    }
    if (reached != n_sites) throw Error(T4A_GPU_INVALID_ARGUMENT, "TreeTCI graph must be connected");
}

void TreeGraph::require_edge(const TreeEdge& e) const
{
    if (!has_edge(e))
        throw Error(T4A_GPU_INVALID_ARGUMENT, "edge (" + std::to_string(e.u) + ", " + std::to_string(e.v) + ") is not in the graph");
}

SubtreeKey TreeGraph::subtree_vertices(size_t parent, size_t child) const
{
    if (parent >= n_ || child >= n_) throw Error(T4A_GPU_INVALID_ARGUMENT, "site is out of bounds");
    require_edge(TreeEdge(parent, child));
    SubtreeKey sites;
    std::vector<std::pair<size_t, size_t>> stack{{parent, child}};
    while (!stack.empty()) {
        const auto pc = stack.back();
        stack.pop_back();
        sites.push_back(pc.second);
        for (size_t w : adj_[pc.second])
            if (w != pc.first) stack.push_back({pc.second, w});
    }
    std::sort(sites.begin(), sites.end());
    return sites;
}

std::pair<SubtreeKey, SubtreeKey> TreeGraph::subregion_vertices(const TreeEdge& e) const
{
    require_edge(e);
    return {subtree_vertices(e.v, e.u), subtree_vertices(e.u, e.v)};
}

std::vector<TreeEdge> TreeGraph::adjacent_edges(size_t site, const TreeEdge* excluded) const
{
    std::vector<TreeEdge> out;
    if (site >= n_) return out;
    for (size_t w : adj_[site]) {
        TreeEdge e(site, w);
        if (!excluded || !(e == *excluded)) out.push_back(e);
    }
    std::sort(out.begin(), out.end());
    return out;
}

void TreeGraph::bfs_tree(size_t root, std::vector<size_t>& parents, std::vector<size_t>& distances) const
{
    if (root >= n_) throw Error(T4A_GPU_INVALID_ARGUMENT, "root site " + std::to_string(root) + " is out of bounds");
    parents.assign(n_, n_);
    distances.assign(n_, std::numeric_limits<size_t>::max());
    std::deque<size_t> queue{root};
    distances[root] = 0;
    while (!queue.empty()) {
        const size_t c = queue.front();
        queue.pop_front();
        for (size_t w : adj_[c]) // sorted neighbours
            if (distances[w] == std::numeric_limits<size_t>::max()) {
                parents[w] = c;
                distances[w] = distances[c] + 1;
                queue.push_back(w);
            }
    }
}

std::vector<SubtreeKey> TreeGraph::edge_in_ij_keys(size_t site, const std::vector<TreeEdge>& es) const
{
    if (site >= n_) throw Error(T4A_GPU_INVALID_ARGUMENT, "site is out of bounds");
    std::vector<SubtreeKey> keys;
    for (const TreeEdge& e : es) {
        require_edge(e);
        if (e.u == site)
            keys.push_back(subtree_vertices(e.u, e.v));
        else if (e.v == site)
            keys.push_back(subtree_vertices(e.v, e.u));
        else
            throw Error(T4A_GPU_INVALID_ARGUMENT, "edge is not adjacent to the site");
    }
    return keys;
}

void TreeTciOptions::validate() const // optimize.rs:49-76
{
    if (!std::isfinite(tolerance) || tolerance < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "tolerance must be finite and nonnegative");
    if (max_iter == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "max_iter must be positive");
    if (!std::isfinite(tol_margin_global_search) || tol_margin_global_search < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "tol_margin_global_search must be finite and nonnegative");
}

// ================================================================================================= state.rs
TreeTci::TreeTci(const std::vector<size_t>& dims, const TreeGraph& g) : local_dims(dims), graph(g) // :66-103
{
    if (!(dims.size() > 1)) throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
    if (dims.size() != g.n_sites())
        throw Error(T4A_GPU_INVALID_ARGUMENT, "local_dims length " + std::to_string(dims.size()) +
                                                  " must match graph site count " + std::to_string(g.n_sites()));
    for (size_t s = 0; s < dims.size(); ++s) {
        if (dims[s] == 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension at site " + std::to_string(s) + " must be positive");
        if (dims[s] > 0xFFFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "local dimension too large");
    }
    for (const TreeEdge& e : g.edges()) bond_errors[e] = 0.0;
    offset_.resize(dims.size());
    total_ = 0;
    for (size_t s = 0; s < dims.size(); ++s) {
        offset_[s] = total_;
        total_ += dims[s];
    }
    d_maxbits_.reserve(2);
}

void TreeTci::set_builtin(int fid, int n_acc, const double* params, const uint64_t* weights)
{
    if (fid < 0 || fid >= T4A_FN_COUNT) throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown built-in function id");
    if (n_acc < 1 || n_acc > T4A_FN_MAX_ACC) throw Error(T4A_GPU_INVALID_ARGUMENT, "n_acc out of range");
    fn_dev_.fid = fid;
    fn_dev_.n_acc = n_acc;
    std::memcpy(fn_dev_.params, params, sizeof(double) * T4A_FN_MAX_PARAMS);
    weights_.assign(weights, weights + (size_t)n_acc * total_);
    fn_kind_ = FnKind::Builtin;
}

void TreeTci::set_callback(t4a_gpu_batch_eval_fn cb, void* ctx)
{
    if (!cb) throw Error(T4A_GPU_NULL_POINTER, "callback is null");
    cb_ = cb;
    cb_ctx_ = ctx;
    fn_kind_ = FnKind::Callback;
}

void TreeTci::require_fn() const
{
    if (fn_kind_ == FnKind::None)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "no function set: call t4a_gpu_treetci_set_builtin_function or _set_callback");
}

const IndexSet& TreeTci::pivots_of(const SubtreeKey& key) const
{
    auto it = ijset.find(key);
    if (it == ijset.end()) throw Error(T4A_GPU_INVALID_ARGUMENT, "missing pivot set for subtree key");
    return it->second;
}

void TreeTci::add_global_pivots(const std::vector<std::vector<uint32_t>>& pivots) // :110-165
{
    const size_t n = local_dims.size();
    for (const auto& p : pivots)
        if (p.size() != n) throw Error(T4A_GPU_INVALID_ARGUMENT, "each global pivot must contain one index per site");
    for (const auto& p : pivots)
        for (size_t s = 0; s < n; ++s)
            if (p[s] >= local_dims[s])
                throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot value " + std::to_string(p[s]) +
                                                          " is out of bounds for site " + std::to_string(s));
    const auto edges = graph.edges();
    std::vector<std::pair<SubtreeKey, SubtreeKey>> regions;
    for (const TreeEdge& e : edges) regions.push_back(graph.subregion_vertices(e));
    std::vector<uint32_t> proj;
    for (const auto& p : pivots)
        for (const auto& keys : regions)
            for (const SubtreeKey* key : {&keys.first, &keys.second}) {
                IndexSet& set = ijset[*key];
                set.width = key->size();
                proj.clear();
                for (size_t s : *key) proj.push_back(p[s]);
                if (!set.contains(proj.data())) set.push(proj.data());
            }
    SubtreeKey full(n);
    for (size_t s = 0; s < n; ++s) full[s] = s;
    if (!ijset.count(full)) {
        IndexSet empty;
        empty.width = n;
        ijset[full] = empty;
    }
    has_net_ = false;
}

void TreeTci::update_pivot_errors(const std::vector<double>& e) // :178-184
{
    if (pivot_errors.size() < e.size()) pivot_errors.resize(e.size(), 0.0);
    for (size_t k = 0; k < e.size(); ++k) pivot_errors[k] = std::max(pivot_errors[k], e[k]);
}

double TreeTci::max_bond_error() const
{
    double m = 0.0;
    for (const auto& kv : bond_errors) m = std::max(m, kv.second);
    return m;
}

size_t TreeTci::max_bond_dim() const
{
    size_t m = 0;
    for (const auto& kv : ijset) m = std::max(m, kv.second.count);
    return m;
}

// ================================================================================================= proposer.rs
namespace {

size_t subtree_position(const SubtreeKey& key, size_t site)
{
    auto it = std::lower_bound(key.begin(), key.end(), site);
    if (it == key.end() || *it != site) throw Error(T4A_GPU_INTERNAL_ERROR, "site not found in subtree key");
    return (size_t)(it - key.begin());
}

// union_with_history (:268-290): first occurrences of `values`, then unseen history columns
IndexSet union_with_history(const IndexSet& values, const IndexSet* history)
{
    IndexSet out;
    out.width = values.width;
    const size_t w = values.width;
    auto less = [w](const uint32_t* a, const uint32_t* b) { return std::lexicographical_compare(a, a + w, b, b + w); };
    std::set<const uint32_t*, decltype(less)> seen(less);
    out.d.reserve(values.d.size());
    // pointers into `values` / `history` stay valid for the whole call
    for (size_t k = 0; k < values.count; ++k)
        if (seen.insert(values.at(k)).second) out.push(values.at(k));
    if (history)
        for (size_t k = 0; k < history->count; ++k)
            if (seen.insert(history->at(k)).second) out.push(history->at(k));
    return out;
}

} // namespace

void TreeTci::candidates(const TreeEdge& edge, IndexSet& left, IndexSet& right) const
{
    if (proposer == 0) {
        default_candidates(edge, left, right);
        return;
    }
    graph.require_edge(edge);
    const auto keys = graph.subregion_vertices(edge);
    auto ncols = [&](const SubtreeKey& k) {
        auto it = ijset.find(k);
        return it == ijset.end() ? (uint64_t)0 : (uint64_t)it->second.count;
    };
    // rng_for_edge (:360-387): std DefaultHasher (SipHash-1-3, zero key) over seed, tag, edge { u <= v }, history length and the two
    // pivot counts, then SmallRng::seed_from_u64 (smallrng.hpp)
    DefaultHasher hasher;
    hasher.write_u64(proposer_seed);
    if (proposer == 1) hasher.write_str("simple", 6);
    else hasher.write_str("truncated_default", 17);
    hasher.write_usize(std::min(edge.u, edge.v));
    hasher.write_usize(std::max(edge.u, edge.v));
    hasher.write_usize(ijset_history.size());
    hasher.write_usize((size_t)ncols(keys.first));
    hasher.write_usize((size_t)ncols(keys.second));
    SmallRng rng(hasher.finish());
    const size_t ichi = local_dims[edge.u] * pivots_of(keys.first).count;
    const size_t jchi = local_dims[edge.v] * pivots_of(keys.second).count;
    const std::map<SubtreeKey, IndexSet>* history = ijset_history.empty() ? nullptr : &ijset_history.back();
    auto hist_of = [&](const SubtreeKey& key) -> const IndexSet* {
        if (!history) return nullptr;
        auto it = history->find(key);
        return it == history->end() ? nullptr : &it->second;
    };
    if (proposer == 1) { // SimpleProposer (:127-160): d * chi uniformly random candidates per side
        auto random = [&](const SubtreeKey& key, size_t size) {
            IndexSet out;
            out.width = key.size();
            std::vector<uint32_t> c(key.size());
            for (size_t k = 0; k < size; ++k) {
                for (size_t s = 0; s < key.size(); ++s) c[s] = (uint32_t)rng.random_range(local_dims[key[s]]);
                out.push(c.data());
            }
            return out;
        };
        const IndexSet iset = random(keys.first, ichi);
        const IndexSet jset = random(keys.second, jchi);
        left = union_with_history(iset, hist_of(keys.first));
        right = union_with_history(jset, hist_of(keys.second));
        return;
    }
    if (proposer == 2) { // TruncatedDefaultProposer (:205-249): ordered sample of the default candidates
        IndexSet di, dj;
        default_candidates(edge, di, dj);
        auto sample = [&](const IndexSet& cand, size_t max_size) { // sample_ordered_candidates (:389-409)
            if (cand.count <= max_size) return cand;
            std::vector<size_t> idx(cand.count);
            for (size_t k = 0; k < idx.size(); ++k) idx[k] = k;
            rng.shuffle(idx); // selected_indices.shuffle(rng)
            idx.resize(max_size);
            std::sort(idx.begin(), idx.end());
            IndexSet out;
            out.width = cand.width;
            for (size_t k : idx) out.push(cand.at(k));
            return out;
        };
        left = sample(di, ichi);
        right = sample(dj, jchi);
        return;
    }
    throw Error(T4A_GPU_INVALID_ARGUMENT, "unknown proposer");
}

void TreeTci::default_candidates(const TreeEdge& edge, IndexSet& left, IndexSet& right) const // :57-88
{
    graph.require_edge(edge);
    const auto keys = graph.subregion_vertices(edge);
    const std::map<SubtreeKey, IndexSet>* history = ijset_history.empty() ? nullptr : &ijset_history.back();
    auto side = [&](size_t vtx, const SubtreeKey& key) {
        const auto adjacent = graph.adjacent_edges(vtx, &edge);
        const auto in_keys = graph.edge_in_ij_keys(vtx, adjacent);
        // pivot_set (:292-329): cartesian product of the incoming pivot tables, the last key running fastest
        IndexSet pivots;
        pivots.width = key.size();
        std::vector<uint32_t> zero(key.size(), 0);
        pivots.push(zero.data());
        for (const SubtreeKey& in_key : in_keys) {
            const IndexSet& incoming = pivots_of(in_key);
            if (incoming.width != in_key.size()) throw Error(T4A_GPU_INTERNAL_ERROR, "pivot length does not match its subtree key");
            std::vector<size_t> pos(in_key.size());
            for (size_t k = 0; k < in_key.size(); ++k) pos[k] = subtree_position(key, in_key[k]);
            IndexSet next;
            next.width = key.size();
            next.d.reserve(pivots.count * incoming.count * key.size());
            std::vector<uint32_t> merged(key.size());
            for (size_t b = 0; b < pivots.count; ++b)
                for (size_t j = 0; j < incoming.count; ++j) {
                    std::copy(pivots.at(b), pivots.at(b) + key.size(), merged.begin());
                    for (size_t k = 0; k < in_key.size(); ++k) merged[pos[k]] = incoming.at(j)[k];
                    next.push(merged.data());
                }
            pivots = std::move(next);
        }
        // kronecker (:331-349) with the local index of the edge endpoint
        const size_t site_index = subtree_position(key, vtx);
        IndexSet set;
        set.width = key.size();
        set.d.reserve(pivots.count * local_dims[vtx] * key.size());
        std::vector<uint32_t> cand(key.size());
        for (size_t b = 0; b < pivots.count; ++b)
            for (size_t v = 0; v < local_dims[vtx]; ++v) {
                std::copy(pivots.at(b), pivots.at(b) + key.size(), cand.begin());
                cand[site_index] = (uint32_t)v;
                set.push(cand.data());
            }
        const IndexSet* hist = nullptr;
        if (history) {
            auto it = history->find(key);
            if (it != history->end()) hist = &it->second;
        }
        return union_with_history(set, hist);
    };
    left = side(edge.u, keys.first);
    right = side(edge.v, keys.second);
}

// ================================================================================================= evaluation
void TreeTci::accumulate(const IndexSet& set, const std::vector<size_t>& sites, std::vector<uint64_t>& acc) const
{
    const int K = fn_dev_.n_acc;
    acc.assign(std::max<size_t>(set.count, 1) * (size_t)K, 0);
    for (size_t e = 0; e < set.count; ++e) {
        const uint32_t* v = set.at(e);
        for (int k = 0; k < K; ++k) {
            uint64_t a = 0;
            const uint64_t* w = weights_.data() + (size_t)k * total_;
            for (size_t s = 0; s < sites.size(); ++s) a += w[offset_[sites[s]] + v[s]];
            acc[e * K + k] = a;
        }
    }
}

void TreeTci::eval_matrix(const IndexSet& rows, const std::vector<size_t>& row_sites, const IndexSet& cols,
                          const std::vector<size_t>& col_sites, double* d_out, bool transposed, unsigned long long* d_maxbits)
{
    require_fn();
    const size_t n = local_dims.size();
    const size_t nr = rows.count, nc = cols.count;
    if (nr == 0 || nc == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "at least one point is required");
    if (rows.width != row_sites.size() || cols.width != col_sites.size() || row_sites.size() + col_sites.size() != n)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "global point assembly left some sites unassigned");
    {
        std::vector<char> seen(n, 0);
        for (const auto* lst : {&row_sites, &col_sites})
            for (size_t s : *lst) {
                if (s >= n) throw Error(T4A_GPU_INVALID_ARGUMENT, "site " + std::to_string(s) + " is out of bounds");
                if (seen[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "site " + std::to_string(s) + " was assigned more than once");
                seen[s] = 1;
            }
    }
    hipStream_t st = eng.stream();
    if (nr > 0x7FFFFFFFull || nc > 0x7FFFFFFFull) throw Error(T4A_GPU_INVALID_ARGUMENT, "candidate matrix too large");
    if (fn_kind_ == FnKind::Builtin) {
        std::vector<uint64_t> ra, rb;
        accumulate(rows, row_sites, ra);
        accumulate(cols, col_sites, rb);
        d_acc_.reserve(ra.size() + rb.size());
        h_acc_.reserve(ra.size() + rb.size());
        std::memcpy(h_acc_.get(), ra.data(), ra.size() * sizeof(uint64_t));
        std::memcpy(h_acc_.get() + ra.size(), rb.data(), rb.size() * sizeof(uint64_t));
        T4A_HIP(hipMemcpyAsync(d_acc_.get(), h_acc_.get(), (ra.size() + rb.size()) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        pi_eval_launch(fn_dev_, d_acc_.get(), (int)nr, d_acc_.get() + ra.size(), (int)nc, d_out,
                       transposed ? (int)nc : (int)nr, transposed, d_maxbits, st);
        T4A_HIP(hipGetLastError());
        T4A_HIP(hipStreamSynchronize(st)); // (the pinned staging block is reused by the next call)
    } else {
        // GlobalIndexBatch (batch.rs): (n_sites, n_points) column-major, rows running fastest (update.rs:218-231)
        const size_t npts = nr * nc;
        std::vector<uint32_t> idx(npts * n);
        for (size_t j = 0; j < nc; ++j)
            for (size_t i = 0; i < nr; ++i) {
                uint32_t* p = idx.data() + (j * nr + i) * n;
                for (size_t s = 0; s < row_sites.size(); ++s) p[row_sites[s]] = rows.at(i)[s];
                for (size_t s = 0; s < col_sites.size(); ++s) p[col_sites[s]] = cols.at(j)[s];
            }
        std::vector<double> vals(npts);
        const int64_t got = cb_(cb_ctx_, idx.data(), n, npts, vals.data());
        if (got < 0 || (size_t)got != npts)
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch evaluator returned " + std::to_string(got) + " values for " +
                                                    std::to_string(npts) + " candidate-matrix entries");
        if (transposed) {
            d_vals_.reserve(npts);
            T4A_HIP(hipMemcpyAsync(d_vals_.get(), vals.data(), npts * sizeof(double), hipMemcpyHostToDevice, st));
            transpose_launch(d_vals_.get(), (int)nr, (int)nc, (int)nr, d_out, (int)nc, st);
        } else {
            T4A_HIP(hipMemcpyAsync(d_out, vals.data(), npts * sizeof(double), hipMemcpyHostToDevice, st));
        }
        if (d_maxbits) absmax_launch(d_out, npts, d_maxbits, st);
        T4A_HIP(hipGetLastError());
        T4A_HIP(hipStreamSynchronize(st));
    }
}

std::vector<double> TreeTci::eval_points(const std::vector<uint32_t>& idx, size_t n_pts)
{
    require_fn();
    const size_t n = local_dims.size();
    std::vector<double> out(n_pts);
    if (n_pts == 0) return out;
    if (fn_kind_ == FnKind::Builtin) {
        // every point is a "row" over all sites; the single column carries no site
        IndexSet rows, cols;
        rows.width = n;
        rows.count = n_pts;
        rows.d = idx;
        cols.width = 0;
        cols.count = 1;
        std::vector<size_t> all(n);
        for (size_t s = 0; s < n; ++s) all[s] = s;
        d_c_.reserve(n_pts);
        eval_matrix(rows, all, cols, {}, d_c_.get(), false, nullptr);
        T4A_HIP(hipMemcpy(out.data(), d_c_.get(), n_pts * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        const int64_t got = cb_(cb_ctx_, idx.data(), n, n_pts, out.data());
        if (got < 0 || (size_t)got != n_pts)
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch evaluator returned " + std::to_string(got) + " values for " +
                                                    std::to_string(n_pts) + " points");
    }
    return out;
}

// ================================================================================================= update.rs
EdgeSelection TreeTci::update_edge(const TreeEdge& edge, const RrLUOptions& options) // :22-115
{
    require_fn();
    static const bool host_prof = std::getenv("T4A_HOST_PROFILE") != nullptr;
    static double hp[4] = {0, 0, 0, 0};
    static long hp_n = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    const auto keys = graph.subregion_vertices(edge);
    IndexSet lc, rc;
    candidates(edge, lc, rc);
    if (lc.count == 0 || rc.count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "proposer returned empty candidate list for edge");
    const size_t M = lc.count, N = rc.count;
    const auto t1 = now();
    auto t2 = t1;
    RrLUOptions o = options;
    o.left_orthogonal = true;
    LuciResult lu;
    static const bool no_fuse = diag_env("T4A_NO_FUSED_PI") != nullptr;
    if (fn_kind_ == FnKind::Builtin && !no_fuse) {
        // only the integer accumulators travel: the rrLU kernel evaluates the candidate matrix into its registers
        std::vector<uint64_t> ra, rb;
        accumulate(lc, keys.first, ra);
        accumulate(rc, keys.second, rb);
        t2 = now();
        d_acc_.reserve(ra.size() + rb.size());
        hipStream_t st = eng.stream();
        // (one copy out of pinned memory — two copies out of pageable vectors cost ~30 us of an edge update; luci() below ends with a
        // stream synchronisation, the staging block is free again then)
        h_acc_.reserve(ra.size() + rb.size());
        std::memcpy(h_acc_.get(), ra.data(), ra.size() * sizeof(uint64_t));
        std::memcpy(h_acc_.get() + ra.size(), rb.data(), rb.size() * sizeof(uint64_t));
        T4A_HIP(hipMemcpyAsync(d_acc_.get(), h_acc_.get(), (ra.size() + rb.size()) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        FusedPi fp;
        fp.fn = fn_dev_;
        fp.d_rowacc = d_acc_.get();
        fp.d_colacc = d_acc_.get() + ra.size();
        lu = eng.luci(nullptr, (int)M, (int)N, o, false, false, &fp); // synchronises the stream
    } else {
        double* d_pi = eng.pi(M * N);
        eval_matrix(lc, keys.first, rc, keys.second, d_pi, false, nullptr);
        lu = eng.luci(d_pi, (int)M, (int)N, o, false, false);
    }
    if (host_prof) {
        const auto t3 = now();
        hp[0] += us(t0, t1);
        hp[1] += us(t1, t2);
        hp[2] += us(t2, t3);
        if (++hp_n % 200 == 0)
            std::fprintf(stderr, "[host profile] tree update_edge: candidates %.1f us, accumulators %.1f us, upload+rrLU+wait %.1f us (avg of %ld)\n",
                         hp[0] / hp_n, hp[1] / hp_n, hp[2] / hp_n, hp_n);
    }
    if (lu.abs_max > max_sample_value) max_sample_value = lu.abs_max;

    EdgeSelection sel;
    sel.rank = (size_t)lu.rank;
    sel.pivot_errors = lu.pivot_errors;
    for (int k = 0; k < lu.rank; ++k) {
        sel.row_indices.push_back((size_t)lu.row_perm[k]);
        sel.col_indices.push_back((size_t)lu.col_perm[k]);
    }
    // keep at least one index per side (update.rs:62-78)
    std::vector<size_t> rows = sel.row_indices.empty() ? std::vector<size_t>{0} : sel.row_indices;
    std::vector<size_t> cols = sel.col_indices.empty() ? std::vector<size_t>{0} : sel.col_indices;
    IndexSet li, ri;
    li.width = lc.width;
    ri.width = rc.width;
    for (size_t r : rows) li.push(lc.at(r));
    for (size_t c : cols) ri.push(rc.at(c));
    ijset[keys.first] = std::move(li);
    ijset[keys.second] = std::move(ri);
    bond_errors[edge] = sel.pivot_errors.empty() ? 0.0 : sel.pivot_errors.back();
    update_pivot_errors(sel.pivot_errors);
    has_net_ = false;
    return sel;
}

// ================================================================================================= optimize.rs
void TreeTci::optimize(const TreeTciOptions& options) // :95-220 (DefaultProposer, AllEdges)
{
    options.validate();
    require_fn();
    ranks_hist.clear();
    errors_hist.clear();
    std::vector<size_t> nglobal;
    constexpr size_t INNER_EDGE_PASSES = 2, NCHECK_HISTORY = 3;
    const auto edges = graph.edges();
    for (size_t iter = 0; iter < options.max_iter; ++iter) {
        for (size_t pass = 0; pass < INNER_EDGE_PASSES; ++pass) {
            const double scale = options.normalize_error && max_sample_value > 0.0 ? max_sample_value : 1.0;
            RrLUOptions ko;
            ko.rel_tol = 1e-14;
            ko.abs_tol = options.tolerance * scale;
            ko.max_bond_dim = options.max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : options.max_bond_dim;
            ko.left_orthogonal = true;
            ijset_history.push_back(ijset);
            flush_pivot_errors();
            for (const TreeEdge& e : edges) update_edge(e, ko);
        }
        ranks_hist.push_back(max_bond_dim());
        errors_hist.push_back(options.normalize_error && max_sample_value > 0.0 ? max_bond_error() / max_sample_value
                                                                                 : max_bond_error());
        if (options.enable_global_pivots && iter + 1 < options.max_iter) {
            const double scale = options.normalize_error && max_sample_value > 0.0 ? max_sample_value : 1.0;
            // reference: `rand::random()` when no seed is given; a fixed stream here keeps runs reproducible
            const uint64_t seed = options.has_seed ? options.seed + (uint64_t)iter : 0x243F6A8885A308D3ull + (uint64_t)iter;
            auto pivots = find_global_pivots(options.nsearch, options.max_nglobal_pivot, options.tol_margin_global_search,
                                             options.tolerance * scale, seed);
            add_global_pivots(pivots);
            nglobal.push_back(pivots.size());
        } else {
            nglobal.push_back(0);
        }
        if (errors_hist.size() >= NCHECK_HISTORY) {
            const size_t m = errors_hist.size();
            bool errors_converged = true, no_global = true, saturated = options.max_bond_dim != 0;
            size_t min_rank = std::numeric_limits<size_t>::max();
            for (size_t k = m - NCHECK_HISTORY; k < m; ++k) {
                errors_converged = errors_converged && errors_hist[k] < options.tolerance;
                no_global = no_global && nglobal[k] == 0;
                min_rank = std::min(min_rank, ranks_hist[k]);
                saturated = saturated && ranks_hist[k] >= options.max_bond_dim;
            }
            const bool rank_stable = min_rank == ranks_hist.back();
            if ((errors_converged && no_global && rank_stable) || saturated) break;
        }
    }
}

void TreeTci::crossinterpolate2(std::vector<std::vector<uint32_t>> pivots, const TreeTciOptions& options) // api.rs:21-96
{
    options.validate();
    require_fn();
    const size_t n = local_dims.size();
    if (pivots.empty()) pivots.push_back(std::vector<uint32_t>(n, 0));
    add_global_pivots(pivots);
    std::vector<uint32_t> flat;
    for (const auto& p : pivots) flat.insert(flat.end(), p.begin(), p.end());
    const std::vector<double> vals = eval_points(flat, pivots.size());
    double m = 0.0;
    for (double v : vals) m = std::max(m, std::sqrt(v * v));
    max_sample_value = m;
    if (!(max_sample_value > 0.0)) throw Error(T4A_GPU_INVALID_ARGUMENT, "initial pivots must not all evaluate to zero");
    optimize(options);
}

// ================================================================================================= globalpivot.rs
std::vector<std::vector<uint32_t>> TreeTci::find_global_pivots(size_t nsearch, size_t max_nglobal_pivot, double tol_margin,
                                                               double abs_tol, uint64_t seed) // :24-172
{
    if (!std::isfinite(abs_tol) || abs_tol < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot search abs_tol must be finite and nonnegative");
    if (!std::isfinite(tol_margin) || tol_margin < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "global pivot search tol_margin must be finite and nonnegative");
    std::vector<std::vector<uint32_t>> pivots;
    if (nsearch == 0 || max_nglobal_pivot == 0) return pivots;
    const size_t n = local_dims.size();
    materialize(0);
    StdRng rng(seed); // globalpivot.rs:118 (stdrng.hpp)
    std::vector<uint32_t> idx;
    for (size_t k = 0; k < nsearch; ++k) {
        std::vector<uint32_t> start(n);
        for (size_t s = 0; s < n; ++s) start[s] = (uint32_t)rng.random_range(local_dims[s]);
        for (size_t s = 0; s < n; ++s)
            for (size_t v = 0; v < local_dims[s]; ++v) {
                const size_t base = idx.size();
                idx.insert(idx.end(), start.begin(), start.end());
                idx[base + s] = (uint32_t)v;
            }
    }
    const size_t npts = idx.size() / n;
    const std::vector<double> fv = eval_points(idx, npts);
    const std::vector<double> tv = evaluate(idx.data(), npts);
    std::vector<std::pair<double, size_t>> best; // (error, point number)
    size_t q = 0;
    for (size_t k = 0; k < nsearch; ++k) {
        bool have = false;
        double be = 0.0;
        size_t bq = 0;
        for (size_t s = 0; s < n; ++s)
            for (size_t v = 0; v < local_dims[s]; ++v, ++q) {
                const double re = fv[q] - tv[q];
                const double err = std::sqrt(re * re + 0.0 * 0.0);
                if (!have || err > be) {
                    have = true;
                    be = err;
                    bq = q;
                }
            }
        if (have && be > abs_tol * tol_margin) best.push_back({be, bq});
    }
    std::stable_sort(best.begin(), best.end(),
                     [](const std::pair<double, size_t>& a, const std::pair<double, size_t>& b) { return a.first > b.first; });
    for (const auto& b : best) {
        std::vector<uint32_t> p(idx.begin() + b.second * n, idx.begin() + (b.second + 1) * n);
        if (std::find(pivots.begin(), pivots.end(), p) == pivots.end()) {
            pivots.push_back(std::move(p));
            if (pivots.size() >= max_nglobal_pivot) break;
        }
    }
    return pivots;
}

// ================================================================================================= materialize.rs
void TreeTci::site_rows(size_t site, const std::vector<SubtreeKey>& in_keys, IndexSet& rows, std::vector<size_t>& sites) const
{
    // fill_tensor_values (:198-243): central values run fastest, then the pivots of in_keys[0], in_keys[1], ...
    sites.clear();
    sites.push_back(site);
    std::vector<const IndexSet*> sets;
    size_t combos = 1;
    for (const SubtreeKey& k : in_keys) {
        const IndexSet& s = pivots_of(k);
        sets.push_back(&s);
        combos *= s.count;
        sites.insert(sites.end(), k.begin(), k.end());
    }
    const size_t d = local_dims[site];
    rows.clear();
    rows.width = sites.size();
    rows.d.reserve(combos * d * rows.width);
    std::vector<size_t> ctr(in_keys.size(), 0);
    std::vector<uint32_t> entry(rows.width);
    for (size_t c = 0; c < combos; ++c) {
        size_t off = 1;
        for (size_t a = 0; a < in_keys.size(); ++a) {
            std::copy(sets[a]->at(ctr[a]), sets[a]->at(ctr[a]) + sets[a]->width, entry.begin() + off);
            off += sets[a]->width;
        }
        for (size_t v = 0; v < d; ++v) {
            entry[0] = (uint32_t)v;
            rows.push(entry.data());
        }
        for (size_t a = 0; a < in_keys.size(); ++a) {
            if (++ctr[a] < sets[a]->count) break;
            ctr[a] = 0;
        }
    }
}

void TreeTci::materialize(size_t center_site) // to_treetn :17-103
{
    require_fn();
    const size_t n = graph.n_sites();
    std::vector<size_t> parents, distances;
    graph.bfs_tree(center_site, parents, distances);
    std::map<TreeEdge, size_t> bond_dim;
    for (const TreeEdge& e : graph.edges()) {
        const auto keys = graph.subregion_vertices(e);
        auto li = ijset.find(keys.first), ri = ijset.find(keys.second);
        const size_t lr = li == ijset.end() ? 0 : li->second.count;
        const size_t rr = ri == ijset.end() ? 0 : ri->second.count;
        if (lr != rr)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "bond ranks disagree across edge (" + std::to_string(e.u) + ", " +
                                                      std::to_string(e.v) + "): left " + std::to_string(lr) + ", right " +
                                                      std::to_string(rr));
        bond_dim[e] = std::max<size_t>(lr, 1);
    }
    net_order_.resize(n);
    for (size_t s = 0; s < n; ++s) net_order_[s] = s;
    std::sort(net_order_.begin(), net_order_.end(), [&](size_t a, size_t b) {
        return distances[a] != distances[b] ? distances[a] < distances[b] : a < b;
    });
    std::vector<SiteTensor> net(n);
    hipStream_t st = eng.stream();
    for (size_t site : net_order_) {
        SiteTensor& t = net[site];
        t.has_parent = parents[site] != n;
        TreeEdge parent_edge;
        if (t.has_parent) parent_edge = TreeEdge(site, parents[site]);
        const auto incoming = graph.adjacent_edges(site, t.has_parent ? &parent_edge : nullptr);
        const auto in_keys = graph.edge_in_ij_keys(site, incoming);
        IndexSet rows;
        std::vector<size_t> row_sites;
        site_rows(site, in_keys, rows, row_sites);
        if (rows.count == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "at least one point is required");
        t.dims.push_back(local_dims[site]);
        for (const TreeEdge& e : incoming) {
            t.dims.push_back(bond_dim.at(e));
            t.in_neighbors.push_back(e.u == site ? e.v : e.u);
        }
        if (!t.has_parent) {
            IndexSet one;
            one.width = 0;
            one.count = 1;
            t.count = rows.count;
            t.data.reserve(t.count);
            eval_matrix(rows, row_sites, one, {}, t.data.get(), false, nullptr);
            continue;
        }
        // site_tensor_with_parent (:105-166): T = Pi1 * P^{-1}
        t.dims.push_back(bond_dim.at(parent_edge));
        const SubtreeKey out_key = graph.edge_in_ij_keys(site, {parent_edge})[0];
        const IndexSet& out_piv = pivots_of(out_key);
        const auto keys = graph.subregion_vertices(parent_edge);
        const SubtreeKey& side_key = std::binary_search(keys.first.begin(), keys.first.end(), site) ? keys.first : keys.second;
        const IndexSet& side_piv = pivots_of(side_key);
        const size_t R = rows.count, C = out_piv.count;
        if (C == 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "at least one point is required");
        if (side_piv.count != C)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "pivot matrix for site " + std::to_string(site) + " is not square: " +
                                                      std::to_string(side_piv.count) + " x " + std::to_string(C));
        t.count = R * C;
        t.data.reserve(t.count);
        // A = P^T (C x C) and B = Pi1^T (C x R), straight from the evaluator (backend.rs:181-246 transposes both)
        d_a_.reserve(C * C);
        d_b_.reserve(C * R);
        T4A_HIP(hipMemsetAsync(d_maxbits_.get(), 0, sizeof(unsigned long long), st));
        eval_matrix(side_piv, side_key, out_piv, out_key, d_a_.get(), true, d_maxbits_.get());
        unsigned long long bits = 0;
        T4A_HIP(hipMemcpy(&bits, d_maxbits_.get(), sizeof(bits), hipMemcpyDeviceToHost));
        double pmax;
        std::memcpy(&pmax, &bits, sizeof(pmax));
        if (pmax < std::numeric_limits<double>::epsilon()) { // zero pivot matrix: zero site tensor (:146-160)
            T4A_HIP(hipMemsetAsync(t.data.get(), 0, t.count * sizeof(double), st));
            continue;
        }
        eval_matrix(rows, row_sites, out_piv, out_key, d_b_.get(), true, nullptr);
        RrLUOptions fo;
        fo.rel_tol = 0.0;
        fo.abs_tol = 0.0;
        fo.left_orthogonal = true;
        LuciResult lu = eng.luci(d_a_.get(), (int)C, (int)C, fo, false, true);
        if ((size_t)lu.rank < C) throw Error(T4A_GPU_SINGULAR_MATRIX, "full_piv_lu_solve failed: singular pivot matrix");
        d_perm_.reserve(2 * C);
        T4A_HIP(hipMemcpyAsync(d_perm_.get(), lu.row_perm.data(), C * sizeof(int), hipMemcpyHostToDevice, st));
        T4A_HIP(hipMemcpyAsync(d_perm_.get() + C, lu.col_perm.data(), C * sizeof(int), hipMemcpyHostToDevice, st));
        d_c_.reserve(C * R);
        gather_launch(d_b_.get(), (int)C, d_perm_.get(), (int)C, nullptr, (int)R, d_c_.get(), (int)C, st);
        TrsmProblem tp[2];
        tp[0] = TrsmProblem{eng.lu_buf(), (int)C, (int)C, d_c_.get(), (int)C, (int)R, 1, 1, nullptr};
        tp[1] = TrsmProblem{eng.lu_buf(), (int)C, (int)C, d_c_.get(), (int)C, (int)R, 0, 0, nullptr};
        d_trsm_.reserve(2);
        T4A_HIP(hipMemcpyAsync(d_trsm_.get(), tp, sizeof(tp), hipMemcpyHostToDevice, st));
        trsm_left_batched_launch(d_trsm_.get(), 1, (int)C, (int)R, st);
        trsm_left_batched_launch(d_trsm_.get() + 1, 1, (int)C, (int)R, st);
        // Z[col_perm[k], :] = Zc[k, :], then T = Z^T (R x C)
        scatter_rows_launch(d_c_.get(), (int)C, d_perm_.get() + C, (int)C, (int)R, d_b_.get(), (int)C, st);
        transpose_launch(d_b_.get(), (int)C, (int)R, (int)C, t.data.get(), (int)R, st);
        T4A_HIP(hipGetLastError());
        T4A_HIP(hipStreamSynchronize(st)); // lu / tp are host objects
    }
    // descriptors for the contraction kernel: leaves first
    std::vector<TreeSiteDesc> desc(n);
    std::vector<int> msg_off(n, 0);
    int off = 0;
    for (size_t s = 0; s < n; ++s) {
        msg_off[s] = off;
        off += net[s].has_parent ? (int)net[s].dims.back() : 1;
    }
    for (size_t oi = 0; oi < n; ++oi) {
        const size_t site = net_order_[n - 1 - oi];
        const SiteTensor& t = net[site];
        TreeSiteDesc& d = desc[oi];
        std::memset(&d, 0, sizeof(d));
        if (t.in_neighbors.size() > (size_t)TREE_MAX_INCOMING)
            throw Error(T4A_GPU_NOT_IMPLEMENTED, "tree evaluation supports at most " + std::to_string(TREE_MAX_INCOMING) +
                                                     " incoming bonds per site");
        d.data = t.data.get();
        d.site = (int)site;
        d.d = (int)local_dims[site];
        d.n_in = (int)t.in_neighbors.size();
        d.out_dim = t.has_parent ? (int)t.dims.back() : 1;
        for (size_t a = 0; a < t.in_neighbors.size(); ++a) {
            d.in_off[a] = msg_off[t.in_neighbors[a]];
            d.in_dim[a] = (int)t.dims[1 + a];
        }
        d.msg_off = msg_off[site];
    }
    d_desc_.reserve(n);
    T4A_HIP(hipMemcpy(d_desc_.get(), desc.data(), n * sizeof(TreeSiteDesc), hipMemcpyHostToDevice));
    T4A_HIP(hipStreamSynchronize(st));
    msg_total_ = off;
    net_ = std::move(net);
    net_root_ = center_site;
    has_net_ = true;
}

std::vector<double> TreeTci::site_tensor_host(size_t site, std::vector<size_t>& dims)
{
    if (!has_net_) throw Error(T4A_GPU_INVALID_ARGUMENT, "no materialised network: call materialize first");
    if (site >= net_.size()) throw Error(T4A_GPU_INVALID_ARGUMENT, "site out of range");
    const SiteTensor& t = net_[site];
    dims = t.dims;
    std::vector<double> out(t.count);
    T4A_HIP(hipStreamSynchronize(eng.stream()));
    if (t.count) T4A_HIP(hipMemcpy(out.data(), t.data.get(), t.count * sizeof(double), hipMemcpyDeviceToHost));
    return out;
}

// out[l + L*(s + S*r)] = in[s + S*(r + R*l)]
__global__ void chain_core_kernel(const double* __restrict__ in, int S, int R, int L, double* __restrict__ out)
{
    const size_t total = (size_t)S * R * L;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t l = e % (size_t)L;
        const size_t s = (e / (size_t)L) % (size_t)S;
        const size_t r = e / ((size_t)L * (size_t)S);
        out[e] = in[s + (size_t)S * (r + (size_t)R * l)];
    }
}

std::vector<DevCore> TreeTci::chain_cores()
{
    if (!has_net_ || net_root_ != 0) throw Error(T4A_GPU_INVALID_ARGUMENT, "chain_cores: materialise the network around site 0 first");
    const size_t n = local_dims.size();
    const auto edges = graph.edges();
    for (size_t k = 0; k + 1 < n; ++k)
        if (!(edges[k] == TreeEdge(k, k + 1))) throw Error(T4A_GPU_INVALID_ARGUMENT, "chain_cores: the graph is not the linear chain 0-1-...-n");
    std::vector<DevCore> cores(n);
    hipStream_t st = eng.stream();
    for (size_t k = 0; k < n; ++k) {
        const SiteTensor& t = net_[k];
        DevCore& c = cores[k];
        c.s = t.dims[0];
        c.r = (k + 1 < n) ? t.dims[1] : 1;
        c.l = (k > 0) ? t.dims.back() : 1;
        c.buf.reserve(c.size());
        const size_t total = c.size();
        const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 1024);
        hipLaunchKernelGGL(chain_core_kernel, dim3(blocks), dim3(256), 0, st, t.data.get(), (int)c.s, (int)c.r, (int)c.l, c.buf.get());
    }
    T4A_HIP(hipGetLastError());
    T4A_HIP(hipStreamSynchronize(st));
    return cores;
}

// ------------------------------------------------------------------------------------------------ tree contraction
// One workgroup per point.  Sites are visited leaves first; the message of a site towards its parent,
//   m[o] = sum over the incoming bond indices (in_0 fastest) of T[x_site, in_0, .., in_{k-1}, o] * prod_a m_a[in_a],
// is accumulated sequentially per output index, in the order of the CPU restatement.
__global__ void __launch_bounds__(128) tree_evaluate_kernel(const TreeSiteDesc* __restrict__ desc, int n_sites, int msg_total,
                                                            const uint32_t* __restrict__ idx, double* __restrict__ msg,
                                                            double* __restrict__ out)
{
    const size_t p = blockIdx.x;
    double* m = msg + p * (size_t)msg_total;
    const uint32_t* x = idx + p * (size_t)n_sites;
    for (int s = 0; s < n_sites; ++s) {
        const TreeSiteDesc d = desc[s];
        long combos = 1;
        for (int a = 0; a < d.n_in; ++a) combos *= d.in_dim[a];
        const size_t xs = x[d.site];
        for (int o = threadIdx.x; o < d.out_dim; o += blockDim.x) {
            int ctr[TREE_MAX_INCOMING];
            for (int a = 0; a < TREE_MAX_INCOMING; ++a) ctr[a] = 0;
            double acc = 0.0;
            const double* t = d.data + xs + (size_t)d.d * (size_t)combos * (size_t)o;
            for (long c = 0; c < combos; ++c) {
                double term = t[(size_t)d.d * (size_t)c];
                for (int a = 0; a < d.n_in; ++a) term = term * m[d.in_off[a] + ctr[a]];
                acc = acc + term;
                for (int a = 0; a < d.n_in; ++a) {
                    if (++ctr[a] < d.in_dim[a]) break;
                    ctr[a] = 0;
                }
            }
            m[d.msg_off + o] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[p] = m[desc[n_sites - 1].msg_off];
}

void tree_evaluate_launch(const TreeSiteDesc* d_desc, int n_sites, int msg_total, const uint32_t* d_idx, int n_pts,
                          double* d_msg, double* d_out, hipStream_t stream)
{
    if (n_pts <= 0) return;
    hipLaunchKernelGGL(tree_evaluate_kernel, dim3((unsigned)n_pts), dim3(128), 0, stream, d_desc, n_sites, msg_total, d_idx,
                       d_msg, d_out);
}

std::vector<double> TreeTci::evaluate(const uint32_t* idx, size_t n_pts)
{
    if (!has_net_) throw Error(T4A_GPU_INVALID_ARGUMENT, "no materialised network: call materialize first");
    const size_t n = local_dims.size();
    std::vector<double> out(n_pts);
    if (n_pts == 0) return out;
    for (size_t p = 0; p < n_pts; ++p)
        for (size_t s = 0; s < n; ++s)
            if (idx[p * n + s] >= local_dims[s]) throw Error(T4A_GPU_INVALID_ARGUMENT, "index out of bounds");
    hipStream_t st = eng.stream();
    const size_t chunk = 1u << 16;
    d_idx_.reserve(std::min(n_pts, chunk) * n);
    d_msg_.reserve(std::min(n_pts, chunk) * (size_t)msg_total_);
    d_vals_.reserve(std::min(n_pts, chunk));
    for (size_t p0 = 0; p0 < n_pts; p0 += chunk) {
        const size_t np = std::min(chunk, n_pts - p0);
        T4A_HIP(hipMemcpyAsync(d_idx_.get(), idx + p0 * n, np * n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        tree_evaluate_launch(d_desc_.get(), (int)n, msg_total_, d_idx_.get(), (int)np, d_msg_.get(), d_vals_.get(), st);
        T4A_HIP(hipGetLastError());
        T4A_HIP(hipMemcpyAsync(out.data() + p0, d_vals_.get(), np * sizeof(double), hipMemcpyDeviceToHost, st));
        T4A_HIP(hipStreamSynchronize(st));
    }
    return out;
}

} // namespace t4a
