// oracle/t4a_oracle_tree.hpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE (same rules as t4a_oracle.hpp).
// CPU restatement of the tree tensor cross interpolation crate, crates/tensor4all-treetci/src:
//   graph.rs       TreeTciEdge :17-41, TreeTciGraph::new :51-106, subtree_vertices :120-147, subregion_vertices :150-156,
//                  adjacent_edges :159-174, candidate_edges :177-185, distance_edges :188-195, edges :198-206,
//                  neighbors :214-227, bfs_tree :237-261, edge_in_ij_keys :264-289, linear_chain :335-345
//   key.rs         SubtreeKey :1-16 (sorted, de-duplicated site list)
//   state.rs       TreeTCI2::new :66-103, add_global_pivots :110-165, update_pivot_errors :178-184, max_bond_dim :192-199
//   proposer.rs    DefaultProposer :57-88, union_with_history :268-290, pivot_set :292-329, kronecker :331-349
//   update.rs      update_edge :22-115, evaluate_candidate_matrix :143-243
//   optimize.rs    TreeTciOptions :13-76, optimize_with_proposer :95-220
//   materialize.rs to_treetn :17-103, site_tensor_with_parent :105-166, fill_tensor_values :198-243,
//                  cartesian_entries :245-291, central_assignments :293-316
//   globalpivot.rs find_global_pivots :24-172
//   api.rs         crossinterpolate2 :21-96
// The tree network object (tensor4all-treetn) is a plain per-site dense tensor list here; its evaluation contracts
// from the leaves towards the root.  Parity: the pivot selection is the bit-exact rrLU of t4a_oracle.hpp; the
// full-pivot solve of materialize.rs goes through tenferro in the reference ("parity unpinned", tolerance level);
// the global pivot finder uses rand 0.9 StdRng (t4a_oracle_rng.hpp); SimpleProposer / TruncatedDefaultProposer (proposer.rs:90-249,
// :357-409) draw from rand 0.9 SmallRng (xoshiro256++) seeded through std DefaultHasher (SipHash-1-3, zero key): restated from the
// published algorithms in t4a_oracle_rng2.hpp, pinned to published vectors in tests/test_cpu_stdrng.py (the byte layout `Hash` feeds
// the hasher and the seed -> candidate mapping stay unpinned against the Rust binary: no fixture of the reference fixes them).
#pragma once

#include <deque>
#include <map>
#include <set>

#include "t4a_oracle_tt.hpp"
#include "t4a_oracle_rng2.hpp"

namespace t4a_oracle {

using SubtreeKey = std::vector<size_t>; // sorted site list

struct TreeEdge { // graph.rs:17-41
    size_t u = 0, v = 0;
    TreeEdge() = default;
    TreeEdge(size_t a, size_t b) : u(a <= b ? a : b), v(a <= b ? b : a) {}
    bool operator<(const TreeEdge& o) const { return u != o.u ? u < o.u : v < o.v; }
    bool operator==(const TreeEdge& o) const { return u == o.u && v == o.v; }
};

struct TreeGraph {
    size_t n = 0;
    std::set<TreeEdge> edge_set;
    std::vector<std::vector<size_t>> adj;

    TreeGraph() = default;
    TreeGraph(size_t n_sites, const std::vector<TreeEdge>& edges) : n(n_sites), adj(n_sites)
    {
        if (n_sites == 0) throw OracleError(ERR_INVALID_ARGUMENT, "TreeTCI graph must contain at least one site");
        for (const TreeEdge& e : edges) {
            if (e.u == e.v) throw OracleError(ERR_INVALID_ARGUMENT, "self-loops are not allowed in TreeTCI graphs");
            if (e.v >= n_sites) throw OracleError(ERR_INVALID_ARGUMENT, "edge endpoint is out of bounds");
            if (!edge_set.insert(e).second) throw OracleError(ERR_INVALID_ARGUMENT, "duplicate edge");
            adj[e.u].push_back(e.v);
            adj[e.v].push_back(e.u);
        }
        if (edge_set.size() + 1 != n_sites) throw OracleError(ERR_INVALID_ARGUMENT, "TreeTCI graph must be a tree");
        std::vector<char> seen(n_sites, 0);
        std::vector<size_t> stack{0};
        seen[0] = 1;
        size_t count = 1;
        while (!stack.empty()) {
            size_t c = stack.back();
            stack.pop_back();
            for (size_t w : adj[c])
                if (!seen[w]) {
                    seen[w] = 1;
                    ++count;
                    stack.push_back(w);
                }
        }
        if (count != n_sites) throw OracleError(ERR_INVALID_ARGUMENT, "TreeTCI graph must be connected");
    }
    static TreeGraph linear_chain(size_t n_sites)
    {
        if (n_sites == 0) throw OracleError(ERR_INVALID_ARGUMENT, "linear_chain requires at least 1 site");
        std::vector<TreeEdge> e;
        for (size_t i = 0; i + 1 < n_sites; ++i) e.emplace_back(i, i + 1);
        return TreeGraph(n_sites, e);
    }
    bool has_edge(const TreeEdge& e) const { return edge_set.count(e) != 0; }
    std::pair<size_t, size_t> separate_vertices(const TreeEdge& e) const
    {
        if (!has_edge(e)) throw OracleError(ERR_INVALID_ARGUMENT, "edge is not in the graph");
        return {e.u, e.v};
    }
    SubtreeKey subtree_vertices(size_t parent, const std::vector<size_t>& children) const
    {
        if (parent >= n) throw OracleError(ERR_INVALID_ARGUMENT, "parent site is out of bounds");
        std::vector<size_t> sites;
        std::vector<char> seen(n, 0);
        for (size_t child : children) {
            if (child >= n) throw OracleError(ERR_INVALID_ARGUMENT, "child site is out of bounds");
            if (!has_edge(TreeEdge(parent, child))) throw OracleError(ERR_INVALID_ARGUMENT, "sites are not adjacent");
            std::vector<std::pair<size_t, size_t>> stack{{parent, child}};
            while (!stack.empty()) {
                auto pc = stack.back();
                stack.pop_back();
                if (seen[pc.second]) continue;
                seen[pc.second] = 1;
                sites.push_back(pc.second);
                for (size_t w : adj[pc.second])
                    if (w != pc.first) stack.push_back({pc.second, w});
            }
        }
        std::sort(sites.begin(), sites.end());
        sites.erase(std::unique(sites.begin(), sites.end()), sites.end());
        return sites;
    }
    std::pair<SubtreeKey, SubtreeKey> subregion_vertices(const TreeEdge& e) const
    {
        auto uv = separate_vertices(e);
        return {subtree_vertices(uv.second, {uv.first}), subtree_vertices(uv.first, {uv.second})};
    }
    std::vector<TreeEdge> adjacent_edges(size_t site, const std::vector<TreeEdge>& excluded) const
    {
        std::vector<TreeEdge> out;
        if (site >= n) return out;
        for (size_t w : adj[site]) {
            TreeEdge e(site, w);
            if (std::find(excluded.begin(), excluded.end(), e) == excluded.end()) out.push_back(e);
        }
        std::sort(out.begin(), out.end());
        return out;
    }
    std::vector<TreeEdge> candidate_edges(const TreeEdge& e) const
    {
        auto uv = separate_vertices(e);
        std::set<TreeEdge> s;
        for (const auto& x : adjacent_edges(uv.first, {e})) s.insert(x);
        for (const auto& x : adjacent_edges(uv.second, {e})) s.insert(x);
        return std::vector<TreeEdge>(s.begin(), s.end());
    }
    std::map<TreeEdge, size_t> distance_edges(const TreeEdge& e) const
    {
        auto uv = separate_vertices(e);
        std::map<TreeEdge, size_t> dist;
        auto collect = [&](size_t root, size_t blocked) {
            std::deque<std::array<size_t, 3>> q{{root, root, 0}};
            std::vector<char> seen(n, 0);
            seen[blocked] = 1;
            while (!q.empty()) {
                auto cur = q.front();
                q.pop_front();
                if (seen[cur[1]]) continue;
                seen[cur[1]] = 1;
                if (cur[1] != root) dist[TreeEdge(cur[0], cur[1])] = cur[2];
                for (size_t w : adj[cur[1]])
                    if (!seen[w]) q.push_back({cur[1], w, cur[2] + 1});
            }
        };
        collect(uv.first, uv.second);
        collect(uv.second, uv.first);
        dist[e] = 0;
        return dist;
    }
    std::vector<TreeEdge> edges() const { return std::vector<TreeEdge>(edge_set.begin(), edge_set.end()); }
    std::vector<size_t> neighbors(size_t site) const
    {
        if (site >= n) throw OracleError(ERR_INVALID_ARGUMENT, "site is out of bounds");
        std::vector<size_t> r = adj[site];
        std::sort(r.begin(), r.end());
        return r;
    }
    TreeEdge edge_between(size_t a, size_t b) const
    {
        TreeEdge e(a, b);
        separate_vertices(e);
        return e;
    }
    // parents[site] == n for the root
    void bfs_tree(size_t root, std::vector<size_t>& parents, std::vector<size_t>& distances) const
    {
        if (root >= n) throw OracleError(ERR_INVALID_ARGUMENT, "root site is out of bounds");
        parents.assign(n, n);
        distances.assign(n, (size_t)-1);
        std::deque<size_t> q{root};
        distances[root] = 0;
        while (!q.empty()) {
            size_t c = q.front();
            q.pop_front();
            for (size_t w : neighbors(c))
                if (distances[w] == (size_t)-1) {
                    parents[w] = c;
                    distances[w] = distances[c] + 1;
                    q.push_back(w);
                }
        }
    }
    std::vector<SubtreeKey> edge_in_ij_keys(size_t site, const std::vector<TreeEdge>& es) const
    {
        if (site >= n) throw OracleError(ERR_INVALID_ARGUMENT, "site is out of bounds");
        std::vector<SubtreeKey> keys;
        for (const TreeEdge& e : es) {
            auto uv = separate_vertices(e);
            if (uv.first == site)
                keys.push_back(subtree_vertices(uv.first, {uv.second}));
            else if (uv.second == site)
                keys.push_back(subtree_vertices(uv.second, {uv.first}));
            else
                throw OracleError(ERR_INVALID_ARGUMENT, "edge is not adjacent to site");
        }
        return keys;
    }
};

using PivotTable = std::map<SubtreeKey, std::vector<MultiIndex>>;
// batch evaluator: data is (n_sites, n_points) column-major (batch.rs:13-66)
using TreeBatchFn = std::function<std::vector<double>(const std::vector<size_t>& data, size_t n_sites, size_t n_points)>;

inline TreeBatchFn tree_batch_from_scalar(const ScalarFn& f)
{
    return [f](const std::vector<size_t>& data, size_t n_sites, size_t n_points) {
        std::vector<double> out(n_points);
        MultiIndex mi(n_sites);
        for (size_t p = 0; p < n_points; ++p) {
            for (size_t s = 0; s < n_sites; ++s) mi[s] = data[s + n_sites * p];
            out[p] = f(mi);
        }
        return out;
    };
}

struct TreeTCI2 { // state.rs:41-58
    PivotTable ijset;
    std::vector<size_t> local_dims;
    TreeGraph graph;
    std::map<TreeEdge, double> bond_errors;
    std::vector<double> pivot_errors;
    double max_sample_value = 0.0;
    std::vector<PivotTable> ijset_history;
    int proposer = 0;            // 0 DefaultProposer, 1 SimpleProposer, 2 TruncatedDefaultProposer
    uint64_t proposer_seed = 0;  // ::seeded(seed)

    TreeTCI2(const std::vector<size_t>& dims, const TreeGraph& g) : local_dims(dims), graph(g)
    {
        if (!(dims.size() > 1)) throw OracleError(ERR_INVALID_ARGUMENT, "local_dims should have at least 2 elements");
        if (dims.size() != g.n) throw OracleError(ERR_INVALID_ARGUMENT, "local_dims length must match graph site count");
        for (size_t d : dims)
            if (d == 0) throw OracleError(ERR_INVALID_ARGUMENT, "local dimension must be positive");
        for (const TreeEdge& e : g.edges()) bond_errors[e] = 0.0;
    }

    static void push_unique(std::vector<MultiIndex>& cols, const MultiIndex& c)
    {
        for (const auto& x : cols)
            if (x == c) return;
        cols.push_back(c);
    }
    void add_global_pivots(const std::vector<MultiIndex>& pivots)
    {
        const size_t n = local_dims.size();
        for (const auto& p : pivots)
            if (p.size() != n) throw OracleError(ERR_INVALID_ARGUMENT, "each global pivot must contain one index per site");
        for (const auto& p : pivots)
            for (size_t s = 0; s < n; ++s)
                if (p[s] >= local_dims[s]) throw OracleError(ERR_INVALID_ARGUMENT, "global pivot value is out of bounds");
        for (const auto& p : pivots)
            for (const TreeEdge& e : graph.edges()) {
                auto keys = graph.subregion_vertices(e);
                MultiIndex lp, rp;
                for (size_t s : keys.first) lp.push_back(p[s]);
                for (size_t s : keys.second) rp.push_back(p[s]);
                push_unique(ijset[keys.first], lp);
                push_unique(ijset[keys.second], rp);
            }
        SubtreeKey full(n);
        for (size_t s = 0; s < n; ++s) full[s] = s;
        ijset.emplace(full, std::vector<MultiIndex>());
    }
    void flush_pivot_errors() { pivot_errors.clear(); }
    void update_bond_error(const TreeEdge& e, double err) { bond_errors[e] = err; }
    void update_pivot_errors(const std::vector<double>& errs)
    {
        if (pivot_errors.size() < errs.size()) pivot_errors.resize(errs.size(), 0.0);
        for (size_t k = 0; k < errs.size(); ++k) pivot_errors[k] = std::max(pivot_errors[k], errs[k]);
    }
    double max_bond_error() const
    {
        double m = 0.0;
        for (const auto& kv : bond_errors) m = std::max(m, kv.second);
        return m;
    }
    size_t max_bond_dim() const
    {
        size_t m = 0;
        for (const auto& kv : ijset) m = std::max(m, kv.second.size());
        return m;
    }
    const std::vector<MultiIndex>& pivots_of(const SubtreeKey& key) const
    {
        auto it = ijset.find(key);
        if (it == ijset.end()) throw OracleError(ERR_INVALID_ARGUMENT, "missing pivot set for subtree key");
        return it->second;
    }
};

namespace tree_detail {

inline size_t subtree_position(const SubtreeKey& key, size_t site)
{
    auto it = std::find(key.begin(), key.end(), site);
    if (it == key.end()) throw OracleError(ERR_INVALID_ARGUMENT, "site not found in subtree key");
    return (size_t)(it - key.begin());
}

inline std::vector<MultiIndex> union_with_history(const std::vector<MultiIndex>& values, const PivotTable* history,
                                                  const SubtreeKey& key)
{
    std::vector<MultiIndex> unique;
    std::set<MultiIndex> seen;
    for (const auto& c : values)
        if (seen.insert(c).second) unique.push_back(c);
    if (history) {
        auto it = history->find(key);
        if (it != history->end())
            for (const auto& c : it->second)
                if (seen.insert(c).second) unique.push_back(c);
    }
    return unique;
}

inline std::vector<MultiIndex> pivot_set(const TreeTCI2& st, const std::vector<SubtreeKey>& in_keys, const SubtreeKey& out_key)
{
    std::vector<MultiIndex> pivots{MultiIndex(out_key.size(), 0)};
    for (const SubtreeKey& in_key : in_keys) {
        const auto& incoming = st.pivots_of(in_key);
        std::vector<MultiIndex> next;
        next.reserve(pivots.size() * incoming.size());
        for (const auto& base : pivots)
            for (const auto& index : incoming) {
                if (index.size() != in_key.size()) throw OracleError(ERR_INVALID_ARGUMENT, "pivot length mismatch");
                MultiIndex merged = base;
                for (size_t k = 0; k < in_key.size(); ++k) merged[subtree_position(out_key, in_key[k])] = index[k];
                next.push_back(std::move(merged));
            }
        pivots.swap(next);
    }
    return pivots;
}

inline std::vector<MultiIndex> kronecker(const std::vector<MultiIndex>& pivots, size_t site_index, size_t local_dim)
{
    std::vector<MultiIndex> r;
    r.reserve(pivots.size() * local_dim);
    for (const auto& p : pivots)
        for (size_t v = 0; v < local_dim; ++v) {
            MultiIndex c = p;
            c[site_index] = v;
            r.push_back(std::move(c));
        }
    return r;
}

} // namespace tree_detail

// proposer.rs:57-88
inline void default_proposer_candidates(const TreeTCI2& st, const TreeEdge& edge, std::vector<MultiIndex>& icand,
                                        std::vector<MultiIndex>& jcand)
{
    using namespace tree_detail;
    auto pq = st.graph.separate_vertices(edge);
    auto keys = st.graph.subregion_vertices(edge);
    const PivotTable* history = st.ijset_history.empty() ? nullptr : &st.ijset_history.back();
    auto side = [&](size_t vtx, const SubtreeKey& key) {
        auto adjacent = st.graph.adjacent_edges(vtx, {edge});
        auto in_keys = st.graph.edge_in_ij_keys(vtx, adjacent);
        auto pivots = pivot_set(st, in_keys, key);
        auto set = kronecker(pivots, subtree_position(key, vtx), st.local_dims[vtx]);
        return union_with_history(set, history, key);
    };
    icand = side(pq.first, keys.first);
    jcand = side(pq.second, keys.second);
}

// proposer.rs:360-387 rng_for_edge: DefaultHasher (SipHash-1-3, zero key) over seed, tag, edge, history length and the two pivot
// counts, then SmallRng::seed_from_u64 (t4a_oracle_rng2.hpp)
inline OracleSmallRng tree_rng_for_edge(const TreeTCI2& st, const TreeEdge& edge, uint64_t seed, const std::string& tag)
{
    auto keys = st.graph.subregion_vertices(edge);
    auto ncols = [&](const SubtreeKey& k) {
        auto it = st.ijset.find(k);
        return it == st.ijset.end() ? (uint64_t)0 : (uint64_t)it->second.size();
    };
    HashBytes h;
    h.u64(seed);
    h.str(tag);
    h.u64((uint64_t)std::min(edge.u, edge.v)); // TreeTciEdge { u, v } with u <= v (graph.rs:21-35), derived Hash: field order
    h.u64((uint64_t)std::max(edge.u, edge.v));
    h.u64((uint64_t)st.ijset_history.size());
    h.u64(ncols(keys.first));
    h.u64(ncols(keys.second));
    return OracleSmallRng(h.default_hasher_finish());
}

// proposer.rs:389-409 sample_ordered_candidates: `max_size` distinct candidates in their original order
inline std::vector<MultiIndex> tree_sample_ordered(const std::vector<MultiIndex>& cand, size_t max_size, OracleSmallRng& rng)
{
    if (cand.size() <= max_size) return cand;
    std::vector<size_t> idx(cand.size());
    for (size_t k = 0; k < idx.size(); ++k) idx[k] = k;
    rng2_shuffle(idx, rng); // selected_indices.shuffle(rng)
    idx.resize(max_size);
    std::sort(idx.begin(), idx.end());
    std::vector<MultiIndex> out;
    for (size_t k : idx) out.push_back(cand[k]);
    return out;
}

// PivotCandidateProposer::candidates for the proposer selected on the state (proposer.rs:57-249)
inline void tree_candidates(const TreeTCI2& st, const TreeEdge& edge, std::vector<MultiIndex>& icand, std::vector<MultiIndex>& jcand)
{
    if (st.proposer == 0) {
        default_proposer_candidates(st, edge, icand, jcand);
        return;
    }
    auto pq = st.graph.separate_vertices(edge);
    auto keys = st.graph.subregion_vertices(edge);
    const size_t ichi = st.local_dims[pq.first] * st.pivots_of(keys.first).size();
    const size_t jchi = st.local_dims[pq.second] * st.pivots_of(keys.second).size();
    if (st.proposer == 1) { // SimpleProposer :127-160
        OracleSmallRng rng = tree_rng_for_edge(st, edge, st.proposer_seed, "simple");
        auto random = [&](const SubtreeKey& key, size_t size) {
            std::vector<MultiIndex> out;
            for (size_t k = 0; k < size; ++k) {
                MultiIndex c;
                for (size_t site : key) c.push_back(rng.range(st.local_dims[site]));
                out.push_back(c);
            }
            return out;
        };
        const PivotTable* history = st.ijset_history.empty() ? nullptr : &st.ijset_history.back();
        auto iset = random(keys.first, ichi);
        auto jset = random(keys.second, jchi);
        icand = tree_detail::union_with_history(iset, history, keys.first);
        jcand = tree_detail::union_with_history(jset, history, keys.second);
        return;
    }
    if (st.proposer == 2) { // TruncatedDefaultProposer :205-249
        std::vector<MultiIndex> di, dj;
        default_proposer_candidates(st, edge, di, dj);
        OracleSmallRng rng = tree_rng_for_edge(st, edge, st.proposer_seed, "truncated_default");
        icand = tree_sample_ordered(di, ichi, rng);
        jcand = tree_sample_ordered(dj, jchi, rng);
        return;
    }
    throw OracleError(ERR_INVALID_ARGUMENT, "unknown proposer");
}

// update.rs:143-243
inline std::vector<double> evaluate_candidate_matrix(size_t n_sites, const SubtreeKey& left_key,
                                                     const std::vector<MultiIndex>& left, const SubtreeKey& right_key,
                                                     const std::vector<MultiIndex>& right, const std::vector<size_t>& dims,
                                                     const TreeBatchFn& evaluate)
{
    std::vector<char> assigned(n_sites, 0);
    for (const SubtreeKey* key : {&left_key, &right_key})
        for (size_t s : *key) {
            if (s >= n_sites) throw OracleError(ERR_INVALID_ARGUMENT, "site is out of bounds");
            if (assigned[s]) throw OracleError(ERR_INVALID_ARGUMENT, "site was assigned more than once");
            assigned[s] = 1;
        }
    for (char a : assigned)
        if (!a) throw OracleError(ERR_INVALID_ARGUMENT, "global point assembly left some sites unassigned");
    auto check = [&](const std::vector<MultiIndex>& cand, const SubtreeKey& key) {
        for (const auto& c : cand) {
            if (c.size() != key.size()) throw OracleError(ERR_INVALID_ARGUMENT, "candidate length does not match its subtree key");
            for (size_t k = 0; k < key.size(); ++k)
                if (c[k] >= dims[key[k]]) throw OracleError(ERR_INVALID_ARGUMENT, "candidate value out of range");
        }
    };
    check(left, left_key);
    check(right, right_key);
    const size_t n_points = left.size() * right.size();
    if (!(n_sites > 0 && n_points > 0)) throw OracleError(ERR_INVALID_ARGUMENT, "at least one point with one site is required");
    std::vector<size_t> data(n_sites * n_points, 0);
    size_t off = 0;
    for (const auto& r : right)
        for (const auto& l : left) {
            for (size_t k = 0; k < left_key.size(); ++k) data[off + left_key[k]] = l[k];
            for (size_t k = 0; k < right_key.size(); ++k) data[off + right_key[k]] = r[k];
            off += n_sites;
        }
    std::vector<double> values = evaluate(data, n_sites, n_points);
    if (values.size() != n_points) throw OracleError(ERR_INVALID_ARGUMENT, "batch evaluator returned a wrong number of values");
    return values;
}

// update.rs:22-115
inline MatrixLuciFactors tree_update_edge(TreeTCI2& st, const TreeEdge& edge, const TreeBatchFn& evaluate, const RrLUOptions& options)
{
    auto keys = st.graph.subregion_vertices(edge);
    std::vector<MultiIndex> lc, rc;
    tree_candidates(st, edge, lc, rc);
    if (lc.empty() || rc.empty()) throw OracleError(ERR_INVALID_ARGUMENT, "proposer returned empty candidate list");
    std::vector<double> values =
        evaluate_candidate_matrix(st.local_dims.size(), keys.first, lc, keys.second, rc, st.local_dims, evaluate);
    for (double v : values) st.max_sample_value = std::max(st.max_sample_value, std::sqrt(v * v));
    Matrix m(lc.size(), rc.size(), values.data());
    MatrixLuciFactors sel = matrix_luci_factors_from_matrix(m, options);
    std::vector<size_t> rows = sel.row_indices.empty() ? std::vector<size_t>{0} : sel.row_indices;
    std::vector<size_t> cols = sel.col_indices.empty() ? std::vector<size_t>{0} : sel.col_indices;
    std::vector<MultiIndex> li, ri;
    for (size_t r : rows) li.push_back(lc.at(r));
    for (size_t c : cols) ri.push_back(rc.at(c));
    st.ijset[keys.first] = li;
    st.ijset[keys.second] = ri;
    st.update_bond_error(edge, sel.pivot_errors.empty() ? 0.0 : sel.pivot_errors.back());
    st.update_pivot_errors(sel.pivot_errors);
    return sel;
}

// ---------------------------------------------------------------------------------------------
// materialize.rs — per-site dense tensors [d_site, incoming bonds..., bond to the parent]
// ---------------------------------------------------------------------------------------------
struct TreeSiteTensor {
    size_t site = 0;
    std::vector<size_t> dims;          // [d, in_0, ..., in_{k-1}, (out)]
    std::vector<size_t> in_neighbors;  // child site of every incoming bond, in index order
    bool has_parent = false;
    size_t parent = 0;
    std::vector<double> data;          // column-major over dims
};
struct TreeNetwork {
    size_t root = 0;
    std::vector<size_t> order;            // sites sorted by (distance from the root, site)
    std::vector<TreeSiteTensor> tensors;  // indexed by site
    // contraction from the leaves to the root; per site the sum runs over the incoming bond indices in
    // column-major order (in_0 fastest) for every value of the outgoing index
    double evaluate(const MultiIndex& x) const
    {
        std::vector<std::vector<double>> msg(tensors.size());
        for (size_t oi = order.size(); oi-- > 0;) {
            const TreeSiteTensor& t = tensors[order[oi]];
            const size_t d = t.dims[0];
            const size_t k = t.in_neighbors.size();
            size_t combos = 1;
            for (size_t a = 0; a < k; ++a) combos *= t.dims[1 + a];
            const size_t out = t.has_parent ? t.dims[1 + k] : 1;
            std::vector<double> m(out, 0.0);
            std::vector<size_t> ctr(k, 0);
            for (size_t o = 0; o < out; ++o) {
                double acc = 0.0;
                std::fill(ctr.begin(), ctr.end(), 0);
                for (size_t c = 0; c < combos; ++c) {
                    double term = t.data[x[t.site] + d * (c + combos * o)];
                    for (size_t a = 0; a < k; ++a) term = term * msg[t.in_neighbors[a]][ctr[a]];
                    acc = acc + term;
                    for (size_t a = 0; a < k; ++a) {
                        if (++ctr[a] < t.dims[1 + a]) break;
                        ctr[a] = 0;
                    }
                }
                m[o] = acc;
            }
            msg[t.site] = std::move(m);
        }
        return msg[root][0];
    }
};

namespace tree_detail {

// materialize.rs:245-291: combos with key 0 running fastest
inline std::vector<std::vector<MultiIndex>> cartesian_entries(const TreeTCI2& st, const std::vector<SubtreeKey>& keys)
{
    std::vector<std::vector<MultiIndex>> combos;
    if (keys.empty()) {
        combos.push_back({});
        return combos;
    }
    std::vector<const std::vector<MultiIndex>*> sets;
    size_t total = 1;
    for (const auto& k : keys) {
        sets.push_back(&st.pivots_of(k));
        total *= sets.back()->size();
    }
    std::vector<size_t> ctr(keys.size(), 0);
    for (size_t c = 0; c < total; ++c) {
        std::vector<MultiIndex> cur(keys.size());
        for (size_t a = 0; a < keys.size(); ++a) cur[a] = (*sets[a])[ctr[a]];
        combos.push_back(std::move(cur));
        for (size_t a = 0; a < keys.size(); ++a) {
            if (++ctr[a] < sets[a]->size()) break;
            ctr[a] = 0;
        }
    }
    return combos;
}

// materialize.rs:198-243: point order = out combos (outer), in combos, central values (inner)
inline std::vector<double> fill_tensor_values(const TreeTCI2& st, const std::vector<SubtreeKey>& in_keys,
                                              const std::vector<SubtreeKey>& out_keys, const std::vector<size_t>& central_sites,
                                              const TreeBatchFn& evaluate)
{
    const size_t n = st.local_dims.size();
    auto in_combos = cartesian_entries(st, in_keys);
    auto out_combos = cartesian_entries(st, out_keys);
    std::vector<std::vector<std::pair<size_t, size_t>>> central{{}};
    for (size_t site : central_sites) {
        std::vector<std::vector<std::pair<size_t, size_t>>> next;
        for (const auto& combo : central)
            for (size_t v = 0; v < st.local_dims[site]; ++v) {
                auto e = combo;
                e.push_back({site, v});
                next.push_back(std::move(e));
            }
        central.swap(next);
    }
    std::vector<size_t> data;
    size_t n_points = 0;
    const size_t unassigned = (size_t)-1;
    for (const auto& oc : out_combos)
        for (const auto& ic : in_combos)
            for (const auto& cc : central) {
                MultiIndex point(n, unassigned);
                auto put = [&](size_t site, size_t value) {
                    if (site >= n) throw OracleError(ERR_INVALID_ARGUMENT, "site is out of bounds");
                    if (point[site] != unassigned) throw OracleError(ERR_INVALID_ARGUMENT, "site was assigned more than once");
                    point[site] = value;
                };
                for (size_t a = 0; a < in_keys.size(); ++a) {
                    if (in_keys[a].size() != ic[a].size()) throw OracleError(ERR_INVALID_ARGUMENT, "subtree key / multi-index length mismatch");
                    for (size_t k = 0; k < in_keys[a].size(); ++k) put(in_keys[a][k], ic[a][k]);
                }
                for (size_t a = 0; a < out_keys.size(); ++a) {
                    if (out_keys[a].size() != oc[a].size()) throw OracleError(ERR_INVALID_ARGUMENT, "subtree key / multi-index length mismatch");
                    for (size_t k = 0; k < out_keys[a].size(); ++k) put(out_keys[a][k], oc[a][k]);
                }
                for (const auto& sv : cc) put(sv.first, sv.second);
                for (size_t v : point)
                    if (v == unassigned) throw OracleError(ERR_INVALID_ARGUMENT, "global point assembly left some sites unassigned");
                data.insert(data.end(), point.begin(), point.end());
                ++n_points;
            }
    if (n_points == 0) throw OracleError(ERR_INVALID_ARGUMENT, "at least one point is required");
    std::vector<double> values = evaluate(data, n, n_points);
    if (values.size() != n_points) throw OracleError(ERR_INVALID_ARGUMENT, "batch evaluator returned a wrong number of values");
    return values;
}

inline size_t product_pivot_dims(const TreeTCI2& st, const std::vector<SubtreeKey>& keys)
{
    size_t p = 1;
    for (const auto& k : keys) p *= std::max<size_t>(st.pivots_of(k).size(), 1);
    return p;
}

// backend.rs:181-246: T * P = Pi1 through the full-pivot LU of P^T (rrLU with zero tolerances)
inline std::vector<double> solve_right_full_piv_lu(const std::vector<double>& pi1, size_t rows, size_t cols,
                                                   const std::vector<double>& p, size_t p_rows, size_t p_cols)
{
    if (p_rows != p_cols) throw OracleError(ERR_INVALID_ARGUMENT, "full-pivot solve requires a square pivot matrix");
    if (cols != p_rows) throw OracleError(ERR_INVALID_ARGUMENT, "cannot solve T * P = Pi1: shape mismatch");
    const size_t n = p_rows;
    Matrix at = transpose(Matrix(n, n, p.data()));
    Matrix bt = transpose(Matrix(rows, cols, pi1.data())); // n x rows
    RrLUOptions o;
    o.rel_tol = 0.0;
    o.abs_tol = 0.0;
    o.left_orthogonal = true;
    RrLU lu = rrlu(at, o);
    if (lu.npivots() < n) throw OracleError(ERR_SINGULAR, "full_piv_lu_solve failed: singular matrix");
    Matrix l(n, n), u(n, n), br(n, rows);
    for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < n; ++i) {
            l(i, j) = lu.l(i, j);
            u(i, j) = lu.u(i, j);
        }
    for (size_t c = 0; c < rows; ++c)
        for (size_t k = 0; k < n; ++k) br(k, c) = bt(lu.row_permutation[k], c);
    Matrix y = triangular_solve(l, br, true, true, false, true);
    Matrix zc = triangular_solve(u, y, true, false, false, false);
    std::vector<double> x(rows * cols);
    for (size_t c = 0; c < rows; ++c)
        for (size_t k = 0; k < n; ++k) x[c + rows * lu.col_permutation[k]] = zc(k, c);
    return x;
}

} // namespace tree_detail

// materialize.rs:17-103
inline TreeNetwork tree_materialize(const TreeTCI2& st, const TreeBatchFn& evaluate, size_t center_site)
{
    using namespace tree_detail;
    const size_t n = st.graph.n;
    TreeNetwork net;
    net.root = center_site;
    std::vector<size_t> parents, distances;
    st.graph.bfs_tree(center_site, parents, distances);
    std::map<TreeEdge, size_t> bond_dim;
    for (const TreeEdge& e : st.graph.edges()) {
        auto keys = st.graph.subregion_vertices(e);
        auto li = st.ijset.find(keys.first), ri = st.ijset.find(keys.second);
        const size_t lr = li == st.ijset.end() ? 0 : li->second.size();
        const size_t rr = ri == st.ijset.end() ? 0 : ri->second.size();
        if (lr != rr) throw OracleError(ERR_INVALID_ARGUMENT, "bond ranks disagree across edge");
        bond_dim[e] = std::max<size_t>(lr, 1);
    }
    net.order.resize(n);
    for (size_t s = 0; s < n; ++s) net.order[s] = s;
    std::sort(net.order.begin(), net.order.end(), [&](size_t a, size_t b) {
        return distances[a] != distances[b] ? distances[a] < distances[b] : a < b;
    });
    net.tensors.resize(n);
    for (size_t site : net.order) {
        TreeSiteTensor& t = net.tensors[site];
        t.site = site;
        t.has_parent = parents[site] != n;
        std::vector<TreeEdge> out_edges;
        if (t.has_parent) {
            t.parent = parents[site];
            out_edges.push_back(st.graph.edge_between(site, t.parent));
        }
        auto incoming = st.graph.adjacent_edges(site, out_edges);
        auto in_keys = st.graph.edge_in_ij_keys(site, incoming);
        auto out_keys = st.graph.edge_in_ij_keys(site, out_edges);
        if (!t.has_parent) {
            t.data = fill_tensor_values(st, in_keys, out_keys, {site}, evaluate);
        } else {
            std::vector<double> pi1 = fill_tensor_values(st, in_keys, out_keys, {site}, evaluate);
            const size_t rows = st.local_dims[site] * product_pivot_dims(st, in_keys);
            const size_t cols = product_pivot_dims(st, out_keys);
            auto keys = st.graph.subregion_vertices(out_edges[0]);
            const SubtreeKey& side_key =
                std::find(keys.first.begin(), keys.first.end(), site) != keys.first.end() ? keys.first : keys.second;
            std::vector<double> pv = fill_tensor_values(st, {side_key}, out_keys, {}, evaluate);
            const size_t p_rows = st.pivots_of(side_key).size();
            if (p_rows != cols) throw OracleError(ERR_INVALID_ARGUMENT, "pivot matrix is not square");
            bool all_zero = true;
            for (double v : pv)
                if (!(std::sqrt(v * v) < std::numeric_limits<double>::epsilon())) {
                    all_zero = false;
                    break;
                }
            if (all_zero)
                t.data.assign(rows * cols, 0.0);
            else
                t.data = solve_right_full_piv_lu(pi1, rows, cols, pv, p_rows, cols);
        }
        t.dims.push_back(st.local_dims[site]);
        for (const TreeEdge& e : incoming) {
            t.dims.push_back(bond_dim.at(e));
            t.in_neighbors.push_back(e.u == site ? e.v : e.u);
        }
        for (const TreeEdge& e : out_edges) t.dims.push_back(bond_dim.at(e));
    }
    return net;
}

// ---------------------------------------------------------------------------------------------
// optimize.rs
// ---------------------------------------------------------------------------------------------
struct TreeTciOptions { // :13-76
    double tolerance = 1e-8;
    size_t max_iter = 20;
    size_t max_bond_dim = 0; // 0 == None
    bool has_max_bond_dim = false;
    bool normalize_error = true;
    bool enable_global_pivots = true;
    size_t nsearch = 5;
    size_t max_nglobal_pivot = 5;
    double tol_margin_global_search = 10.0;
    bool has_seed = false;
    uint64_t seed = 0;
    void validate() const
    {
        if (!std::isfinite(tolerance) || tolerance < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "tolerance must be finite and nonnegative");
        if (max_iter == 0) throw OracleError(ERR_INVALID_ARGUMENT, "max_iter must be positive");
        if (has_max_bond_dim && max_bond_dim == 0) throw OracleError(ERR_INVALID_ARGUMENT, "max_bond_dim must be positive when specified");
        if (!std::isfinite(tol_margin_global_search) || tol_margin_global_search < 0.0)
            throw OracleError(ERR_INVALID_ARGUMENT, "tol_margin_global_search must be finite and nonnegative");
    }
};

// globalpivot.rs:24-172
inline std::vector<MultiIndex> tree_find_global_pivots(const TreeTCI2& st, const TreeBatchFn& evaluate, size_t nsearch,
                                                       size_t max_nglobal_pivot, double tol_margin, double abs_tol, uint64_t seed)
{
    if (!std::isfinite(abs_tol) || abs_tol < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "global pivot search abs_tol must be finite and nonnegative");
    if (!std::isfinite(tol_margin) || tol_margin < 0.0) throw OracleError(ERR_INVALID_ARGUMENT, "global pivot search tol_margin must be finite and nonnegative");
    if (nsearch == 0 || max_nglobal_pivot == 0) return {};
    const size_t n = st.local_dims.size();
    TreeNetwork net = tree_materialize(st, evaluate, 0);
    OracleStdRng rng(seed); // globalpivot.rs:118
    std::vector<MultiIndex> points;
    for (size_t k = 0; k < nsearch; ++k) {
        MultiIndex start(n);
        for (size_t s = 0; s < n; ++s) start[s] = rng.range(st.local_dims[s]);
        for (size_t s = 0; s < n; ++s)
            for (size_t v = 0; v < st.local_dims[s]; ++v) {
                MultiIndex p = start;
                p[s] = v;
                points.push_back(std::move(p));
            }
    }
    std::vector<size_t> flat;
    for (const auto& p : points) flat.insert(flat.end(), p.begin(), p.end());
    std::vector<double> fv = evaluate(flat, n, points.size());
    if (fv.size() != points.size()) throw OracleError(ERR_INVALID_ARGUMENT, "batch evaluator returned a wrong number of values");
    std::vector<std::pair<double, MultiIndex>> best;
    size_t pi = 0;
    for (size_t k = 0; k < nsearch; ++k) {
        bool have = false;
        double be = 0.0;
        MultiIndex bp;
        for (size_t s = 0; s < n; ++s)
            for (size_t v = 0; v < st.local_dims[s]; ++v) {
                const double re = fv[pi] - net.evaluate(points[pi]);
                const double err = std::sqrt(re * re + 0.0 * 0.0);
                if (!have || err > be) {
                    have = true;
                    be = err;
                    bp = points[pi];
                }
                ++pi;
            }
        if (have && be > abs_tol * tol_margin) best.push_back({be, bp});
    }
    // sort_by total_cmp descending (stable)
    std::stable_sort(best.begin(), best.end(), [](const std::pair<double, MultiIndex>& a, const std::pair<double, MultiIndex>& b) {
        return a.first > b.first;
    });
    std::vector<MultiIndex> pivots;
    for (const auto& b : best) {
        if (std::find(pivots.begin(), pivots.end(), b.second) == pivots.end()) {
            pivots.push_back(b.second);
            if (pivots.size() >= max_nglobal_pivot) break;
        }
    }
    return pivots;
}

struct TreeOptimizeResult {
    std::vector<size_t> ranks;
    std::vector<double> errors;
};

// optimize.rs:95-220 (DefaultProposer, AllEdges visitor)
inline TreeOptimizeResult tree_optimize(TreeTCI2& st, const TreeBatchFn& evaluate, const TreeTciOptions& options)
{
    options.validate();
    TreeOptimizeResult res;
    std::vector<size_t> nglobal;
    const size_t INNER_EDGE_PASSES = 2, NCHECK_HISTORY = 3;
    for (size_t iter = 0; iter < options.max_iter; ++iter) {
        for (size_t pass = 0; pass < INNER_EDGE_PASSES; ++pass) {
            const double scale = options.normalize_error && st.max_sample_value > 0.0 ? st.max_sample_value : 1.0;
            RrLUOptions ko;
            ko.rel_tol = 1e-14;
            ko.abs_tol = options.tolerance * scale;
            ko.max_bond_dim = options.has_max_bond_dim ? options.max_bond_dim : std::numeric_limits<size_t>::max();
            ko.left_orthogonal = true;
            st.ijset_history.push_back(st.ijset);
            st.flush_pivot_errors();
            for (const TreeEdge& e : st.graph.edges()) tree_update_edge(st, e, evaluate, ko);
        }
        res.ranks.push_back(st.max_bond_dim());
        res.errors.push_back(options.normalize_error && st.max_sample_value > 0.0 ? st.max_bond_error() / st.max_sample_value
                                                                                   : st.max_bond_error());
        if (options.enable_global_pivots && iter + 1 < options.max_iter) {
            const double scale = options.normalize_error && st.max_sample_value > 0.0 ? st.max_sample_value : 1.0;
            const uint64_t seed = options.has_seed ? options.seed + (uint64_t)iter : 0x243F6A8885A308D3ull + (uint64_t)iter;
            auto pivots = tree_find_global_pivots(st, evaluate, options.nsearch, options.max_nglobal_pivot,
                                                  options.tol_margin_global_search, options.tolerance * scale, seed);
            st.add_global_pivots(pivots);
            nglobal.push_back(pivots.size());
        } else {
            nglobal.push_back(0);
        }
        if (res.errors.size() >= NCHECK_HISTORY) {
            const size_t m = res.errors.size();
            bool errors_converged = true, no_global = true, saturated = options.has_max_bond_dim;
            size_t min_rank = (size_t)-1;
            for (size_t k = m - NCHECK_HISTORY; k < m; ++k) {
                errors_converged = errors_converged && res.errors[k] < options.tolerance;
                no_global = no_global && nglobal[k] == 0;
                min_rank = std::min(min_rank, res.ranks[k]);
                saturated = saturated && res.ranks[k] >= options.max_bond_dim;
            }
            const bool rank_stable = min_rank == res.ranks.back();
            if ((errors_converged && no_global && rank_stable) || saturated) break;
        }
    }
    return res;
}

// api.rs:21-96
inline TreeOptimizeResult tree_crossinterpolate2(TreeTCI2& st, const TreeBatchFn& evaluate, std::vector<MultiIndex> initial_pivots,
                                                 const TreeTciOptions& options)
{
    options.validate();
    const size_t n = st.local_dims.size();
    if (initial_pivots.empty()) initial_pivots.push_back(MultiIndex(n, 0));
    st.add_global_pivots(initial_pivots);
    std::vector<size_t> flat;
    for (const auto& p : initial_pivots) flat.insert(flat.end(), p.begin(), p.end());
    std::vector<double> vals = evaluate(flat, n, initial_pivots.size());
    if (vals.size() != initial_pivots.size()) throw OracleError(ERR_INVALID_ARGUMENT, "initial evaluator returned a wrong number of values");
    double m = 0.0;
    for (double v : vals) m = std::max(m, std::sqrt(v * v));
    st.max_sample_value = m;
    if (!(st.max_sample_value > 0.0)) throw OracleError(ERR_INVALID_ARGUMENT, "initial pivots must not all evaluate to zero");
    return tree_optimize(st, evaluate, options);
}

} // namespace t4a_oracle
