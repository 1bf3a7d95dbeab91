// flatten.hpp -- trees -> the plain arrays of include/quartetscores_hip.h (C++ host).
//
// Reference tree  -> qs_ref_tree   : lookup ids = depth-first leaf order (QuartetCounterLookup.hpp:252-258)
// Evaluation tree -> one entry of qs_tree_batch: the tour's leaf ids (QuartetCounterLookup.hpp:211-221),
//                    LCA depths of tour-adjacent leaves, and the circular leaf ranges behind every link
//                    of every inner node (subtreeLeafIndices, QuartetCounterLookup.hpp:117-121).
// Evaluation trees are re-rooted at their centre (keeps LCA depths small for the 8-bit panel); counts do
// not depend on the rooting.
#pragma once

#include "newick.hpp"

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace qsh {

struct RefFlat {
    std::vector<int32_t> parent;     // per node
    std::vector<uint32_t> leaf_node; // lookup id -> node
    std::vector<std::string> names;  // lookup id -> taxon
    std::unordered_map<std::string, uint32_t> name_to_id;
};

inline RefFlat flatten_reference(const Tree &t) {
    RefFlat r;
    r.parent = t.parent;
    for (size_t v = 0; v < t.node_count(); ++v) // preorder restricted to leaves = depth-first leaf order
        if (t.is_leaf(v)) {
            if (r.name_to_id.count(t.name[v])) throw std::runtime_error("duplicate taxon name in the reference tree: " + t.name[v]);
            r.name_to_id[t.name[v]] = (uint32_t)r.leaf_node.size();
            r.leaf_node.push_back((uint32_t)v);
            r.names.push_back(t.name[v]);
        }
    return r;
}

struct BatchFlat {
    std::vector<uint32_t> leaf_off{0}, node_off{0}, rng_off{0};
    std::vector<uint16_t> leaf_ids, adj_depth, ranges;
    uint32_t n_trees = 0;
    void clear() { *this = BatchFlat(); }
};

struct UnknownTaxon : std::out_of_range {
    using std::out_of_range::out_of_range;
};

namespace detail {
inline int32_t centre(const std::vector<std::vector<int32_t>> &adj, int32_t start) {
    const size_t N = adj.size();
    std::vector<int32_t> dist(N), prev(N), order;
    auto bfs = [&](int32_t src) {
        std::fill(dist.begin(), dist.end(), -1);
        order.clear();
        dist[src] = 0; prev[src] = -1;
        order.push_back(src);
        for (size_t k = 0; k < order.size(); ++k) {
            int32_t x = order[k];
            for (int32_t y : adj[x]) if (dist[y] < 0) { dist[y] = dist[x] + 1; prev[y] = x; order.push_back(y); }
        }
        return order.back();
    };
    int32_t u = bfs(start);
    int32_t v = bfs(u);
    std::vector<int32_t> path{v};
    while (path.back() != u) path.push_back(prev[path.back()]);
    return path[path.size() / 2];
}
} // namespace detail

// Appends one evaluation tree to the batch. Throws UnknownTaxon for a label the reference lacks
// (the reference program dies there with std::out_of_range, QuartetCounterLookup.hpp:218).
inline void flatten_append(const Tree &t, const std::unordered_map<std::string, uint32_t> &name_to_id, BatchFlat &b,
                           bool recentre = true) {
    const size_t N = t.node_count();
    std::vector<std::vector<int32_t>> adj(N);
    for (size_t v = 0; v < N; ++v) { // neighbour order = [parent, children...]
        if (t.parent[v] >= 0) adj[v].push_back(t.parent[v]);
        for (int32_t c : t.children[v]) adj[v].push_back(c);
    }
    int32_t r = 0;
    if (recentre && N > 2) {
        r = detail::centre(adj, 0);
        if (adj[r].size() == 1) r = adj[r][0];
    }
    std::vector<uint32_t> start(N, 0), end(N, 0), depth(N, 0);
    std::vector<std::vector<int32_t>> kids(N);
    struct Frame { int32_t x, par; size_t k; };
    std::vector<Frame> st;
    st.push_back({r, -1, 0});
    uint32_t L = 0, cur_min = 0;
    const size_t ids0 = b.leaf_ids.size();
    while (!st.empty()) {
        Frame f = st.back();
        st.pop_back();
        if (f.k == 0) {
            start[f.x] = L;
            const auto &nb = adj[f.x];
            if (f.par >= 0) {
                size_t i = std::find(nb.begin(), nb.end(), f.par) - nb.begin();
                kids[f.x].assign(nb.begin() + i + 1, nb.end());
                kids[f.x].insert(kids[f.x].end(), nb.begin(), nb.begin() + i);
            } else kids[f.x] = nb;
            if (kids[f.x].empty()) {
                auto it = name_to_id.find(t.name[f.x]);
                if (it == name_to_id.end()) throw UnknownTaxon("unknown taxon '" + t.name[f.x] + "' in evaluation tree " + std::to_string(b.n_trees));
                if (L > 0) b.adj_depth.push_back((uint16_t)std::min<uint32_t>(cur_min, 0xFFFFu));
                cur_min = 1u << 30;
                b.leaf_ids.push_back((uint16_t)it->second);
                ++L;
                end[f.x] = L;
                continue;
            }
        }
        if (f.k < kids[f.x].size()) {
            st.push_back({f.x, f.par, f.k + 1});
            cur_min = std::min(cur_min, depth[f.x]);
            int32_t c = kids[f.x][f.k];
            depth[c] = depth[f.x] + 1;
            st.push_back({c, f.x, 0});
        } else end[f.x] = L;
    }
    if (L > 0) b.adj_depth.push_back(0);
    // duplicate taxa inside one tree
    {
        std::vector<uint16_t> tmp(b.leaf_ids.begin() + ids0, b.leaf_ids.end());
        std::sort(tmp.begin(), tmp.end());
        if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end())
            throw std::runtime_error("duplicate taxon in evaluation tree " + std::to_string(b.n_trees));
    }
    b.leaf_off.push_back((uint32_t)b.leaf_ids.size());
    for (size_t x = 0; x < N && L > 0; ++x) {
        const size_t nlinks = kids[x].size() + ((int32_t)x != r ? 1 : 0);
        if (kids[x].empty() || nlinks < 3) continue;
        if ((int32_t)x != r) { b.ranges.push_back((uint16_t)(end[x] % L)); b.ranges.push_back((uint16_t)(start[x] % L)); }
        for (int32_t c : kids[x]) { b.ranges.push_back((uint16_t)(start[c] % L)); b.ranges.push_back((uint16_t)(end[c] % L)); }
        b.rng_off.push_back((uint32_t)(b.ranges.size() / 2));
    }
    b.node_off.push_back((uint32_t)(b.rng_off.size() - 1));
    ++b.n_trees;
}

} // namespace qsh
