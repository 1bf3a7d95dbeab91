// fast_ingest.hpp -- allocation-free parse + flatten of ONE evaluation tree (the inner loop of the ingest).
//
// newick.hpp's Tree keeps a vector of children and two strings per node: ~1500 heap allocations for a 256-taxon
// tree, which is what bounded flatten_parallel (134 us per tree on one thread, and no scaling from 2 to 4 threads:
// the allocator). Here a tree is parsed straight into a parent array + label spans inside reusable scratch
// buffers, the adjacency is a CSR built from the parent array, and the traversal of flatten.hpp's flatten_append
// runs on it unchanged in meaning: same neighbour order ([parent, children...]), same re-rooting at the centre,
// same depth-first leaf order, same arrays -- tests/test_cli.py compares them with the Python host's.
#pragma once

#include "flatten.hpp"
#include "newick.hpp"

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace qsh {

// Taxon name -> lookup id without building a std::string and hashing it through std::unordered_map per leaf (which was
// ~1/3 of the flatten time of a 512-taxon tree): open addressing over FNV-1a hashes of the bytes, the names themselves
// stay in the caller's map (verified with memcmp, so a hash collision can never return a wrong id).
struct NameTable {
    struct Slot { uint64_t h; const std::string *name; uint32_t id; };
    std::vector<Slot> slots;
    uint64_t mask = 0;
    const void *built_for = nullptr;
    size_t built_n = 0;
    static uint64_t hash(const char *p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        return h | 1ull;                                   // 0 marks an empty slot
    }
    void build(const std::unordered_map<std::string, uint32_t> &m) {
        size_t cap = 16;
        while (cap < 4 * m.size()) cap <<= 1;
        slots.assign(cap, Slot{0, nullptr, 0});
        mask = cap - 1;
        for (const auto &kv : m) {
            const uint64_t h = hash(kv.first.data(), kv.first.size());
            size_t i = (size_t)(h & mask);
            while (slots[i].h) i = (i + 1) & mask;
            slots[i] = Slot{h, &kv.first, kv.second};
        }
        built_for = &m; built_n = m.size();
    }
    // id of the name text[p, p + n), or -1
    int64_t find(const char *p, size_t n) const {
        const uint64_t h = hash(p, n);
        for (size_t i = (size_t)(h & mask);; i = (i + 1) & mask) {
            const Slot &sl = slots[i];
            if (!sl.h) return -1;
            if (sl.h == h && sl.name->size() == n && std::memcmp(sl.name->data(), p, n) == 0) return sl.id;
        }
    }
};

struct FlatScratch {
    NameTable names;
    std::vector<uint32_t> seen;           // per lookup id: number of the last tree that held it + 1 (duplicate check without a sort)
    uint32_t tree_no = 0;
    std::vector<int32_t> parent;
    std::vector<uint32_t> lab_b, lab_e;   // label = text[lab_b, lab_e) (empty span = none); quoted labels: see qidx
    std::vector<int32_t> qidx;            // -1, or index into `quoted` (labels that needed unescaping)
    std::vector<std::string> quoted;
    std::vector<uint32_t> adj_off, nchild, fill;
    std::vector<int32_t> adj, dist, du, prev, order, dfs_par, ipar;
    std::vector<uint32_t> start, end, depth;
    struct Frame { int32_t x, par; uint32_t k; };
    std::vector<Frame> st;
    std::vector<uint16_t> tmp;
    std::string key;
};

namespace detail {

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
// character classes of the scanner: 1 = ends an unquoted label or a branch length ( ( ) , : ; [ and white space ), 2 = white space or '['
// (what skip() has to look at). Labels and branch lengths are ~70 % of a Newick file: one table look-up per character.
struct CharClass {
    unsigned char t[256];
    constexpr CharClass() : t() {
        for (int i = 0; i < 256; ++i) t[i] = 0;
        t[(unsigned char)'('] = 1; t[(unsigned char)')'] = 1; t[(unsigned char)','] = 1; t[(unsigned char)':'] = 1; t[(unsigned char)';'] = 1;
        t[(unsigned char)'['] = 3; t[(unsigned char)' '] = 3; t[(unsigned char)'\t'] = 3; t[(unsigned char)'\n'] = 3; t[(unsigned char)'\r'] = 3;
    }
};
static constexpr CharClass kCharClass{};

// Newick text [b, e) -> s.parent / label spans. Same dialect and errors as NewickReader::parse_one.
inline void parse_flat(const std::string &text, size_t b, size_t e, FlatScratch &s) {
    s.parent.clear(); s.lab_b.clear(); s.lab_e.clear(); s.qidx.clear(); s.quoted.clear();
    size_t i = b;
    auto skip = [&]() {
        if (i < e && !(kCharClass.t[(unsigned char)text[i]] & 2)) return;   // (the usual case: nothing to skip)
        while (i < e) {
            const char c = text[i];
            if (is_ws(c)) ++i;
            else if (c == '[') {
                const size_t j = text.find(']', i);
                if (j == std::string::npos || j >= e) throw NewickError("unterminated comment");
                i = j + 1;
            } else break;
        }
    };
    auto add = [&](int32_t par) {
        s.parent.push_back(par); s.lab_b.push_back(0); s.lab_e.push_back(0); s.qidx.push_back(-1);
        return (int32_t)s.parent.size() - 1;
    };
    auto label = [&](int32_t node) {
        skip();
        if (i < e && text[i] == '\'') {
            ++i;
            std::string out;
            while (i < e) {
                if (text[i] == '\'') {
                    if (i + 1 < e && text[i + 1] == '\'') { out.push_back('\''); i += 2; continue; }
                    ++i;
                    break;
                }
                out.push_back(text[i++]);
            }
            s.qidx[node] = (int32_t)s.quoted.size();
            s.quoted.push_back(std::move(out));
            return;
        }
        const size_t j0 = i;
        while (i < e && !kCharClass.t[(unsigned char)text[i]]) ++i;
        s.lab_b[node] = (uint32_t)j0; s.lab_e[node] = (uint32_t)i;
    };
    int32_t cur = add(-1);
    skip();
    if (i < e && text[i] != '(') label(cur);
    while (i < e) {
        skip();
        if (i >= e) break;
        const char c = text[i];
        if (c == '(') {
            cur = add(cur);
            ++i; skip();
            if (i < e && text[i] != '(' && text[i] != ',' && text[i] != ')') label(cur);
        } else if (c == ',') {
            if (s.parent[cur] < 0) throw NewickError("',' outside parentheses");
            cur = add(s.parent[cur]);
            ++i; skip();
            if (i < e && text[i] != '(' && text[i] != ',' && text[i] != ')') label(cur);
        } else if (c == ')') {
            if (s.parent[cur] < 0) throw NewickError("unbalanced ')'");
            cur = s.parent[cur];
            ++i; skip();
            if (i < e && text[i] != '(' && text[i] != ')' && text[i] != ',' && text[i] != ':' && text[i] != ';') label(cur);
        } else if (c == ':') {
            ++i; skip();
            while (i < e && (!kCharClass.t[(unsigned char)text[i]] || text[i] == ':')) ++i;   // (a ':' inside a branch length does not end it)
        } else if (c == ';') {
            ++i;
            break;
        } else {
            throw NewickError("unexpected character '" + std::string(1, c) + "' at offset " + std::to_string(i - b));
        }
    }
    if (cur != 0) throw NewickError("unbalanced '('");
}

} // namespace detail

// Parse text[b, e) and append the tree to `batch`; same result as NewickReader + flatten_append.
inline void parse_flatten_append(const std::string &text, size_t b, size_t e, const std::unordered_map<std::string, uint32_t> &name_to_id,
                                 BatchFlat &batch, FlatScratch &s, bool recentre = true, bool want_ranges = true) {
    detail::parse_flat(text, b, e, s);
    if (s.names.built_for != &name_to_id || s.names.built_n != name_to_id.size()) {
        s.names.build(name_to_id);
        uint32_t top = 0;
        for (const auto &kv : name_to_id) top = std::max(top, kv.second + 1);
        s.seen.assign(top, 0); s.tree_no = 0;
    }
    if (++s.tree_no == 0) { std::fill(s.seen.begin(), s.seen.end(), 0u); s.tree_no = 1; }
    const size_t N = s.parent.size();
    // CSR adjacency, neighbour order = [parent, children in input order]
    s.nchild.assign(N, 0);
    for (size_t v = 1; v < N; ++v) s.nchild[s.parent[v]]++;
    s.adj_off.assign(N + 1, 0);
    for (size_t v = 0; v < N; ++v) s.adj_off[v + 1] = s.adj_off[v] + s.nchild[v] + (s.parent[v] >= 0 ? 1u : 0u);
    s.adj.resize(s.adj_off[N]);
    s.fill.assign(N, 0);
    for (size_t v = 0; v < N; ++v)
        if (s.parent[v] >= 0) { s.adj[s.adj_off[v]] = s.parent[v]; s.fill[v] = 1; }
    for (size_t v = 1; v < N; ++v) { const int32_t p = s.parent[v]; s.adj[s.adj_off[p] + s.fill[p]++] = (int32_t)v; }
    auto deg = [&](int32_t x) { return s.adj_off[x + 1] - s.adj_off[x]; };
    auto nb = [&](int32_t x, uint32_t k) { return s.adj[s.adj_off[x] + k]; };

    int32_t r = 0;
    if (recentre && N > 2) {
        // Middle of a longest path, the node detail::centre finds with its two BFS passes -- without the queues. Nodes are
        // numbered in pre-order (a parent before its children, children in input order), so a BFS from node 0 visits every
        // level in increasing index: its last node u is the LAST deepest node of a linear sweep. The distances from u follow in a
        // second sweep (on the chain u..root directly, elsewhere parent + 1). Every node farthest from u ends a longest path, and
        // all of them share the path's first half from u -- so the element (len + 1) / 2 steps from v does not depend on which
        // farthest v is taken (for an even length it is the tree's centre, for an odd one the end of the central edge on u's side).
        s.dist.resize(N); s.du.assign(N, -1);
        s.dist[0] = 0;
        int32_t u = 0;
        for (size_t x = 1; x < N; ++x) { s.dist[x] = s.dist[s.parent[x]] + 1; if (s.dist[x] >= s.dist[u]) u = (int32_t)x; }
        s.order.clear();                                   // the chain u .. root: order[i] = the node at distance i from u
        for (int32_t x = u; x >= 0; x = s.parent[x]) { s.du[x] = (int32_t)s.order.size(); s.order.push_back(x); }
        int32_t v = u;
        for (size_t x = 0; x < N; ++x) {
            if (s.du[x] < 0) s.du[x] = s.du[s.parent[x]] + 1;
            if (s.du[x] >= s.du[v]) v = (int32_t)x;
        }
        const int32_t len = s.du[v];
        int32_t x = v;
        for (int32_t step = 0; step < (len + 1) / 2; ++step) {
            const int32_t dx = s.du[x];
            x = ((size_t)dx < s.order.size() && s.order[dx] == x) ? s.order[dx - 1] : s.parent[x];   // one step towards u
        }
        r = x;
        if (deg(r) == 1) r = nb(r, 0);
    }
    s.start.assign(N, 0); s.end.assign(N, 0); s.depth.assign(N, 0);
    s.dfs_par.assign(N, -1); s.ipar.assign(N, -1);
    // children of x in the tree rooted at r: the neighbours after its parent, cyclically
    auto nkids = [&](int32_t x) { return deg(x) - (s.ipar[x] >= 0 ? 1u : 0u); };
    auto kid = [&](int32_t x, uint32_t k) {
        if (s.ipar[x] < 0) return nb(x, k);
        uint32_t q = (uint32_t)s.ipar[x] + 1 + k;   // < 2 * deg
        const uint32_t dg = deg(x);
        if (q >= dg) q -= dg;
        return nb(x, q);
    };
    s.st.clear();
    s.st.push_back({r, -1, 0});
    uint32_t L = 0, cur_min = 0;
    while (!s.st.empty()) {
        FlatScratch::Frame &f = s.st.back();   // edited in place; re-read nothing from it after a push_back
        const int32_t x = f.x;
        if (f.k == 0) {
            s.start[x] = L;
            s.dfs_par[x] = f.par;
            if (f.par >= 0) {
                uint32_t q = 0;
                while (nb(x, q) != f.par) ++q;
                s.ipar[x] = (int32_t)q;
            }
            if (nkids(x) == 0) {
                const char *lp = s.qidx[x] >= 0 ? s.quoted[s.qidx[x]].data() : text.data() + s.lab_b[x];
                const size_t ln = s.qidx[x] >= 0 ? s.quoted[s.qidx[x]].size() : (size_t)(s.lab_e[x] - s.lab_b[x]);
                const int64_t id = s.names.find(lp, ln);
                if (id < 0) throw UnknownTaxon("unknown taxon '" + std::string(lp, ln) + "' in evaluation tree " + std::to_string(batch.n_trees));
                if (s.seen[(size_t)id] == s.tree_no) throw std::runtime_error("duplicate taxon in evaluation tree " + std::to_string(batch.n_trees));
                s.seen[(size_t)id] = s.tree_no;
                if (L > 0) batch.adj_depth.push_back((uint16_t)std::min<uint32_t>(cur_min, 0xFFFFu));
                cur_min = 1u << 30;
                batch.leaf_ids.push_back((uint16_t)id);
                ++L;
                s.end[x] = L;
                s.st.pop_back();
                continue;
            }
        }
        if (f.k < nkids(x)) {
            const uint32_t k = f.k++;
            cur_min = std::min(cur_min, s.depth[x]);
            const int32_t c = kid(x, k);
            s.depth[c] = s.depth[x] + 1;
            s.st.push_back({c, x, 0});
        } else {
            s.end[x] = L;
            s.st.pop_back();
        }
    }
    if (L > 0) batch.adj_depth.push_back(0);
    batch.leaf_off.push_back((uint32_t)batch.leaf_ids.size());
    // leaf ranges of the links of the inner nodes: only the scatter kernel reads them
    for (size_t x = 0; want_ranges && x < N && L > 0; ++x) {
        const uint32_t nk = nkids((int32_t)x);
        const size_t nlinks = nk + ((int32_t)x != r ? 1 : 0);
        if (nk == 0 || nlinks < 3) continue;
        if ((int32_t)x != r) { batch.ranges.push_back((uint16_t)(s.end[x] % L)); batch.ranges.push_back((uint16_t)(s.start[x] % L)); }
        for (uint32_t k = 0; k < nk; ++k) {
            const int32_t c = kid((int32_t)x, k);
            batch.ranges.push_back((uint16_t)(s.start[c] % L)); batch.ranges.push_back((uint16_t)(s.end[c] % L));
        }
        batch.rng_off.push_back((uint32_t)(batch.ranges.size() / 2));
    }
    batch.node_off.push_back((uint32_t)(batch.rng_off.size() - 1));
    ++batch.n_trees;
}

} // namespace qsh
