// synth.hpp -- seeded synthetic tree sets for benchmarks and size-independent tests.
//
// The reference ships no data; BASELINE.json's configs are "n taxa, m random eval trees". Two
// distributions, the same two as quartetscores_amd/synth.py (numpy, small cases), generated natively because
// configs[2..4] need 10^4..10^5 trees of 256..1024 taxa:
//   random : uniformly random pairwise joining of a shuffled taxon list until three subtrees remain -> an unrooted
//            binary tree with a trifurcating root holding all n taxa (each slot of a tuple receives about m/3);
//   nni    : the reference tree + k random nearest-neighbour interchanges, k ~ Poisson(n/8) (concentrated counts).
// PRNG: xoshiro256** seeded through splitmix64 from (seed, tree index), so tree t of a set depends only on
// (n, seed, t): any thread count and any m give the same trees. Taxa are named t0..t{n-1}; no branch lengths.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace qsh {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull + 0x1234567ull;
        for (auto &v : s) v = splitmix(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // uniform in [0, k), k >= 1 (Lemire's multiply-shift with rejection)
    uint32_t below(uint32_t k) {
        uint64_t m = (uint64_t)(uint32_t)next() * k;
        uint32_t l = (uint32_t)m;
        if (l < k) {
            const uint32_t t = (0u - k) % k;
            while (l < t) { m = (uint64_t)(uint32_t)next() * k; l = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t poisson(double lam) { // Knuth's product method in chunks (lam up to a few hundred)
        uint32_t k = 0;
        double left = lam;
        while (left > 0) {
            const double step = left > 30.0 ? 30.0 : left;
            const double lim = std::exp(-step);
            double p = unit();
            while (p > lim) { ++k; p *= unit(); }
            left -= step;
        }
        return k;
    }
};

// Rooted node arrays of one tree: children lists; leaves carry a taxon number.
struct SynthTree {
    std::vector<std::vector<int32_t>> kids; // empty for leaves
    std::vector<int32_t> taxon;             // -1 for inner nodes
    int32_t root = -1;
    int32_t add_leaf(int32_t t) { kids.emplace_back(); taxon.push_back(t); return (int32_t)taxon.size() - 1; }
    int32_t add_inner(std::vector<int32_t> k) { kids.push_back(std::move(k)); taxon.push_back(-1); return (int32_t)taxon.size() - 1; }
};

inline void synth_write(const SynthTree &t, std::string &out) {
    struct Fr { int32_t x; size_t k; };
    std::vector<Fr> st;
    st.push_back({t.root, 0});
    while (!st.empty()) {
        Fr &f = st.back();
        const auto &ks = t.kids[f.x];
        if (ks.empty()) { out += 't'; out += std::to_string(t.taxon[f.x]); st.pop_back(); continue; }
        if (f.k == 0) out += '(';
        else if (f.k < ks.size()) out += ',';
        if (f.k == ks.size()) { out += ')'; st.pop_back(); continue; }
        const int32_t c = ks[f.k++];
        st.push_back({c, 0});
    }
    out += ";\n";
}

inline void synth_random_tree(uint32_t n, Rng &rng, std::string &out) {
    if (n < 4) throw std::runtime_error("synth: n >= 4");
    SynthTree t;
    std::vector<int32_t> items(n);
    for (uint32_t i = 0; i < n; ++i) items[i] = t.add_leaf((int32_t)i);
    for (uint32_t i = n - 1; i > 0; --i) std::swap(items[i], items[rng.below(i + 1)]); // Fisher-Yates shuffle
    while (items.size() > 3) {
        const uint32_t k = (uint32_t)items.size();
        uint32_t i = rng.below(k), j = rng.below(k - 1);
        if (j >= i) ++j;                       // uniform unordered pair {i, j}
        if (i > j) std::swap(i, j);
        const int32_t a = items[i], b = items[j];
        items[j] = items.back(); items.pop_back();
        if (i < items.size()) { items[i] = items.back(); items.pop_back(); } // (i < j, so i is still in range)
        items.push_back(t.add_inner({a, b}));
    }
    t.root = t.add_inner(std::vector<int32_t>(items.begin(), items.end()));
    synth_write(t, out);
}

template <typename F> inline std::string synth_parallel(uint64_t m, unsigned threads, F one) {
    threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads ? threads : 1, (m + 63) / 64));
    std::vector<std::string> parts(threads);
    auto work = [&](unsigned w) {
        const uint64_t lo = m * w / threads, hi = m * (w + 1) / threads;
        for (uint64_t i = lo; i < hi; ++i) one(i, parts[w]);
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < threads; ++w) pool.emplace_back(work, w);
        for (auto &th : pool) th.join();
    }
    std::string out;
    size_t total = 0;
    for (auto &p : parts) total += p.size();
    out.reserve(total);
    for (auto &p : parts) out += p;
    return out;
}

inline std::string synth_random_trees(uint32_t n, uint64_t m, uint64_t seed, unsigned threads) {
    return synth_parallel(m, threads, [&](uint64_t i, std::string &out) {
        Rng rng(seed, i);
        synth_random_tree(n, rng, out);
    });
}

// the generator's own dialect: names t<k>, parentheses, commas
inline SynthTree synth_parse(const std::string &s) {
    SynthTree t;
    size_t pos = 0;
    std::vector<std::vector<int32_t>> open;
    int32_t last = -1;
    while (pos < s.size() && s[pos] != ';') {
        const char c = s[pos];
        if (c == '(') { open.emplace_back(); ++pos; }
        else if (c == ',') { ++pos; }
        else if (c == ')') {
            if (open.empty()) throw std::runtime_error("synth: unbalanced reference tree");
            last = t.add_inner(std::move(open.back()));
            open.pop_back();
            if (!open.empty()) open.back().push_back(last);
            ++pos;
        } else if (c == 't') {
            size_t j = pos + 1;
            int32_t v = 0;
            while (j < s.size() && s[j] >= '0' && s[j] <= '9') { v = v * 10 + (s[j] - '0'); ++j; }
            if (j == pos + 1) throw std::runtime_error("synth: bad label in the reference tree");
            const int32_t leaf = t.add_leaf(v);
            if (open.empty()) throw std::runtime_error("synth: leaf outside parentheses");
            open.back().push_back(leaf);
            pos = j;
        } else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') ++pos;
        else throw std::runtime_error("synth: the NNI generator reads only its own dialect (t<k> labels, no lengths)");
    }
    if (!open.empty() || last < 0) throw std::runtime_error("synth: unbalanced reference tree");
    t.root = last;
    return t;
}

// One random NNI: a random internal edge (p, c), a random child g of c and a random sibling s of c swap places.
inline void synth_nni(SynthTree &t, const std::vector<std::pair<int32_t, int32_t>> &edges, Rng &rng) {
    if (edges.empty()) return;
    const auto e = edges[rng.below((uint32_t)edges.size())];
    auto &pk = t.kids[e.first];
    auto &ck = t.kids[e.second];
    std::vector<uint32_t> sib;
    for (uint32_t i = 0; i < pk.size(); ++i) if (pk[i] != e.second) sib.push_back(i);
    const uint32_t si = sib[rng.below((uint32_t)sib.size())], gi = rng.below((uint32_t)ck.size());
    std::swap(pk[si], ck[gi]);
}

inline std::string synth_nni_trees(const std::string &ref_text, uint64_t m, uint64_t seed, double mean_nni, unsigned threads) {
    const SynthTree base = synth_parse(ref_text);
    uint32_t n = 0;
    for (int32_t v : base.taxon) n += v >= 0;
    const double lam = mean_nni >= 0 ? mean_nni : n / 8.0;
    // internal edges = (parent, inner child) pairs; rebuilt after every NNI because the swap re-parents two subtrees
    std::vector<std::pair<int32_t, int32_t>> edges;
    for (int32_t x = 0; x < (int32_t)base.kids.size(); ++x)
        for (int32_t c : base.kids[x]) if (!base.kids[c].empty()) edges.emplace_back(x, c);
    return synth_parallel(m, threads, [&](uint64_t i, std::string &out) {
        Rng rng(seed, i);
        SynthTree t = base;
        const uint32_t k = rng.poisson(lam);
        std::vector<std::pair<int32_t, int32_t>> ed = edges;
        for (uint32_t q = 0; q < k; ++q) {
            synth_nni(t, ed, rng);
            ed.clear();
            for (int32_t x = 0; x < (int32_t)t.kids.size(); ++x)
                for (int32_t c : t.kids[x]) if (!t.kids[c].empty()) ed.emplace_back(x, c);
        }
        synth_write(t, out);
    });
}

} // namespace qsh
