// host_abi.cpp -- the C++ host's Newick ingest (newick.hpp, flatten.hpp, ingest.hpp) behind a small C interface, so
// that the Python multi-GPU driver (quartetscores_amd/dist_cli.py) reads evaluation files at native speed: every
// rank cuts the file into tree spans once and parses + flattens only its own share with a pool of threads.
// Plain g++ (no HIP); builds quartetscores_amd/lib/libquartetscores_host.so. Host plumbing, not the drop-in
// boundary (that is include/quartetscores_hip.h).
#include "ingest.hpp"

#include "synth.hpp"

#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

using namespace qsh;

namespace {
thread_local std::string g_err;
std::string read_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot read ") + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
} // namespace

struct qsh_batch { BatchFlat b; };

extern "C" {

const char *qsh_last_error(void) { return g_err.c_str(); }

// Flattens the trees [tree_lo, min(tree_hi, m)) of eval_path against the taxa of the reference tree in ref_path.
// *n_trees_total = m (trees in the file). 0 on success, 1 on error (qsh_last_error()).
// want_ranges = 0 skips the per-link leaf ranges (only the scatter kernel reads them).
int qsh_ingest(const char *ref_path, const char *eval_path, uint64_t tree_lo, uint64_t tree_hi, unsigned threads,
               int want_ranges, qsh_batch **out, uint64_t *n_trees_total) {
    try {
        if (!ref_path || !eval_path || !out) throw std::runtime_error("qsh_ingest: NULL argument");
        const std::string refText = read_file(ref_path);
        NewickReader rr(refText);
        Tree ref;
        if (!rr.next(ref)) throw std::runtime_error("empty reference tree file");
        const RefFlat rf = flatten_reference(ref);
        const std::string text = read_file(eval_path);
        const auto spans = split_trees(text);
        if (n_trees_total) *n_trees_total = spans.size();
        const size_t lo = std::min<size_t>(tree_lo, spans.size()), hi = std::max(lo, std::min<size_t>(tree_hi, spans.size()));
        if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
        qsh_batch *b = new qsh_batch();
        try {
            b->b = flatten_parallel(text, spans, lo, hi, rf.name_to_id, threads, want_ranges != 0);
        } catch (...) { delete b; throw; }
        *out = b;
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// The same on Newick text already in memory (ref_text holds one tree; eval_text any number of ';'-terminated trees).
int qsh_ingest_text(const char *ref_text, uint64_t ref_len, const char *eval_text, uint64_t eval_len, uint64_t tree_lo,
                    uint64_t tree_hi, unsigned threads, int want_ranges, qsh_batch **out, uint64_t *n_trees_total) {
    try {
        if (!ref_text || !eval_text || !out) throw std::runtime_error("qsh_ingest_text: NULL argument");
        const std::string refText(ref_text, (size_t)ref_len);
        NewickReader rr(refText);
        Tree ref;
        if (!rr.next(ref)) throw std::runtime_error("empty reference tree");
        const RefFlat rf = flatten_reference(ref);
        const std::string text(eval_text, (size_t)eval_len);
        const auto spans = split_trees(text);
        if (n_trees_total) *n_trees_total = spans.size();
        const size_t lo = std::min<size_t>(tree_lo, spans.size()), hi = std::max(lo, std::min<size_t>(tree_hi, spans.size()));
        if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
        qsh_batch *b = new qsh_batch();
        try {
            b->b = flatten_parallel(text, spans, lo, hi, rf.name_to_id, threads, want_ranges != 0);
        } catch (...) { delete b; throw; }
        *out = b;
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

// Seeded synthetic tree sets (synth.hpp): m trees on taxa t0..t{n-1} as ';'-terminated Newick lines.
// kind 0: uniformly random pairwise joining; kind 1: ref_text + Poisson(mean_nni) random NNIs per tree (mean_nni < 0:
// n / 8). *out_text is malloc'ed (qsh_free_text). Tree t depends only on (seed, t): any thread count gives the same text.
int qsh_synth_trees(uint32_t n, uint64_t m, uint64_t seed, int kind, const char *ref_text, double mean_nni, unsigned threads,
                    char **out_text, uint64_t *out_len) {
    try {
        if (!out_text || !out_len) throw std::runtime_error("qsh_synth_trees: NULL argument");
        if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
        std::string text;
        if (kind == 0) text = synth_random_trees(n, m, seed, threads);
        else if (kind == 1) {
            if (!ref_text) throw std::runtime_error("qsh_synth_trees: kind 1 needs the reference tree");
            text = synth_nni_trees(std::string(ref_text), m, seed, mean_nni, threads);
        } else throw std::runtime_error("qsh_synth_trees: unknown kind");
        char *p = (char *)malloc(text.size() + 1);
        if (!p) throw std::runtime_error("qsh_synth_trees: out of memory");
        memcpy(p, text.data(), text.size());
        p[text.size()] = 0;
        *out_text = p; *out_len = text.size();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}
void qsh_free_text(char *p) { free(p); }

uint32_t qsh_batch_n_trees(const qsh_batch *b) { return b ? b->b.n_trees : 0; }
// which: 0 leaf_off (u32), 1 leaf_ids (u16), 2 adj_depth (u16), 3 node_off (u32), 4 rng_off (u32), 5 ranges (u16)
const void *qsh_batch_array(const qsh_batch *b, int which, uint64_t *n_elems) {
    if (!b) return nullptr;
    const BatchFlat &f = b->b;
    switch (which) {
        case 0: if (n_elems) *n_elems = f.leaf_off.size(); return f.leaf_off.data();
        case 1: if (n_elems) *n_elems = f.leaf_ids.size(); return f.leaf_ids.data();
        case 2: if (n_elems) *n_elems = f.adj_depth.size(); return f.adj_depth.data();
        case 3: if (n_elems) *n_elems = f.node_off.size(); return f.node_off.data();
        case 4: if (n_elems) *n_elems = f.rng_off.size(); return f.rng_off.data();
        case 5: if (n_elems) *n_elems = f.ranges.size(); return f.ranges.data();
        default: return nullptr;
    }
}
void qsh_batch_free(qsh_batch *b) { delete b; }

} // extern "C"
