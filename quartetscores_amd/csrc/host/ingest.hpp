// ingest.hpp -- single-pass, multi-threaded Newick ingest of the evaluation-tree file (C++ host).
//
// The reference parses the evaluation file twice on one thread: once only to count the trees
// (QuartetScores.cpp:23-32) and once to stream them (QuartetCounterLookup.hpp:202-206,235). Here the file
// is read once, cut into per-tree spans by a scan for top-level ';' (which also gives m), and spans are
// parsed + flattened by a pool of threads, batch by batch, while the GPU counts the previous batch
// (qs_count_batch is asynchronous). SURVEY.md 8(f) rank 1.
#pragma once

#include "fast_ingest.hpp"
#include "flatten.hpp"
#include "newick.hpp"

#include <algorithm>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace qsh {

// [begin, end) byte spans of the trees in a Newick text: a tree ends at a ';' outside quotes and comments.
inline std::vector<std::pair<size_t, size_t>> split_trees(const std::string &s) {
    std::vector<std::pair<size_t, size_t>> spans;
    const char *base = s.data();
    const size_t n = s.size();
    size_t start = 0;
    while (start < n) {
        // fast path: the next ';' ends the tree if no quote or comment opens before it (memchr scans)
        const char *semi = static_cast<const char *>(memchr(base + start, ';', n - start));
        const size_t stop = semi ? (size_t)(semi - base) : n;
        if (!memchr(base + start, '\'', stop - start) && !memchr(base + start, '[', stop - start)) {
            size_t i = start;
            while (i < stop && (base[i] == ' ' || base[i] == '\t' || base[i] == '\n' || base[i] == '\r')) ++i;
            if (i < stop) spans.emplace_back(start, semi ? stop + 1 : n);
            start = stop + 1;
            continue;
        }
        // slow path for this tree: a ';' inside quotes or comments does not end it. A quote opens a quoted label only
        // where a label can start -- after '(' ',' ')' or at the start of the tree (white space and comments skipped),
        // as in the parsers (parse_flat / NewickReader::label); an apostrophe inside an unquoted label (O'Brien) is an
        // ordinary character.
        bool in_quote = false, in_comment = false, content = false, closed = false, label_may_start = true;
        size_t i = start;
        for (; i < n; ++i) {
            const char ch = base[i];
            if (in_comment) { if (ch == ']') in_comment = false; continue; }
            if (in_quote) {
                if (ch == '\'') { if (i + 1 < n && base[i + 1] == '\'') ++i; else in_quote = false; } // '' = escaped quote
                continue;
            }
            if (ch == '[') { in_comment = true; continue; }
            if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') continue;
            if (ch == '\'' && label_may_start) { in_quote = true; content = true; label_may_start = false; continue; }
            if (ch == ';') { closed = true; break; }
            content = true;
            label_may_start = (ch == '(' || ch == ',' || ch == ')');
        }
        if (content) spans.emplace_back(start, closed ? i + 1 : n); // the last tree may lack its ';'
        start = i + 1;
    }
    return spans;
}

inline void append_batch(BatchFlat &dst, const BatchFlat &src) {
    const uint32_t leaf0 = (uint32_t)dst.leaf_ids.size();
    const uint32_t node0 = (uint32_t)(dst.rng_off.size() - 1);
    const uint32_t link0 = (uint32_t)(dst.ranges.size() / 2);
    dst.leaf_ids.insert(dst.leaf_ids.end(), src.leaf_ids.begin(), src.leaf_ids.end());
    dst.adj_depth.insert(dst.adj_depth.end(), src.adj_depth.begin(), src.adj_depth.end());
    dst.ranges.insert(dst.ranges.end(), src.ranges.begin(), src.ranges.end());
    for (size_t t = 1; t < src.leaf_off.size(); ++t) dst.leaf_off.push_back(src.leaf_off[t] + leaf0);
    for (size_t t = 1; t < src.node_off.size(); ++t) dst.node_off.push_back(src.node_off[t] + node0);
    for (size_t v = 1; v < src.rng_off.size(); ++v) dst.rng_off.push_back(src.rng_off[v] + link0);
    dst.n_trees += src.n_trees;
}

// Parse + flatten the trees spans[i0..i1) with `threads` workers; the result keeps file order.
// An unknown taxon / syntax error in any tree is rethrown on the caller's thread.
// want_ranges = false skips the per-link leaf ranges (node_off / rng_off / ranges stay all-zero / empty): the gather
// kernels read only leaf_ids and adj_depth.
inline BatchFlat flatten_parallel(const std::string &text, const std::vector<std::pair<size_t, size_t>> &spans, size_t i0,
                                  size_t i1, const std::unordered_map<std::string, uint32_t> &name_to_id, unsigned threads,
                                  bool want_ranges = true) {
    const size_t count = i1 - i0;
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, (count + 63) / 64));
    std::vector<BatchFlat> parts(threads);
    std::vector<std::exception_ptr> errors(threads);
    auto work = [&](unsigned w) {
        try {
            const size_t lo = i0 + count * w / threads, hi = i0 + count * (w + 1) / threads;
            FlatScratch scratch; // parse + flatten without per-node allocations (fast_ingest.hpp)
            BatchFlat b;         // worker-local (the vector headers of adjacent parts[] share cache lines)
            for (size_t i = lo; i < hi; ++i) {
                try {
                    parse_flatten_append(text, spans[i].first, spans[i].second, name_to_id, b, scratch, true, want_ranges);
                } catch (const UnknownTaxon &e) { // report the tree's index in the file, not in the part
                    throw UnknownTaxon(std::string(e.what()) + " (tree " + std::to_string(i) + " of the file)");
                }
            }
            parts[w] = std::move(b);
        } catch (...) { errors[w] = std::current_exception(); }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < threads; ++w) pool.emplace_back(work, w);
        for (auto &th : pool) th.join();
    }
    for (auto &e : errors) if (e) std::rethrow_exception(e);
    if (threads == 1) return std::move(parts[0]);
    BatchFlat out;
    for (auto &p : parts) append_batch(out, p);
    return out;
}

} // namespace qsh
