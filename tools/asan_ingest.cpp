// sanitizer driver for the host ingest (AddressSanitizer + UBSan; CPU only -- GPU sanitizers are not available on the pool):
// synthesise, decorate, parse, flatten with 1..8 threads; malformed inputs must throw.   make -C tools asan && tools/bin/asan_ingest
#include "ingest.hpp"
#include "synth.hpp"
#include <iostream>
#include <random>
using namespace qsh;
int main() {
    std::mt19937_64 rng(7);
    size_t trees = 0, errors = 0;
    for (uint32_t n : {4u, 5u, 9u, 33u, 130u, 300u}) {
        const std::string refText = synth_random_trees(n, 1, 100 + n, 1);
        NewickReader rr(refText);
        Tree ref;
        if (!rr.next(ref)) return 2;
        const RefFlat rf = flatten_reference(ref);
        for (int kind = 0; kind < 2; ++kind) {
            std::string text = kind == 0 ? synth_random_trees(n, 200, 200 + n, 4) : synth_nni_trees(refText, 200, 300 + n, n / 8.0, 4);
            // decorate: comments (with ';' inside), branch lengths, quoted labels, blank lines, a missing last ';'
            std::string deco;
            for (size_t i = 0; i < text.size(); ++i) {
                const char ch = text[i];
                if (ch == ',' && rng() % 7 == 0) deco += "[c;omment]";
                if (ch == ')' && rng() % 5 == 0) deco += ":0.125";
                deco += ch;
                if (ch == '\n' && rng() % 9 == 0) deco += "\n  \t\n";
            }
            while (!deco.empty() && (deco.back() == '\n' || deco.back() == ';')) deco.pop_back();
            for (const std::string *t : {&text, &deco}) {
                const auto spans = split_trees(*t);
                for (unsigned threads : {1u, 3u, 8u})
                    for (bool ranges : {false, true}) {
                        BatchFlat b = flatten_parallel(*t, spans, 0, spans.size(), rf.name_to_id, threads, ranges);
                        trees += b.n_trees;
                        if (b.n_trees != spans.size() || b.leaf_off.back() != b.leaf_ids.size()) return 3;
                    }
                // sub-ranges
                if (spans.size() > 10) { BatchFlat b = flatten_parallel(*t, spans, 3, 9, rf.name_to_id, 2, true); if (b.n_trees != 6) return 4; }
            }
        }
        // malformed / hostile inputs must throw, not crash
        const char *bad[] = {"(", "((a,b),", "(a,b));", ");", "(a,b,c", "((((((((((", "(t0,t1,(t2,t3)'unterminated", "(t0,t1,[unterminated",
                             "(t0:1e999,t1,t2);", "(t0,t0,t1,t2);", "(nosuch,t1,t2,t3);", ";;;;", "", "\n\n", "(t0,(t1,(t2,(t3))))));", "(,,,);"};
        for (const char *s : bad) {
            const std::string t = s;
            try {
                const auto spans = split_trees(t);
                BatchFlat b = flatten_parallel(t, spans, 0, spans.size(), rf.name_to_id, 2, true);
                trees += b.n_trees;
            } catch (const std::exception &) { ++errors; }
        }
    }
    std::cout << "trees flattened: " << trees << ", malformed inputs rejected: " << errors << std::endl;
    return 0;
}
