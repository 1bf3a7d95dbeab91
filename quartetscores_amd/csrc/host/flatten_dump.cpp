// flatten_dump.cpp -- prints what the C++ host hands to the C-ABI for a reference tree and an
// evaluation-tree file (test tool: tests/test_cli.py compares it with the Python host's flattening
// and checks that the multi-threaded ingest is deterministic). No GPU needed.
#include "QuartetScoreComputer.hpp"

#include <cstdlib>
#include <iostream>

using namespace qsh;

template <typename V> static void dump(const char *name, const V &v) {
    std::cout << name;
    for (auto x : v) std::cout << " " << (long long)x;
    std::cout << "\n";
}

int main(int argc, char **argv) {
    if (argc < 3) { std::cerr << "usage: flatten_dump ref.nwk eval.nwk [threads]\n"; return 2; }
    try {
        const std::string refText = slurp(argv[1]);
        NewickReader rr(refText);
        Tree ref;
        if (!rr.next(ref)) throw std::runtime_error("empty reference tree");
        RefFlat rf = flatten_reference(ref);
        std::cout << "names";
        for (auto &n : rf.names) std::cout << " " << n;
        std::cout << "\n";
        dump("parent", rf.parent);
        dump("leaf_node", rf.leaf_node);
        const std::string text = slurp(argv[2]);
        const auto spans = split_trees(text);
        const unsigned threads = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
        BatchFlat b = flatten_parallel(text, spans, 0, spans.size(), rf.name_to_id, threads);
        std::cout << "n_trees " << b.n_trees << "\n";
        dump("leaf_off", b.leaf_off);
        dump("leaf_ids", b.leaf_ids);
        dump("adj_depth", b.adj_depth);
        dump("node_off", b.node_off);
        dump("rng_off", b.rng_off);
        dump("ranges", b.ranges);
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
