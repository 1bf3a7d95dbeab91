// count_bench.cpp -- A/B harness for kernel work: times qs_count_batch of one or more builds of the C-ABI library on the
// same seeded workload and prints a checksum of random table lookups, so that variants can be compared (speed AND
// result) in one gpurun call without Python start-up.
//   tools/bin/count_bench <taxa> <trees> <count_bits> <reps> lib1.so [lib2.so ...]
// Build: make -C tools   (plain g++; the libraries are dlopen'ed)
#include "../include/quartetscores_hip.h"
#include "../quartetscores_amd/csrc/host/ingest.hpp"
#include "../quartetscores_amd/csrc/host/synth.hpp"

#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace qsh;

#define SYM(name) auto name = (decltype(&::name))dlsym(h, #name); if (!name) { fprintf(stderr, "missing %s\n", #name); return 1; }

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s taxa trees count_bits reps lib.so...\n", argv[0]); return 2; }
    const uint32_t n = (uint32_t)atoi(argv[1]);
    const uint64_t m = (uint64_t)atoll(argv[2]);
    const uint32_t bits = (uint32_t)atoi(argv[3]);
    const int reps = atoi(argv[4]);
    const char *nni = getenv("CB_NNI");
    const unsigned threads = std::max(1u, std::thread::hardware_concurrency());
    std::string refText = synth_random_trees(n, 1, 9000, 1);
    if (getenv("CB_LADDER")) {   // caterpillar reference; with CB_NNI the evaluation trees are the ladder + Poisson(n/8) NNIs
        std::string lad = "(t0,t1)";
        for (uint32_t i = 2; i + 2 < n; ++i) lad = "(" + lad + ",t" + std::to_string(i) + ")";
        refText = "(" + lad + ",t" + std::to_string(n - 2) + ",t" + std::to_string(n - 1) + ");\n";
    }
    const std::string text = nni ? synth_nni_trees(refText, m, 9001, -1.0, threads) : synth_random_trees(n, m, 9001, threads);
    NewickReader rr(refText);
    Tree ref;
    rr.next(ref);
    const RefFlat rf = flatten_reference(ref);
    const auto spans = split_trees(text);
    BatchFlat b = flatten_parallel(text, spans, 0, spans.size(), rf.name_to_id, threads, false);
    uint32_t maxd = 0;
    for (uint16_t d : b.adj_depth) maxd = std::max<uint32_t>(maxd, d);
    printf("workload: %u taxa, %llu trees (%s), u%u table, max LCA depth %u\n", n, (unsigned long long)m, nni ? "nni" : "random", bits, maxd);
    qs_tree_batch hb{};
    hb.n_trees = b.n_trees; hb.leaf_off = b.leaf_off.data(); hb.leaf_ids = b.leaf_ids.data(); hb.adj_depth = b.adj_depth.data();
    // random lookups for the checksum
    std::vector<uint16_t> q;
    Rng rng(77, 1);
    for (int i = 0; i < 20000; ++i) {
        uint16_t v[4];
        const uint32_t top = getenv("CB_DHI") ? (uint32_t)atoi(getenv("CB_DHI")) : n;
        for (int k = 0; k < 4;) { v[k] = (uint16_t)rng.below(top); bool dup = false; for (int j = 0; j < k; ++j) dup |= v[j] == v[k]; if (!dup) ++k; }
        q.insert(q.end(), v, v + 4);
    }
    for (int li = 5; li < argc; ++li) {
        void *h = dlopen(argv[li], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[li], dlerror()); return 1; }
        SYM(qs_create) SYM(qs_destroy) SYM(qs_last_error) SYM(qs_table_alloc) SYM(qs_batch_upload) SYM(qs_batch_free)
        SYM(qs_count_batch) SYM(qs_sync) SYM(qs_last_count_ms) SYM(qs_last_count_launches) SYM(qs_last_count_variant) SYM(qs_lookup)
        SYM(qs_set_tuning)
        qs_ctx *c = nullptr;
        const uint32_t dlo = getenv("CB_DLO") ? (uint32_t)atoi(getenv("CB_DLO")) : 0, dhi = getenv("CB_DHI") ? (uint32_t)atoi(getenv("CB_DHI")) : 0;
        if (qs_create(&c, n, bits, 0, 0, nullptr, dlo, dhi) != QS_OK) { fprintf(stderr, "qs_create: %s\n", qs_last_error(nullptr)); return 1; }
        if (const char *sb = getenv("CB_SLICE_BYTES")) qs_set_tuning(c, QS_TUNE_PANEL_SLICE_BYTES, (uint64_t)atoll(sb));
        if (const char *co = getenv("CB_COOP")) qs_set_tuning(c, QS_TUNE_COOP, (uint64_t)atoll(co));
        if (const char *cp = getenv("CB_CLASS_PCT")) qs_set_tuning(c, QS_TUNE_CLASS_PCT, (uint64_t)atoll(cp));
        if (const char *to = getenv("CB_TILE_ORDER")) if (qs_set_tuning(c, QS_TUNE_TILE_ORDER, (uint64_t)atoll(to)) != QS_OK) { fprintf(stderr, "tile order: %s\n", qs_last_error(c)); return 1; }
        if (qs_table_alloc(c) != QS_OK) { fprintf(stderr, "alloc: %s\n", qs_last_error(c)); return 1; }
        qs_device_batch *db = nullptr;
        if (qs_batch_upload(c, &hb, &db) != QS_OK) { fprintf(stderr, "upload: %s\n", qs_last_error(c)); return 1; }
        // warm-up (clock ramp): about 0.3 s of steps
        auto t0 = std::chrono::steady_clock::now();
        int warm = 0;
        do { qs_count_batch(c, db, QS_ALGO_GATHER | QS_COUNT_OVERWRITE); qs_sync(c); ++warm; }
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.3);
        std::vector<float> cnt, pan;
        int launches = 0;
        for (int r = 0; r < reps; ++r) {
            if (qs_count_batch(c, db, QS_ALGO_GATHER | QS_COUNT_OVERWRITE | QS_COUNT_TIMED) != QS_OK || qs_sync(c) != QS_OK) { fprintf(stderr, "count: %s\n", qs_last_error(c)); return 1; }
            float ms[3];
            qs_last_count_ms(c, ms);
            launches = qs_last_count_launches(c);
            pan.push_back(ms[0]); cnt.push_back(ms[1]);
        }
        std::sort(cnt.begin(), cnt.end()); std::sort(pan.begin(), pan.end());
        std::vector<uint64_t> out(q.size() / 4 * 3);
        qs_lookup(c, q.size() / 4, q.data(), out.data());
        uint64_t sum = 0;
        for (size_t i = 0; i < out.size(); ++i) sum = sum * 1000003ull + out[i];
        auto c4 = [](double x) { return x * (x - 1) * (x - 2) * (x - 3) / 24.0; };
        const double nq = c4(dhi ? dhi : n) - c4(dlo);
        printf("%-44s %-44s count %9.4f ms (min %9.4f, %d launches)  panel %8.4f ms  %.3e q/s  checksum %016llx\n", argv[li], qs_last_count_variant(c),
               cnt[cnt.size() / 2], cnt[0], launches, pan[pan.size() / 2], nq * (double)m / (cnt[cnt.size() / 2] * 1e-3), (unsigned long long)sum);
        fflush(stdout);
        qs_batch_free(c, db);
        qs_destroy(c);
        // the library stays loaded (unloading a HIP code object at run time is not worth the risk)
    }
    return 0;
}
