// QuartetScoreComputer.hpp -- C++ host mirror of the reference's two hot-path classes, above the C-ABI.
//
//   QuartetCounterLookup<CINT>   <->  reference QuartetCounterLookup.hpp:24-53
//   QuartetScoreComputer<CINT>   <->  reference QuartetScoreComputer.hpp:43-80
//
// Same constructor arguments, getters and stdout protocol, so main() reads like
// the reference's (QuartetScores.cpp:115-147). Everything that is O(m*C(n,4)) or O(C(n,4)) happens in
// libquartetscores_hip.so on the GPU; this header flattens trees and forwards.
#pragma once

#include "../../../include/quartetscores_hip.h"
#include "flatten.hpp"
#include "ingest.hpp"
#include "newick.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <future>
#include <thread>
#include <stdexcept>
#include <tuple>
#include <type_traits>

namespace qsh {

inline std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// The evaluation file, read and cut into tree spans ONCE (the reference reads it twice: to count the trees,
// QuartetScores.cpp:23-32, and to stream them). countEvalTrees() fills the cache, the counter consumes it.
struct EvalFile {
    std::string path, text;
    std::vector<std::pair<size_t, size_t>> spans;
};
inline std::shared_ptr<const EvalFile> loadEvalFile(const std::string &evalTreesPath, bool drop = false) {
    static std::shared_ptr<const EvalFile> cache;
    if (drop) { auto last = cache; cache.reset(); return last; }   // the counter is done with the file: release the text
    if (!cache || cache->path != evalTreesPath) {
        auto f = std::make_shared<EvalFile>();
        f->path = evalTreesPath;
        f->text = slurp(evalTreesPath);
        f->spans = split_trees(f->text);
        cache = f;
    }
    return cache;
}
// Count the number of evaluation trees: one scan for top-level ';' (ingest.hpp).
inline size_t countEvalTrees(const std::string &evalTreesPath) { return loadEvalFile(evalTreesPath)->spans.size(); }

struct DeviceOptions {
    int device = 0;
    uint32_t algo = QS_ALGO_AUTO;
    size_t batch_trees = 8192;   // trees per device batch = 256 groups of 32 = one panel slice = one launch of the count kernel per depth
                                 // class (each launch reads and writes the whole table once); batch k+1 is parsed and flattened on the
                                 // host threads while batch k counts
    size_t first_batch_trees = 2048;   // ... and only the FIRST batch's parse is exposed: it is a small one (the device starts after
                                 // ~1/4 of the time; 512 taxa x 10000 trees: counting phase 0.43 -> 0.39 s at -t 8)
    unsigned ingest_threads = 0; // host threads that parse + flatten (0 = hardware concurrency); the CLI's -t
    bool qp_exact64 = false;
    bool root_as_edge = false;   // QS_SCORE_ROOT_AS_EDGE: a degree-2 root as a subdivision of one edge (not the reference's quirk Q5)
    bool savemem_lookups = false; // QS_SCORE_SAVEMEM_LOOKUPS (the CLI's -s): a rooted reference tree ends the run with the
                                 // std::runtime_error the reference's compact table throws (quartet_lookup_table.hpp:79-85)
    std::string reduce = "rccl"; // --gpus N: "rccl" (ncclReduceScatter / ncclAllReduce) or "p2p" (peer access, no communicator: multi_gpu.hpp)
    bool gpus_on_one_device = false; // test hook of --reduce p2p: the N "GPUs" are N contexts on device `device`
    bool comm_overlap = false;   // --gpus N, rccl: count while ncclCommInitAll runs (false: the first launch waits for the communicators)
    std::string load_table, save_table; // count-table persistence
    bool trace = false;          // --trace: time stamps of the counting pipeline on stderr
};

// --trace: "[trace] +12.3 ms  what" relative to the first call (process start for practical purposes)
inline void trace_mark(const DeviceOptions &opt, const char *what) {
    if (!opt.trace) return;
    static const auto t0 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "[trace] +%8.1f ms  %s\n", ms, what);
}

template <typename CINT> class QuartetCounterLookup {
public:
    QuartetCounterLookup(Tree const &refTree, const std::string &evalTreesPath, size_t m, bool savemem,
                         DeviceOptions opt = DeviceOptions())
        : ref_(flatten_reference(refTree)), savemem_(savemem) {
        static_assert(sizeof(CINT) <= 4, "m >= 2^32 evaluation trees are not supported by the GPU table");
        // With few host threads the run is bound by the ingest and ends with the LAST batch's count: small batches then (more
        // passes over the table, hidden under the parsing; 512 taxa x 10000 trees at -t 1: 0.66 s with 8192-tree batches, 0.55 s
        // with small ones). From 4 threads on the device is the bottleneck and every batch is one full panel slice.
        if (threads_of(opt) < 4) opt.batch_trees = std::min<size_t>(opt.batch_trees, 2048);
        const uint32_t bits = sizeof(CINT) <= 2 ? 16 : 32;
        // HIP initialisation + table allocation (~0.1 s) run on a helper thread while this one reads the evaluation
        // file and flattens the first batch
        int rc_create = QS_OK, rc_alloc = QS_OK;
        std::string create_err;
        trace_mark(opt, "counter: start (GPU set-up thread + first batch on the host threads)");
        std::thread gpu_init([&] {
            rc_create = qs_create(&ctx_, (uint32_t)ref_.names.size(), bits, QS_FLAG_NONE, opt.device, nullptr, 0, 0);
            if (rc_create != QS_OK) { create_err = qs_last_error(nullptr); return; }
            trace_mark(opt, "gpu thread: context created (HIP initialised)");
            rc_alloc = qs_table_alloc(ctx_);
            trace_mark(opt, "gpu thread: table allocated");
            // launch order of the count kernel + panel: otherwise built by the first qs_count_batch, in front of its first launch
            if (rc_alloc == QS_OK && opt.load_table.empty() && (opt.algo & 0xFFu) != QS_ALGO_SCATTER) (void)qs_prepare(ctx_, std::min<uint64_t>(m, opt.batch_trees));
            trace_mark(opt, "gpu thread: launch order + panel ready");
        });
        std::shared_ptr<const EvalFile> ef;
        BatchFlat first;
        std::exception_ptr host_err;
        const bool counting = opt.load_table.empty();
        try {
            if (counting) {
                ef = loadEvalFile(evalTreesPath);
                first = flatten_batch(*ef, 0, opt);
                trace_mark(opt, "host: first batch flattened");
            }
        } catch (...) { host_err = std::current_exception(); }
        gpu_init.join();
        if (rc_create != QS_OK) throw std::runtime_error(create_err);
        try {   // the destructor does not run when a constructor throws: release the context here
            if (rc_alloc != QS_OK) fail();
            if (host_err) std::rethrow_exception(host_err);
            countQuartets(ef, std::move(first), m, opt);
        } catch (...) {
            qs_destroy(ctx_);
            ctx_ = nullptr;
            throw;
        }
        ef.reset();
        loadEvalFile(std::string(), true);   // drop the cached evaluation file (tens of MB of Newick text)
        std::cout << "lookup table size in bytes: " << qs_table_bytes(ctx_) << "\n"; // QCL:268-272
    }
    ~QuartetCounterLookup() { qs_destroy(ctx_); }
    QuartetCounterLookup(const QuartetCounterLookup &) = delete;
    QuartetCounterLookup &operator=(const QuartetCounterLookup &) = delete;

    // arguments are reference-tree NODE indices (QuartetCounterLookup.hpp:299-318)
    std::tuple<CINT, CINT, CINT> countQuartetOccurrences(size_t aIdx, size_t bIdx, size_t cIdx, size_t dIdx) const {
        uint16_t ids[4] = {lookup_of(aIdx), lookup_of(bIdx), lookup_of(cIdx), lookup_of(dIdx)};
        uint64_t out[3];
        if (qs_lookup(ctx_, 1, ids, out) != QS_OK) throw std::runtime_error(qs_last_error(ctx_));
        return std::tuple<CINT, CINT, CINT>((CINT)out[0], (CINT)out[1], (CINT)out[2]);
    }
    qs_ctx *context() const { return ctx_; }
    const RefFlat &reference() const { return ref_; }

private:
    RefFlat ref_;
    qs_ctx *ctx_ = nullptr;
    bool savemem_;

    [[noreturn]] void fail() const { throw std::runtime_error(qs_last_error(ctx_)); }
    uint16_t lookup_of(size_t node) const {
        for (size_t i = 0; i < ref_.leaf_node.size(); ++i) if (ref_.leaf_node[i] == node) return (uint16_t)i;
        throw std::out_of_range("not a leaf node");
    }
    // QuartetCounterLookup.hpp:196-238. One pass over the file: spans of trees are parsed + flattened by a
    // thread pool, batch by batch, while the GPU counts the previous batch (qs_count_batch is asynchronous).
    static unsigned threads_of(const DeviceOptions &opt) {
        return opt.ingest_threads ? opt.ingest_threads : std::max(1u, std::thread::hardware_concurrency());
    }
    static bool wants_ranges(const DeviceOptions &opt) { return (opt.algo & 0xFFu) == QS_ALGO_SCATTER; } // the gather kernels do not read them
    // batch boundaries: [0, first_batch_trees), then steps of batch_trees
    static size_t batch_end(size_t i0, size_t n, const DeviceOptions &opt) {
        const size_t first = std::max<size_t>(1, std::min(opt.first_batch_trees ? opt.first_batch_trees : opt.batch_trees, opt.batch_trees));
        size_t i1 = std::min(n, i0 == 0 ? first : i0 + std::max<size_t>(1, opt.batch_trees));
        if (n - i1 < opt.batch_trees / 4) i1 = n;   // no small last batch: every batch costs a pass over the table
        return i1;
    }
    BatchFlat flatten_batch(const EvalFile &ef, size_t i0, const DeviceOptions &opt) const {
        const size_t i1 = batch_end(i0, ef.spans.size(), opt);
        return flatten_parallel(ef.text, ef.spans, i0, i1, ref_.name_to_id, threads_of(opt), wants_ranges(opt));
    }
    void countQuartets(const std::shared_ptr<const EvalFile> &ef, BatchFlat first, size_t m, const DeviceOptions &opt) {
        if (!opt.load_table.empty()) { // resume from a saved table instead of counting
            std::string bytes = slurp(opt.load_table);
            if (bytes.size() != qs_table_bytes(ctx_)) throw std::runtime_error("--load-table: size does not match this reference tree / counter width");
            if (qs_table_upload(ctx_, bytes.data(), bytes.size()) != QS_OK) fail();
            return;
        }
        const auto &spans = ef->spans;
        const bool want_ranges = wants_ranges(opt);
        std::vector<qs_device_batch *> in_flight;
        unsigned progress = 1;
        const float onePercent = (float)m / 100;
        // Batch k + 1 is parsed and flattened (on the host threads) while this thread validates, stages and enqueues batch k:
        // with many small trees (256 taxa x 100000: 25 batches) the two used to alternate and the ingest bound the phase.
        std::future<BatchFlat> ahead;
        auto flatten_ahead = [&](size_t i0) {
            const EvalFile *file = ef.get();
            return std::async(std::launch::async, [this, file, i0, &opt] { return flatten_batch(*file, i0, opt); });
        };
        try {
            for (size_t i0 = 0; i0 < spans.size(); i0 = batch_end(i0, spans.size(), opt)) {
                const size_t i1 = batch_end(i0, spans.size(), opt);
                BatchFlat b = i0 == 0 ? std::move(first) : ahead.get();
                if (i1 < spans.size()) ahead = flatten_ahead(i1);
                qs_tree_batch hb;
                hb.n_trees = b.n_trees; hb.leaf_off = b.leaf_off.data(); hb.leaf_ids = b.leaf_ids.data();
                hb.adj_depth = b.adj_depth.data();
                hb.node_off = want_ranges ? b.node_off.data() : nullptr; hb.rng_off = want_ranges ? b.rng_off.data() : nullptr;
                hb.ranges = b.ranges.data();
                // at most two device batches alive: the one being counted and the one being uploaded. Freeing is cheap
                // (the library keeps the device slab for the next upload and orders its reuse behind the kernels).
                if (in_flight.size() == 2) { qs_batch_free(ctx_, in_flight.front()); in_flight.erase(in_flight.begin()); }
                qs_device_batch *db = nullptr;
                // the batch is copied into pinned staging memory here (`b` may go away); the copy to the device runs on
                // the library's copy stream while the previous batch is still being counted
                if (qs_batch_upload(ctx_, &hb, &db) != QS_OK) fail();
                in_flight.push_back(db);
                if (qs_count_batch(ctx_, db, opt.algo) != QS_OK) fail(); // asynchronous
                trace_mark(opt, "host: batch uploaded, count enqueued");
                while ((float)i1 > progress * onePercent && progress <= 100) { // QCL:230-233
                    std::cout << "Counting quartets... " << progress << "%" << std::endl;
                    progress++;
                }
            }
            {   // the scoring set-up (reference tree on the device, round tables, accumulators, candidate log) while the device
                // still counts: this thread would only wait. Optional -- a failure here shows up again in qs_score.
                qs_ref_tree rt;
                rt.n_nodes = (uint32_t)ref_.parent.size(); rt.n_taxa = (uint32_t)ref_.names.size();
                rt.parent = ref_.parent.data(); rt.leaf_node = ref_.leaf_node.data();
                (void)qs_score_prepare(ctx_, &rt, (uint64_t)m);
                trace_mark(opt, "host: scoring set-up done behind the enqueued counts");
            }
            if (qs_sync(ctx_) != QS_OK) fail();
            trace_mark(opt, "host: all counts done (device synchronised)");
        } catch (...) {
            if (ahead.valid()) { try { (void)ahead.get(); } catch (...) {} }   // (the worker still reads the evaluation file)
            (void)qs_sync(ctx_);
            for (auto *db : in_flight) qs_batch_free(ctx_, db);
            throw;
        }
        for (auto *db : in_flight) qs_batch_free(ctx_, db);
        if (!opt.save_table.empty()) {
            std::string bytes(qs_table_bytes(ctx_), '\0');
            if (qs_table_download(ctx_, &bytes[0], bytes.size()) != QS_OK) fail();
            std::ofstream f(opt.save_table, std::ios::binary);
            f.write(bytes.data(), (std::streamsize)bytes.size());
            if (!f) throw std::runtime_error("cannot write " + opt.save_table);
        }
    }
};

    // rank -> sorted ids (rank = C(s3,4)+C(s2,3)+C(s1,2)+s0)
inline void raw_unrank(uint64_t r, uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3) {
        auto c4 = [](uint64_t x) { return x < 4 ? 0 : x * (x - 1) * (x - 2) * (x - 3) / 24; };
        auto c3 = [](uint64_t x) { return x < 3 ? 0 : x * (x - 1) * (x - 2) / 6; };
        auto c2 = [](uint64_t x) { return x * (x - 1) / 2; };
        uint64_t d = 3; while (c4(d + 1) <= r) ++d;
        r -= c4(d);
        uint64_t c = 2; while (c3(c + 1) <= r) ++c;
        r -= c3(c);
        uint64_t b = 1; while (c2(b + 1) <= r) ++b;
        r -= c2(b);
        s0 = (uint32_t)r; s1 = (uint32_t)b; s2 = (uint32_t)c; s3 = (uint32_t)d;
    }

    // QuartetScoreComputer.hpp:135-159 with the host libm
inline double raw_log_score(size_t q1, size_t q2, size_t q3) {
        if (q1 == 0 && q2 == 0 && q3 == 0) return 0;
        size_t sum = q1 + q2 + q3;
        double p1 = (double)q1 / sum, p2 = (double)q2 / sum, p3 = (double)q3 / sum;
        double qic = 1;
        if (p1 != 0) qic += p1 * std::log(p1) / std::log(3);
        if (p2 != 0) qic += p2 * std::log(p2) / std::log(3);
        if (p3 != 0) qic += p3 * std::log(p3) / std::log(3);
        return (q1 < q2 || q1 < q3) ? qic * -1 : qic;
    }


    // QuartetScoreComputer.hpp:623-690: "(a,b|c,d): qic" per quartet resolved in the reference tree.
    // The GPU classifies and looks up a chunk of ranks (qs_raw_qic); the host formats the chunk with all
    // ingest threads into per-thread buffers and writes them in order (the reference
    // formats C(n,4) lines on one thread). Line order: the reference's own -- four nested loops over its Euler-tour
    // leaves = lexicographic in the sorted lookup ids (:626-630; qs_raw_qic_lex) -- or, with raw_rank_order, the
    // table's rank order (coalesced table reads; same set of lines).
inline void print_raw_qic_scores(qs_ctx *ctx, const RefFlat &rf, Tree const &refTree, const std::string &rawPath, unsigned raw_threads, bool raw_rank_order) {
        std::ofstream outfile(rawPath, std::ios::binary);
        qs_ref_tree rt;
        rt.n_nodes = (uint32_t)refTree.node_count(); rt.n_taxa = (uint32_t)rf.names.size();
        rt.parent = rf.parent.data(); rt.leaf_node = rf.leaf_node.data();
        const uint64_t total = qs_table_tuples(ctx), chunk = 1u << 22;
        std::vector<uint8_t> topo(chunk);
        std::vector<uint64_t> q(chunk * 3);
        const unsigned threads = std::max(1u, raw_threads ? raw_threads : std::thread::hardware_concurrency());
        std::vector<std::string> bufs(threads);
        for (uint64_t r = 0; r < total; r += chunk) {
            const uint64_t nq = std::min(chunk, total - r);
            if ((raw_rank_order ? qs_raw_qic(ctx, &rt, r, nq, topo.data(), q.data()) : qs_raw_qic_lex(ctx, &rt, r, nq, topo.data(), q.data())) != QS_OK)
                throw std::runtime_error(qs_last_error(ctx));
            const uint32_t n = rt.n_taxa;
            auto work = [&](unsigned w) {
                const uint64_t lo = nq * w / threads, hi = nq * (w + 1) / threads;
                std::string &out = bufs[w];
                out.clear();
                uint32_t s0, s1, s2, s3;
                if (raw_rank_order) raw_unrank(r + lo, s0, s1, s2, s3);
                else {   // lexicographic index -> ids through the mirrored set's rank (see qs_raw_qic_lex)
                    uint32_t m0, m1, m2, m3;
                    raw_unrank(total - 1 - (r + lo), m0, m1, m2, m3);
                    s0 = n - 1 - m3; s1 = n - 1 - m2; s2 = n - 1 - m1; s3 = n - 1 - m0;
                }
                char num[64];
                for (uint64_t i = lo; i < hi; ++i) {
                    if (topo[i] != 255) {
                        const std::string &A = rf.names[s0], &B = rf.names[s1], &C = rf.names[s2], &D = rf.names[s3];
                        // operator<<(double) with default precision == "%g"
                        snprintf(num, sizeof num, "%g", raw_log_score(q[3 * i], q[3 * i + 1], q[3 * i + 2]));
                        out += '(';
                        if (topo[i] == 0) { out += A; out += ','; out += B; out += '|'; out += C; out += ','; out += D; }
                        else { out += A; out += ','; out += D; out += '|'; out += B; out += ','; out += C; }
                        out += "): "; out += num; out += '\n';
                    }
                    if (raw_rank_order) { if (++s0 == s1) { s0 = 0; if (++s1 == s2) { s1 = 1; if (++s2 == s3) { s2 = 2; ++s3; } } } } // next rank
                    else if (++s3 == n) { if (++s2 == n - 1) { if (++s1 == n - 2) { ++s0; s1 = s0 + 1; } s2 = s1 + 1; } s3 = s2 + 1; } // next 4-subset in lexicographic order
                }
            };
            if (threads == 1) work(0);
            else {
                std::vector<std::thread> pool;
                for (unsigned w = 0; w < threads; ++w) pool.emplace_back(work, w);
                for (auto &th : pool) th.join();
            }
            for (auto &b : bufs) outfile.write(b.data(), (std::streamsize)b.size());
        }
        outfile.close();
    }
    // Binary sidecar of the -q dump: no text formatting, 9 bytes per quartet instead of ~35.
    //   char magic[8] = "QSQIC01"; u32 n_taxa; u32 reserved; u64 n_quartets;
    //   n_taxa x { u32 length; bytes }            taxon names in lookup-id order
    //   u8  topo[n_quartets]                      per rank (rank = C(s3,4)+C(s2,3)+C(s1,2)+s0 of the sorted lookup ids):
    //                                             0 = s0 s1 | s2 s3, 2 = s0 s3 | s1 s2, 255 = not resolved in the reference
    //   f64 qic[n_quartets]                       raw QIC of the reference topology (NaN where topo = 255)
inline void print_raw_qic_binary(qs_ctx *ctx, const RefFlat &rf, Tree const &refTree, const std::string &path) {
        std::ofstream out(path, std::ios::binary);
        if (!out) throw std::runtime_error("cannot write " + path);
        qs_ref_tree rt;
        rt.n_nodes = (uint32_t)refTree.node_count(); rt.n_taxa = (uint32_t)rf.names.size();
        rt.parent = rf.parent.data(); rt.leaf_node = rf.leaf_node.data();
        const uint64_t total = qs_table_tuples(ctx), chunk = 1u << 22;
        const char magic[8] = {'Q', 'S', 'Q', 'I', 'C', '0', '1', 0};
        const uint32_t n32 = rt.n_taxa, zero = 0;
        out.write(magic, 8); out.write((const char *)&n32, 4); out.write((const char *)&zero, 4); out.write((const char *)&total, 8);
        for (const std::string &nm : rf.names) { const uint32_t len = (uint32_t)nm.size(); out.write((const char *)&len, 4); out.write(nm.data(), len); }
        const std::streamoff topo_at = out.tellp(), qic_at = topo_at + (std::streamoff)total;
        std::vector<uint8_t> topo(chunk);
        std::vector<uint64_t> q(chunk * 3);
        std::vector<double> qic(chunk);
        for (uint64_t r = 0; r < total; r += chunk) {
            const uint64_t nq = std::min(chunk, total - r);
            if (qs_raw_qic(ctx, &rt, r, nq, topo.data(), q.data()) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
            for (uint64_t i = 0; i < nq; ++i)
                qic[i] = topo[i] == 255 ? std::numeric_limits<double>::quiet_NaN() : raw_log_score(q[3 * i], q[3 * i + 1], q[3 * i + 2]);
            out.seekp(topo_at + (std::streamoff)r); out.write((const char *)topo.data(), (std::streamsize)nq);
            out.seekp(qic_at + (std::streamoff)(r * 8)); out.write((const char *)qic.data(), (std::streamsize)(nq * 8));
        }
        if (!out) throw std::runtime_error("error writing " + path);
    }

template <typename CINT> class QuartetScoreComputer {
public:
    QuartetScoreComputer(Tree const &refTree, const std::string &evalTreesPath, size_t m, bool verboseOutput,
                         bool enforceSmallMem, DeviceOptions opt = DeviceOptions())
        : referenceTree(refTree), verbose(verboseOutput) {
        std::cout << "There are " << m << " evaluation trees.\n";
        std::cout << "Building subtree informations for reference tree..." << std::endl;
        const size_t n = refTree.leaf_count();
        std::cout << "Finished precomputing subtree informations in reference tree.\n";
        std::cout << "The reference tree has " << n << " taxa.\n";
        // memory estimate (QuartetScoreComputer.hpp:724-731); the GPU always uses the compact table
        const size_t memoryLookupFast = n * n * n * n * sizeof(CINT);
        const size_t memoryLookup = (n * (n - 1) * (n - 2) * (n - 3) / 24) * 3 * sizeof(CINT) + sizeof(size_t);
        std::cout << "Estimated memory usages (in bytes):" << std::endl;
        std::cout << "  Runtime-efficient Lookup table: " << memoryLookupFast << std::endl;
        std::cout << "  Memory-efficient Lookup table: " << memoryLookup << std::endl;
        auto begin = std::chrono::steady_clock::now();
        std::cout << "Using memory-efficient Lookup table\n"; // C(n,4)x3 in HBM
        quartetCounterLookup.reset(new QuartetCounterLookup<CINT>(refTree, evalTreesPath, m, enforceSmallMem, opt));
        auto end = std::chrono::steady_clock::now();
        std::cout << "Finished counting quartets.\n";
        std::cout << "It took: " << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()
                  << " microseconds." << std::endl;
        begin = std::chrono::steady_clock::now();
        const RefFlat &rf = quartetCounterLookup->reference();
        qs_ref_tree rt;
        rt.n_nodes = (uint32_t)refTree.node_count(); rt.n_taxa = (uint32_t)n;
        rt.parent = rf.parent.data(); rt.leaf_node = rf.leaf_node.data();
        std::vector<double> lq(rt.n_nodes), qp(rt.n_nodes), eqp(rt.n_nodes);
        int bif = 0;
        if (qs_score(quartetCounterLookup->context(), &rt,
                     (opt.qp_exact64 ? QS_SCORE_QP_EXACT64 : QS_SCORE_QP_WRAP32) | (opt.root_as_edge ? QS_SCORE_ROOT_AS_EDGE : 0u) |
                         ((enforceSmallMem || opt.savemem_lookups) ? QS_SCORE_SAVEMEM_LOOKUPS : 0u),
                     lq.data(), qp.data(), eqp.data(), &bif) != QS_OK)
            throw std::runtime_error(qs_last_error(quartetCounterLookup->context()));
        if (opt.trace) {
            float ph[6] = {0, 0, 0, 0, 0, 0};
            (void)qs_last_score_ms(quartetCounterLookup->context(), ph);
            std::fprintf(stderr, "[trace] qs_score %.2f ms: set-up %.2f, pass 1 %.2f, pass 2 / log filter %.2f, wait + copies %.2f, host finish %.2f; log %llu records\n",
                         ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], (unsigned long long)qs_last_score_log(quartetCounterLookup->context()));
        }
        // edge e = edge above node e+1 (preorder)
        LQICScores.assign(lq.begin() + 1, lq.end());
        if (bif) {
            std::cout << "The reference tree is bifurcating.\n";
            QPICScores.assign(qp.begin() + 1, qp.end());
            EQPICScores.assign(eqp.begin() + 1, eqp.end());
        } else {
            std::cout << "The reference tree is multifurcating.\n";
        }
        end = std::chrono::steady_clock::now();
        std::cout << "Finished computing scores.\n";
        std::cout << "It took: " << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()
                  << " microseconds." << std::endl;
    }

    std::vector<double> getLQICScores() { return LQICScores; }
    std::vector<double> getQPICScores() { return QPICScores; }
    std::vector<double> getEQPICScores() { return EQPICScores; }

    // -q dump and its binary sidecar (print_raw_qic_scores / print_raw_qic_binary above)
    bool raw_rank_order = false;
    void printRawQICScores(Tree const &refTree, const std::string &rawPath) {
        print_raw_qic_scores(quartetCounterLookup->context(), quartetCounterLookup->reference(), refTree, rawPath, raw_threads, raw_rank_order);
    }
    void printRawQICBinary(Tree const &refTree, const std::string &path) {
        print_raw_qic_binary(quartetCounterLookup->context(), quartetCounterLookup->reference(), refTree, path);
    }
    unsigned raw_threads = 0; // threads that format the -q file (0 = hardware concurrency)

private:
    Tree referenceTree;
    bool verbose;
    std::vector<double> LQICScores, QPICScores, EQPICScores;
    std::unique_ptr<QuartetCounterLookup<CINT>> quartetCounterLookup;
};

} // namespace qsh
