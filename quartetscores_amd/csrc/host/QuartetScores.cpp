// QuartetScores.cpp -- command-line driver of the MI355X engine; keeps the reference's CLI surface
// (QuartetScores.cpp:48-79: -r -e -o required, -q -t -v -s optional) and stdout protocol.
//
//   QuartetScores -r ref.nwk -e eval.nwk -o out.nwk [-q raw.txt] [-t N] [-v] [-s]
//                 [--device N] [--algo gather|scatter] [--exact-qp] [--qic-binary raw.bin]
//
// -t sets the number of host threads that parse + flatten the evaluation trees (the reference's OpenMP
// threads counted quartets; here that happens on the GPU). -s/--savemem does not change the table: the GPU table
// is always the compact C(n,4)x3 layout with semantic (1x) counts. It selects the reference's compact-table lookups
// where those differ observably: with a ROOTED reference tree the reference's table throws (quartet_lookup_table.hpp:79-85)
// and the run ends with that error, here as there (QS_SCORE_SAVEMEM_LOOKUPS).
// --save-table / --load-table write / read the raw count table (resume without recounting).
#include "QuartetScoreComputer.hpp"
#include "multi_gpu.hpp"
#include "table_shards.hpp"

#include <cerrno>
#include <cstdlib>
#include <unistd.h>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

using namespace qsh;

namespace {

bool fast_exit_forced = false;
// true when something in the environment says "a tool rides along that flushes at exit"
bool wrapped_by_tool() {
    const char *pre = std::getenv("LD_PRELOAD");
    if (pre && *pre) return true;
    if (std::getenv("HSA_TOOLS_LIB")) return true;
    for (char **e = ::environ; e && *e; ++e) {
        const std::string kv = *e;
        if (kv.rfind("ROCP_", 0) == 0 || kv.rfind("ROCPROFILER_", 0) == 0 || kv.rfind("ROCTRACER_", 0) == 0 || kv.rfind("ASAN_OPTIONS", 0) == 0 ||
            kv.rfind("LLVM_PROFILE_FILE", 0) == 0 || kv.rfind("GCOV_PREFIX", 0) == 0) return true;
    }
    return false;
}

struct Args {
    std::string ref, eval, out, raw, raw_bin;
    size_t threads = 0;
    bool verbose = false, savemem = false, raw_rank_order = false, fail_fast = false, clean_exit = false;
    int table_shards = -1;   // -1 = off (the whole table on the device); 0 = as many as the device's free memory asks for; K = K shards, one after the other (table_shards.hpp)
    int spill = 0;           // ShardedTableQuartetScoreComputer::Spill
    int gpus = 0;   // 0 = the single-GPU path; N >= 1 = N GPUs of this node: tree-sharded + one table collective (multi_gpu.hpp) or table-sharded (table_shards.hpp)
    std::string mode = "auto";   // --mode with --gpus N: tree | table | auto (the model of table_shards.hpp prefer_table_shards)
    DeviceOptions dev;
};

void usage(std::ostream &os) {
    os << "USAGE:\n   QuartetScores  [-s] [-v] [-t <uint>] [-q <string>] -o <string> -e <string> -r <string> [--version] [-h]\n"
          "   -r, --ref      Path to the reference tree\n"
          "   -e, --eval     Path to the evaluation trees\n"
          "   -o, --output   Path to the annotated newick output file (for lqic/qpic/eqpic scores)\n"
          "   -q, --qic      Path to the file where to write the raw QIC scores for each quartet\n"
          "   -t, --threads  Maximum number of host threads for parsing the evaluation trees (0 = all)\n"
          "   -v, --verbose  Verbose mode\n"
          "   -s, --savemem  Consume less memory (the GPU table is always the compact one; with a ROOTED reference tree the run\n"
          "                  ends with the reference's own std::runtime_error, see --root-as-edge; a note says so before the\n"
          "                  counting starts, --fail-fast ends the run there)\n"
          "   --device N     HIP device ordinal (default 0)\n"
          "   --gpus N       count on N GPUs of this node (devices --device .. --device+N-1)\n"
          "   --mode M       with --gpus: tree = the evaluation trees are split over the GPUs and the count tables combined with one\n"
          "                  reduction over xGMI; table = every GPU counts ALL trees into its shard of the table (by largest taxon\n"
          "                  id): no table collective, no communicator; auto (default) = table unless -q / --qic-binary need the\n"
          "                  whole table on one device or the table collective is the cheaper of the two (many trees, small table)\n"
          "   --reduce R     with --gpus: rccl (default: one RCCL reduce-scatter / all-reduce) | p2p (this one process maps its\n"
          "                  peers' memory and every GPU sums its chunk with plain loads: no communicator to create)\n"
          "   --comm-overlap 0|1  with --gpus, rccl: count while the communicators are being created (default 0: the first launch waits for them)\n"
          "   --algo A       gather (default) | scatter\n"
          "   --exact-qp     64-bit QP sums instead of the reference's 32-bit wrap\n"
          "   --qic-rank-order  -q lines in the order of the count table instead of the reference's loop order\n"
          "   --trace        time stamps of the counting pipeline on stderr\n"
          "   --clean-exit   tear the HIP runtime down before exiting (default: exit right after the output is written,\n"
          "                  unless LD_PRELOAD or a profiler's ROCP_* / ROCPROFILER_* / HSA_TOOLS_LIB environment is set)\n"
          "   --fast-exit    exit right after the output is written even under such a wrapper\n"
          "   --root-as-edge rooted reference tree: score the two root edges as one internode (the reference's\n"
          "                  own handling of a degree-2 root is the default)\n"
          "   --table-shards K  the count table in K shards by largest taxon id (0 = as many as the free device memory asks for):\n"
          "                  alone: a table larger than the device's memory passes through ONE GPU shard by shard;\n"
          "                  with --gpus N: shard s lives on GPU s mod N, every GPU counts all trees into its shard(s), no\n"
          "                  table collective (1024 taxa, 273 GB: --gpus 8 --table-shards 8). --gpus N alone switches to this\n"
          "                  mode by itself when the table does not fit one GPU\n"
          "   --spill M      with --table-shards: host (keep finished shards in host memory) | recount (count every shard a\n"
          "                  second time for the second scoring pass) | auto (host if it fits MemAvailable; default)\n"
          "   --save-table F write the count table to F after counting\n"
          "   --load-table F read the count table from F instead of counting (-e is still needed for m)\n"
          "   --qic-binary F raw per-quartet QIC as a binary file (topology byte + double per quartet, in rank order)\n";
}

// returns 0 ok, 1 error (message printed like the reference prints TCLAP::ArgException), 2 exit quietly
int parse(int argc, char **argv, Args &a) {
    auto need = [&](int &i, const char *flag) -> const char * {
        if (i + 1 >= argc) {
            std::cerr << "ERROR: Missing a value for this argument! for arg " << flag << std::endl;
            return nullptr;
        }
        return argv[++i];
    };
    // numeric flag values: a malformed one is reported like the other argument errors (tclap: "Couldn't read argument
    // value from string"), not an uncaught std::invalid_argument
    auto number = [&](const char *text, const char *flag, unsigned long &out) {
        char *end = nullptr;
        errno = 0;
        const unsigned long val = std::strtoul(text, &end, 10);
        if (end == text || *end != '\0' || errno != 0 || text[0] == '-') {
            std::cerr << "ERROR: Couldn't read argument value from string '" << text << "' for arg " << flag << std::endl;
            return false;
        }
        out = val;
        return true;
    };
    unsigned long num = 0;
    for (int i = 1; i < argc; ++i) {
        std::string f = argv[i];
        const char *v = nullptr;
        if (f == "-r" || f == "--ref") { if (!(v = need(i, "-r (--ref)"))) return 1; a.ref = v; }
        else if (f == "-e" || f == "--eval") { if (!(v = need(i, "-e (--eval)"))) return 1; a.eval = v; }
        else if (f == "-o" || f == "--output") { if (!(v = need(i, "-o (--output)"))) return 1; a.out = v; }
        else if (f == "-q" || f == "--qic") { if (!(v = need(i, "-q (--qic)"))) return 1; a.raw = v; }
        else if (f == "-t" || f == "--threads") { if (!(v = need(i, "-t (--threads)")) || !number(v, "-t (--threads)", num)) return 1; a.threads = num; }
        else if (f == "-v" || f == "--verbose") a.verbose = true;
        else if (f == "-s" || f == "--savemem") a.savemem = true;
        else if (f == "--gpus") { if (!(v = need(i, "--gpus")) || !number(v, "--gpus", num)) return 1; a.gpus = (int)num; }
        else if (f == "--table-shards") { if (!(v = need(i, "--table-shards")) || !number(v, "--table-shards", num)) return 1; a.table_shards = (int)num; }
        else if (f == "--spill") {
            if (!(v = need(i, "--spill"))) return 1;
            const std::string sv = v;
            if (sv == "host") a.spill = ShardedTableQuartetScoreComputer::SPILL_HOST;
            else if (sv == "recount") a.spill = ShardedTableQuartetScoreComputer::SPILL_RECOUNT;
            else if (sv == "auto") a.spill = ShardedTableQuartetScoreComputer::SPILL_AUTO;
            else { std::cerr << "ERROR: --spill takes host, recount or auto" << std::endl; return 1; }
        }
        else if (f == "--device") { if (!(v = need(i, "--device")) || !number(v, "--device", num)) return 1; a.dev.device = (int)num; }
        else if (f == "--algo") {
            if (!(v = need(i, "--algo"))) return 1;
            a.dev.algo = std::string(v) == "scatter" ? QS_ALGO_SCATTER : QS_ALGO_GATHER;
        } else if (f == "--exact-qp") a.dev.qp_exact64 = true;
        else if (f == "--root-as-edge") a.dev.root_as_edge = true;
        else if (f == "--mode") {
            if (!(v = need(i, "--mode"))) return 1;
            a.mode = v;
            if (a.mode != "auto" && a.mode != "tree" && a.mode != "table") { std::cerr << "ERROR: --mode takes auto, tree or table" << std::endl; return 1; }
        }
        else if (f == "--fail-fast") a.fail_fast = true;
        else if (f == "--clean-exit") a.clean_exit = true;
        else if (f == "--fast-exit") fast_exit_forced = true;
        else if (f == "--reduce") {
            if (!(v = need(i, "--reduce"))) return 1;
            a.dev.reduce = v;
            if (a.dev.reduce != "rccl" && a.dev.reduce != "p2p") { std::cerr << "ERROR: --reduce takes rccl or p2p" << std::endl; return 1; }
        }
        else if (f == "--gpus-on-one-device") a.dev.gpus_on_one_device = true;
        else if (f == "--comm-overlap") { if (!(v = need(i, "--comm-overlap")) || !number(v, "--comm-overlap", num)) return 1; a.dev.comm_overlap = num != 0; }
        else if (f == "--trace") a.dev.trace = true;
        else if (f == "--qic-rank-order") a.raw_rank_order = true;
        else if (f == "--save-table") { if (!(v = need(i, "--save-table"))) return 1; a.dev.save_table = v; }
        else if (f == "--qic-binary") { if (!(v = need(i, "--qic-binary"))) return 1; a.raw_bin = v; }
        else if (f == "--load-table") { if (!(v = need(i, "--load-table"))) return 1; a.dev.load_table = v; }
        else if (f == "--version") { std::cout << argv[0] << "  version: 1.0.1 (" << qs_version() << ")" << std::endl; return 2; }
        else if (f == "-h" || f == "--help") { usage(std::cout); return 2; }
        else { std::cerr << "ERROR: Couldn't find match for argument for arg " << f << std::endl; return 1; }
    }
    const char *missing = a.ref.empty() ? "-r (--ref)" : a.eval.empty() ? "-e (--eval)" : a.out.empty() ? "-o (--output)" : nullptr;
    if (missing) { std::cerr << "ERROR: Required argument missing: for arg " << missing << std::endl; return 1; }
    // A wrapper that flushes in its exit handlers (rocprofv3, a tracer, a coverage or sanitizer runtime) would lose its output to the
    // fast exit: with a preloaded library or a profiler's environment the ordinary return is the default (--fast-exit overrides)
    if (!a.clean_exit && !fast_exit_forced && wrapped_by_tool()) a.clean_exit = true;
    return 0;
}

// --gpus N: trees split over N GPUs, one collective on the table, sharded scoring (multi_gpu.hpp)
void run_multi(const Tree &referenceTree, const Args &a, size_t m, uint32_t count_bits, std::vector<double> &lqic,
               std::vector<double> &qpic, std::vector<double> &eqpic) {
    if (!a.dev.load_table.empty() || !a.dev.save_table.empty()) throw std::runtime_error("--save-table / --load-table work on one GPU (omit --gpus)");
    const bool need_full = !a.raw.empty() || !a.raw_bin.empty();   // the -q dump walks the whole table on one GPU
    MultiGpuQuartetScoreComputer mg(referenceTree, a.eval, m, count_bits, a.gpus, need_full, a.dev);
    lqic = mg.scores.lq; qpic = mg.scores.qp; eqpic = mg.scores.eqp;
    if (!a.raw.empty()) print_raw_qic_scores(mg.context0(), mg.reference(), referenceTree, a.raw, a.dev.ingest_threads, a.raw_rank_order);
    if (!a.raw_bin.empty()) print_raw_qic_binary(mg.context0(), mg.reference(), referenceTree, a.raw_bin);
}

// --table-shards K [--gpus N]: the table cut into K shards by the largest taxon id, shard s on GPU s mod N; every GPU counts
// all trees into its shard(s), no table collective (table_shards.hpp; BASELINE configs[4] = --gpus 8 --table-shards 8)
void run_sharded(const Tree &referenceTree, const Args &a, size_t m, uint32_t count_bits, int shards, int gpus, std::vector<double> &lqic,
                 std::vector<double> &qpic, std::vector<double> &eqpic) {
    if (!a.dev.load_table.empty() || !a.dev.save_table.empty() || !a.raw.empty() || !a.raw_bin.empty())
        throw std::runtime_error("--table-shards: -q / --qic-binary / --save-table / --load-table need the whole table on one device");
    ShardedTableQuartetScoreComputer st(referenceTree, a.eval, m, count_bits, shards, (ShardedTableQuartetScoreComputer::Spill)a.spill, a.dev, gpus);
    lqic = st.scores.lq; qpic = st.scores.qp; eqpic = st.scores.eqp;
}

template <typename CINT>
void run(const Tree &referenceTree, const Args &a, size_t m, std::vector<double> &lqic, std::vector<double> &qpic,
         std::vector<double> &eqpic) {
    const uint32_t bits = sizeof(CINT) <= 2 ? 16u : 32u;
    size_t n = 0;
    for (size_t v = 0; v < referenceTree.node_count(); ++v) n += referenceTree.is_leaf(v);
    const uint64_t bytes = (uint64_t)n * (n - 1) * (n - 2) * (n - 3) / 24 * 3 * (bits / 8);
    const int gpus = std::max(1, a.gpus);
    if (a.table_shards >= 0 || a.gpus > 0) {
        // shards one device's free memory asks for (0 / unset = automatic); with --gpus N at least one shard per GPU
        const int needed = ShardedTableQuartetScoreComputer::shards_needed(bytes, a.dev.device);
        if (a.table_shards > 0) return run_sharded(referenceTree, a, m, bits, a.table_shards, gpus, lqic, qpic, eqpic);
        if (needed > 1) {
            if (a.gpus > 0 && a.table_shards < 0)
                std::cout << "The count table (" << bytes << " bytes) does not fit one GPU: table-sharded mode instead of tree-sharded.\n";
            return run_sharded(referenceTree, a, m, bits, std::max(needed, gpus), gpus, lqic, qpic, eqpic);
        }
        if (a.gpus > 0 && a.table_shards == 0) return run_sharded(referenceTree, a, m, bits, gpus, gpus, lqic, qpic, eqpic);
    }
    if (a.gpus > 1 && a.table_shards < 0) {
        // N GPUs, the table fits one of them: tree- or table-sharded. The count work per GPU is the same either way;
        // the table-sharded mode needs no collective and no communicator, the tree-sharded one keeps the whole table on GPU 0
        // (which -q / --qic-binary / --save-table walk).
        const bool need_whole = !a.raw.empty() || !a.raw_bin.empty() || !a.dev.save_table.empty() || !a.dev.load_table.empty();
        double coll_ms = 0.0, extra_ms = 0.0;
        const bool model_table = ShardedTableQuartetScoreComputer::prefer_table_shards((uint32_t)n, m, gpus, a.dev.reduce == "rccl", coll_ms, extra_ms);
        // (--gpus-on-one-device without --mode stays the test hook of the tree-sharded reductions it was written for)
        const bool table = !need_whole && (a.mode == "table" || (a.mode == "auto" && model_table && !a.dev.gpus_on_one_device));
        if (a.mode == "table" && need_whole) std::cout << "--mode table: -q / --qic-binary / --save-table / --load-table need the whole table on one device; tree-sharded mode instead.\n";
        if (table) {
            std::cout << "Table-sharded counting on " << gpus << " GPUs (" << (a.mode == "table" ? "--mode table" : "auto")
                      << ": table collective ~" << coll_ms << " ms against ~" << extra_ms << " ms of replicated panel build and imbalance).\n";
            return run_sharded(referenceTree, a, m, bits, gpus, gpus, lqic, qpic, eqpic);
        }
    }
    if (a.gpus > 0) return run_multi(referenceTree, a, m, bits, lqic, qpic, eqpic);
    // (without --clean-exit the computer is never destroyed: freeing a 17-34 GB table and the context is work the exiting process
    // leaves to the driver -- main ends with std::_Exit once the output is written)
    std::unique_ptr<QuartetScoreComputer<CINT>> holder(new QuartetScoreComputer<CINT>(referenceTree, a.eval, m, a.verbose, a.savemem, a.dev));
    QuartetScoreComputer<CINT> &qsc = *holder;
    lqic = qsc.getLQICScores();
    qpic = qsc.getQPICScores();
    eqpic = qsc.getEQPICScores();
    qsc.raw_threads = a.dev.ingest_threads;
    qsc.raw_rank_order = a.raw_rank_order;
    if (!a.raw.empty()) qsc.printRawQICScores(referenceTree, a.raw);
    if (!a.raw_bin.empty()) qsc.printRawQICBinary(referenceTree, a.raw_bin);
    if (!a.clean_exit) (void)holder.release();
}

} // namespace

int main(int argc, char *argv[]) {
    auto begin = std::chrono::steady_clock::now();
    Args a;
    int pr = parse(argc, argv, a);
    if (pr == 1) return 1;
    if (pr == 2) return 0;

    std::ifstream infile(a.out);
    if (infile.good()) {
        std::cout << "ERROR: The specified output file already exists.\n";
        return 1;
    }
    a.dev.ingest_threads = (unsigned)a.threads;
    a.dev.savemem_lookups = a.savemem;   // -s: the reference's compact table behind the lookups of a rooted reference tree
    trace_mark(a.dev, "main: arguments parsed");
    // HIP start-up (~0.1-0.2 s: driver, device, code objects) begins NOW on a helper thread, while this thread reads and
    // splits the Newick files; the counter's own set-up thread then finds the runtime initialised
    std::thread hip_start([&a] { qs_ctx *probe = nullptr; if (qs_create(&probe, 4, 16, QS_FLAG_NONE, a.dev.device, nullptr, 0, 0) == QS_OK) qs_destroy(probe); trace_mark(a.dev, "hip thread: runtime initialised"); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } hip_start_join{hip_start};

    try {
        std::string refText = slurp(a.ref);
        NewickReader rr(refText);
        Tree referenceTree;
        if (!rr.next(referenceTree)) throw std::runtime_error("empty reference tree file");

        if (a.verbose) {
            for (size_t v = 0; v < referenceTree.node_count(); ++v)
                if (referenceTree.is_leaf(v)) std::cout << referenceTree.name[v] << " " << v << "\n";
            std::cout << std::endl;
        }

        // What the scoring will refuse because of the reference tree alone is known before anything is counted (qs_score_check,
        // host-only): `-s` with a rooted reference tree. The reference program counts first and dies in its scoring loop; this run
        // says so NOW and then does the same -- or ends at once with --fail-fast.
        if (a.savemem) {
            const RefFlat rf0 = flatten_reference(referenceTree);
            qs_ref_tree rt0;
            rt0.n_nodes = (uint32_t)referenceTree.node_count(); rt0.n_taxa = (uint32_t)rf0.names.size();
            rt0.parent = rf0.parent.data(); rt0.leaf_node = rf0.leaf_node.data();
            if (qs_score_check(nullptr, &rt0, QS_SCORE_SAVEMEM_LOOKUPS | (a.dev.root_as_edge ? QS_SCORE_ROOT_AS_EDGE : 0u)) == QS_ERR_REFERENCE_THROWS) {
                const std::string what = qs_last_error(nullptr);
                if (a.fail_fast) throw std::runtime_error(what);
                std::cerr << "Note: -s with a rooted reference tree: the reference program ends in its scoring loop with \"" << what
                          << "\" after it has counted, and so will this run (--fail-fast ends it now).\n";
            }
        }

        std::vector<double> lqic, qpic, eqpic;
        size_t m = countEvalTrees(a.eval);
        trace_mark(a.dev, "main: evaluation file read and split into trees");
        // (round 5: no join of the HIP start-up here -- the counter's GPU set-up thread simply blocks in its first HIP call until the
        // runtime is up, while THIS thread already flattens the first batch: 25-30 ms of the start-up leave the critical path)
        // counter width by m as in QuartetScores.cpp:115-147 (u8 is widened to the GPU's 16-bit cells)
        if (m < (size_t(1) << 8)) run<uint8_t>(referenceTree, a, m, lqic, qpic, eqpic);
        else if (m < (size_t(1) << 16)) run<uint16_t>(referenceTree, a, m, lqic, qpic, eqpic);
        else if (m < (size_t(1) << 32)) run<uint32_t>(referenceTree, a, m, lqic, qpic, eqpic);
        else throw std::runtime_error("more than 2^32 evaluation trees are not supported");

        // annotated Newick: per edge "qp-ic:X;lq-ic:Y;eqp-ic:Z" via std::to_string, parts omitted when +inf;
        // the qp-ic guard tests the LQ value like the reference (quartet_newick_writer.hpp:164-187, quirk Q6)
        const double inf = std::numeric_limits<double>::infinity();
        auto comment = [&](size_t v) -> std::string {
            if (v == 0) return std::string();
            const size_t e = v - 1;
            std::string s;
            auto add = [&](const std::string &part) { if (!s.empty()) s += ";"; s += part; };
            if (!qpic.empty() && lqic[e] != inf) add("qp-ic:" + std::to_string(qpic[e]));
            if (lqic[e] != inf) add("lq-ic:" + std::to_string(lqic[e]));
            if (!eqpic.empty() && eqpic[e] != inf) add("eqp-ic:" + std::to_string(eqpic[e]));
            return s;
        };
        std::ofstream out(a.out);
        if (!out) throw std::runtime_error("cannot write " + a.out);
        out << write_newick(referenceTree, comment) << "\n";
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        if (a.savemem && std::string(e.what()).rfind("id = ", 0) == 0)
            std::cerr << "       (-s with a rooted reference tree: the reference's memory-efficient table throws this std::runtime_error for the\n"
                         "       node pairs of the root, quartet_lookup_table.hpp:79-85, and its run ends here as well. Without -s the root's\n"
                         "       pairs are scored like the reference's runtime-efficient table scores them; --root-as-edge treats the root as\n"
                         "       a point on one edge.)\n";
        return 1;
    }

    auto end = std::chrono::steady_clock::now();
    std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()
              << " microseconds." << std::endl;
    if (!a.clean_exit) {
        // Everything the run produces is written and closed. What is left is tearing down the HIP runtime (and RCCL): tens of
        // milliseconds of a 0.5 s run that buy nothing -- the driver reclaims the process' device memory either way.
        // --clean-exit keeps the ordinary return (profilers and sanitizers flush in their exit handlers).
        std::cout.flush(); std::cerr.flush(); std::fflush(nullptr);
        if (hip_start.joinable()) hip_start.join();
        std::_Exit(0);
    }
    return 0;
}
