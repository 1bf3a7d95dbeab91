// multi_gpu.hpp -- the C++ host on N GPUs of one node (`QuartetScores --gpus N`).
//
// The reference has no counterpart (one process, OpenMP). Evaluation trees are independent and counts
// add (QuartetCounterLookup.hpp:196-238 walks them one by one), so the trees are split over the GPUs:
//   1. one host thread + one C-ABI context per GPU; thread g parses, flattens and counts the trees
//      [g m / N, (g + 1) m / N) into its own full table (the table lives in memory this file allocates, qs_table_attach,
//      padded to N equal chunks);
//   2. ONE reduction of the tables over xGMI:
//        reduce-scatter (default): GPU g ends with tuples [g T, (g + 1) T) of the summed table, half the bytes per link
//        of an all-reduce, and scores that shard in place (qs_score_set_view);
//        all-reduce / reduce to GPU 0 (when the -q dump needs the whole table on one GPU): GPU 0 scores alone (qs_score);
//      carried out either by RCCL (`--reduce rccl`, the default: ncclCommInitAll + one group call) or, since this is ONE
//      process that owns all N devices, by peer access (`--reduce p2p`): GPU g sums chunk g of every peer's table with plain
//      16-byte loads over its xGMI links (qs_sum_words) -- no communicator, whose creation (1.7-5.6 s) costs more than the
//      whole count of BASELINE configs[3] on this engine (0.03 s per GPU);
//   3. sharded scoring: qs_score_pass1 on every GPU, the per-node-pair sums / minima (a few MB) are added / minimised on
//      the host and the minima handed back, qs_score_pass2 + qs_score_overflow on every GPU, qs_score_finish on the host.
// u16 tables travel as packed 32-bit words: every total stays below 2^16 (m < 65536 is what selects u16), so no carry
// crosses a half-word. Same scores as one GPU (tests/test_cli.py runs --gpus 1 through this path against the oracle).
#pragma once

#include "QuartetScoreComputer.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is dlopen'ed (Rccl below)
#include <dlfcn.h>

#include <atomic>
#include <future>
#include <mutex>

namespace qsh {

#define QSM_HIP(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
// RCCL is loaded on first use (dlopen), not linked: librccl.so is 570 MB of code objects that every start of the CLI would map
// and register with the HIP runtime, although only `--gpus N --reduce rccl` ever calls it (round 5: the single-GPU CLI's wall).
struct Rccl {
    decltype(&::ncclCommInitAll) CommInitAll = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclReduceScatter) ReduceScatter = nullptr;
    static Rccl &get() {
        static Rccl r = load();
        return r;
    }
private:
    static Rccl load() {
        Rccl r;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) throw std::runtime_error(std::string("--reduce rccl: cannot load librccl.so.1: ") + dlerror());
        auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) throw std::runtime_error(std::string("librccl.so.1 lacks ") + name); return p; };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
        return r;
    }
};
#define QSM_NCCL(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + Rccl::get().GetErrorString(r__)); \
    } while (0)

struct MultiGpuScores {
    std::vector<double> lq, qp, eqp; // per edge (edge e = edge above node e + 1), qp / eqp empty for a multifurcating reference
    bool bifurcating = false;
};

class MultiGpuQuartetScoreComputer {
public:
    // count_bits 16 | 32 (by m like QuartetScores.cpp:115-147); need_full_table: all-reduce instead of reduce-scatter
    MultiGpuQuartetScoreComputer(Tree const &refTree, const std::string &evalTreesPath, size_t m, uint32_t count_bits, int n_gpus,
                                 bool need_full_table, DeviceOptions opt)
        : ref_(flatten_reference(refTree)), opt_(opt), bits_(count_bits), full_(need_full_table) {
        int ndev = 0;
        QSM_HIP(hipGetDeviceCount(&ndev));
        if (opt.gpus_on_one_device && opt.reduce != "p2p") throw std::runtime_error("--gpus-on-one-device needs --reduce p2p (RCCL refuses two ranks on one device)");
        if (n_gpus < 1 || opt.device < 0 || opt.device + (opt.gpus_on_one_device ? 1 : n_gpus) > ndev)   // the devices used are opt.device .. opt.device + n_gpus - 1
            throw std::runtime_error("--gpus " + std::to_string(n_gpus) + " from --device " + std::to_string(opt.device) + ": " + std::to_string(ndev) + " device(s) visible");
        G_ = n_gpus;
        std::cout << "There are " << m << " evaluation trees.\n";
        std::cout << "The reference tree has " << ref_.names.size() << " taxa.\n";
        std::cout << "Counting on " << G_ << " GPU(s): trees split over the GPUs, one " << (opt_.reduce == "p2p" ? "peer-access " : "RCCL ") << (full_ ? "all-reduce" : "reduce-scatter") << " of the count table.\n";
        const auto t0 = std::chrono::steady_clock::now();
        ctx_.assign(G_, nullptr);
        table_.assign(G_, nullptr);
        send_.assign(G_, nullptr);
        // RCCL's communicators take 1.7-5.6 s to create (ncclCommInitAll), on a helper thread that starts NOW. Nothing but the
        // collective itself needs them: the GPUs' host threads create their contexts, allocate and parse meanwhile. By DEFAULT
        // (opt.comm_overlap = false, `--comm-overlap 0`) the workers wait for the communicators before their first launch: the
        // creation's many small device operations queue behind 100 ms count kernels (5.6 s instead of 2.8 s at 512 taxa x
        // 10000 trees, measured in rounds 3 and 4); `--comm-overlap 1` counts beside the creation (A/B runs only).
        // `--reduce p2p` creates no communicator at all.
        std::vector<int> devs(G_);
        for (int g = 0; g < G_; ++g) devs[g] = dev_of(g);
        comms_.assign(G_, nullptr);
        ncclResult_t comm_rc = ncclSuccess;
        std::promise<void> comm_done;
        const bool use_rccl = opt_.reduce != "p2p";
        if (use_rccl && !opt_.comm_overlap) comm_ready_ = comm_done.get_future().share();
        std::thread comm_init;
        if (use_rccl) (void)Rccl::get();   // dlopen here, on this thread: a missing library is an exception the caller sees
        if (use_rccl)
            comm_init = std::thread([&] { trace_mark(opt_, "rccl thread: ncclCommInitAll starts"); comm_rc = Rccl::get().CommInitAll(comms_.data(), G_, devs.data()); trace_mark(opt_, "rccl thread: communicators ready"); comm_done.set_value(); });
        try {
            try { count(evalTreesPath, m); } catch (...) { if (comm_init.joinable()) comm_init.join(); throw; }
            trace_mark(opt_, "main: all GPUs counted");
            if (comm_init.joinable()) comm_init.join();
            trace_mark(opt_, use_rccl ? "main: communicators joined" : "main: no communicator (peer access)");
            if (comm_rc != ncclSuccess) { comms_.assign(G_, nullptr); throw std::runtime_error(std::string("ncclCommInitAll: ") + Rccl::get().GetErrorString(comm_rc)); }
            if (use_rccl) reduce(m); else reduce_p2p(m);
            trace_mark(opt_, "main: tables reduced");
            const auto t1 = std::chrono::steady_clock::now();
            std::cout << "Finished counting quartets.\nIt took: " << std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() << " microseconds." << std::endl;
            score(refTree);
            const auto t2 = std::chrono::steady_clock::now();
            std::cout << (scores.bifurcating ? "The reference tree is bifurcating.\n" : "The reference tree is multifurcating.\n");
            std::cout << "Finished computing scores.\nIt took: " << std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count() << " microseconds." << std::endl;
        } catch (...) {
            release();
            throw;
        }
    }
    ~MultiGpuQuartetScoreComputer() { release(); }
    MultiGpuQuartetScoreComputer(const MultiGpuQuartetScoreComputer &) = delete;
    MultiGpuQuartetScoreComputer &operator=(const MultiGpuQuartetScoreComputer &) = delete;

    MultiGpuScores scores;
    qs_ctx *context0() const { return ctx_[0]; }        // holds the whole table after an all-reduce (-q)
    const RefFlat &reference() const { return ref_; }

private:
    RefFlat ref_;
    DeviceOptions opt_;
    uint32_t bits_;
    bool full_;
    int G_ = 1;
    std::vector<qs_ctx *> ctx_;
    std::vector<void *> table_, send_;        // send_: the two-cell wire words (binary full trees, u32 tables)
    std::vector<ncclComm_t> comms_;
    std::shared_future<void> comm_ready_;      // the communicators exist (count kernels start only after that)
    std::atomic<uint32_t> flags_and_{~0u};     // AND of qs_batch_flags over every batch of every GPU
    uint64_t tuples_ = 0, chunk_tuples_ = 0, chunk_words_ = 0;

    // device of "GPU" g. --gpus-on-one-device (a test hook of --reduce p2p): all N contexts, tables and the peer sums live on
    // ONE device, so that the N-way partition, the chunk arithmetic and qs_sum_words are exercised on a 1-GPU box.
    int dev_of(int g) const { return opt_.gpus_on_one_device ? opt_.device : opt_.device + g; }

    void release() {
        for (auto &cm : comms_) if (cm) { (void)Rccl::get().CommDestroy(cm); cm = nullptr; }
        for (int g = 0; g < (int)ctx_.size(); ++g) {
            if (ctx_[g]) qs_destroy(ctx_[g]);
            (void)hipSetDevice(dev_of(g));
            if (table_[g]) (void)hipFree(table_[g]);
            if (g < (int)send_.size() && send_[g]) (void)hipFree(send_[g]);
        }
        ctx_.clear(); table_.clear(); send_.clear();
    }

    void count(const std::string &evalTreesPath, size_t m) {
        auto ef = loadEvalFile(evalTreesPath);
        const auto &spans = ef->spans;
        if (spans.size() != m) throw std::runtime_error("evaluation file changed while running");
        const uint32_t n = (uint32_t)ref_.names.size();
        std::vector<std::exception_ptr> errs(G_);
        std::mutex io;
        // geometry of the padded table: N chunks of T tuples (T even: whole 32-bit words for u16 cells too)
        tuples_ = (uint64_t)n * (n - 1) * (n - 2) * (n - 3) / 24;
        chunk_tuples_ = (tuples_ + G_ - 1) / G_;
        chunk_tuples_ = (chunk_tuples_ + 7) & ~(uint64_t)7;   // chunks start on 16-byte boundaries in every wire format (qs_sum_words)
        chunk_words_ = chunk_tuples_ * 3 * (bits_ / 8) / 4;
        const unsigned host_threads = std::max(1u, (opt_.ingest_threads ? opt_.ingest_threads : std::thread::hardware_concurrency()) / (unsigned)G_);
        auto worker = [&](int g) {
            try {
                const int dev = dev_of(g);
                if (qs_create(&ctx_[g], n, bits_, QS_FLAG_NONE, dev, nullptr, 0, 0) != QS_OK) throw std::runtime_error(qs_last_error(nullptr));
                QSM_HIP(hipSetDevice(dev));
                const size_t bytes = (size_t)chunk_words_ * 4 * G_;
                if (hipMalloc(&table_[g], bytes) != hipSuccess) throw std::runtime_error("Insufficient memory!");
                QSM_HIP(hipMemset(table_[g], 0, bytes));
                if (qs_table_attach(ctx_[g], table_[g], bytes) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                // the two-cell wire words (8 bytes per tuple) are allocated HERE, beside the parse of the first batch, not
                // between the count and the collective; freed again if the trees turn out not to be binary and full
                if (!full_ && bits_ == 32 && (opt_.algo & 0xFFu) != QS_ALGO_SCATTER && G_ > 1) {
                    const size_t sb = (size_t)chunk_tuples_ * 2 * 4 * G_;
                    if (hipMalloc(&send_[g], sb) == hipSuccess) QSM_HIP(hipMemset(send_[g], 0, sb));   // padding tuples stay zero (synchronous: ordered before the pack kernel on the context's stream and the peers' reads)
                    else { send_[g] = nullptr; (void)hipGetLastError(); }
                }
                const size_t lo = spans.size() * g / G_, hi = spans.size() * (g + 1) / G_;
                const bool want_ranges = (opt_.algo & 0xFFu) == QS_ALGO_SCATTER;
                std::vector<qs_device_batch *> in_flight;
                try {
                    for (size_t i0 = lo; i0 < hi; i0 += opt_.batch_trees) {
                        const size_t i1 = std::min(hi, i0 + opt_.batch_trees);
                        BatchFlat b = flatten_parallel(ef->text, spans, i0, i1, ref_.name_to_id, host_threads, want_ranges);
                        qs_tree_batch hb;
                        hb.n_trees = b.n_trees; hb.leaf_off = b.leaf_off.data(); hb.leaf_ids = b.leaf_ids.data(); hb.adj_depth = b.adj_depth.data();
                        hb.node_off = want_ranges ? b.node_off.data() : nullptr; hb.rng_off = want_ranges ? b.rng_off.data() : nullptr;
                        hb.ranges = b.ranges.data();
                        if (in_flight.size() == 2) { qs_batch_free(ctx_[g], in_flight.front()); in_flight.erase(in_flight.begin()); }
                        if (i0 == lo && comm_ready_.valid()) comm_ready_.wait();   // --comm-overlap 0: first launch not beside RCCL's set-up
                        qs_device_batch *db = nullptr;
                        if (qs_batch_upload(ctx_[g], &hb, &db) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                        flags_and_.fetch_and(qs_batch_flags(db));
                        in_flight.push_back(db);
                        if (qs_count_batch(ctx_[g], db, opt_.algo) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                    }
                    if (qs_sync(ctx_[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                } catch (...) {
                    (void)qs_sync(ctx_[g]);
                    for (auto *db : in_flight) qs_batch_free(ctx_[g], db);
                    throw;
                }
                for (auto *db : in_flight) qs_batch_free(ctx_[g], db);
                std::lock_guard<std::mutex> lk(io);
                std::cout << "GPU " << dev << ": counted trees [" << lo << ", " << hi << ")" << std::endl;
            } catch (...) { errs[g] = std::current_exception(); }
        };
        std::vector<std::thread> pool;
        for (int g = 0; g < G_; ++g) pool.emplace_back(worker, g);
        for (auto &th : pool) th.join();
        for (auto &e : errs) if (e) std::rethrow_exception(e);
        loadEvalFile(std::string(), true);
        std::cout << "lookup table size in bytes: " << tuples_ * 3 * (bits_ / 8) << "\n";
    }

    // one collective over all GPUs of this process. Reduce-scatter of u32 tables whose trees are all binary and hold all taxa:
    // the TWO-cell wire format, (n0, n1) per tuple = 8 instead of 12 bytes per quartet on xGMI (qs_table_pack32x2 /
    // qs_unpack32x2; n2 = m - n0 - n1) -- BASELINE configs[3] (100 000 trees: u32 cells) moves 1.4 GB per GPU instead of 2.1.
    void reduce(size_t m) {
        const uint32_t both = QS_BATCH_ALL_TAXA | QS_BATCH_BINARY;
        const bool two_cell = !full_ && bits_ == 32 && (flags_and_.load() & both) == both && (opt_.algo & 0xFFu) != QS_ALGO_SCATTER;
        const uint64_t words2 = chunk_tuples_ * 2;                    // wire words per chunk in the two-cell format
        pack_two_cell(two_cell, words2);
        bool group_open = false;
        try {
            QSM_NCCL(Rccl::get().GroupStart());
            group_open = true;
            for (int g = 0; g < G_; ++g) {
                QSM_HIP(hipSetDevice(dev_of(g)));
                uint32_t *buf = (uint32_t *)table_[g];
                if (full_) QSM_NCCL(Rccl::get().AllReduce(buf, buf, chunk_words_ * G_, ncclUint32, ncclSum, comms_[g], nullptr));
                else if (two_cell) { uint32_t *sb = (uint32_t *)send_[g]; QSM_NCCL(Rccl::get().ReduceScatter(sb, sb + (size_t)g * words2, words2, ncclUint32, ncclSum, comms_[g], nullptr)); }
                else QSM_NCCL(Rccl::get().ReduceScatter(buf, buf + (size_t)g * chunk_words_, chunk_words_, ncclUint32, ncclSum, comms_[g], nullptr));
            }
            group_open = false;
            QSM_NCCL(Rccl::get().GroupEnd());
            for (int g = 0; g < G_; ++g) { QSM_HIP(hipSetDevice(dev_of(g))); QSM_HIP(hipDeviceSynchronize()); }
        } catch (...) {
            if (group_open) (void)Rccl::get().GroupEnd();     // never destroy communicators inside an open group
            throw;
        }
        unpack_two_cell(two_cell, words2, m);
        for (int g = 0; g < G_; ++g) (void)qs_set_tuning(ctx_[g], QS_TUNE_TABLE_TREES, (uint64_t)m);   // the reduced table holds all m trees
    }

    // (n0, n1) of every tuple into the pre-allocated wire words of every GPU: all packs are enqueued, then all are awaited
    void pack_two_cell(bool two_cell, uint64_t words2) {
        for (int g = 0; g < G_; ++g) {
            QSM_HIP(hipSetDevice(dev_of(g)));
            if (!two_cell) { if (send_[g]) { (void)hipFree(send_[g]); send_[g] = nullptr; } continue; }
            if (!send_[g]) {
                if (hipMalloc(&send_[g], (size_t)words2 * 4 * G_) != hipSuccess) throw std::runtime_error("Insufficient memory!");
                QSM_HIP(hipMemset(send_[g], 0, (size_t)words2 * 4 * G_));
            }
            if (qs_table_pack32x2(ctx_[g], send_[g], (uint64_t)words2 * 4 * G_) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
        }
        if (two_cell) for (int g = 0; g < G_; ++g) if (qs_sync(ctx_[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
    }
    // the reduced pairs of GPU g's chunk -> [rank][3] tuples where score() expects its shard
    void unpack_two_cell(bool two_cell, uint64_t words2, size_t m) {
        if (!two_cell) return;
        std::cout << "Wire format: two u32 cells per tuple (binary trees holding all taxa).\n";
        for (int g = 0; g < G_; ++g) {
            QSM_HIP(hipSetDevice(dev_of(g)));
            const uint64_t lo = std::min<uint64_t>((uint64_t)g * chunk_tuples_, tuples_), cnt = std::min<uint64_t>(lo + chunk_tuples_, tuples_) - lo;
            char *shard = (char *)table_[g] + (size_t)g * chunk_words_ * 4;
            if (cnt && qs_unpack32x2(ctx_[g], (const uint32_t *)send_[g] + (size_t)g * words2, cnt, (uint64_t)m, shard) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
        }
        for (int g = 0; g < G_; ++g) {
            if (qs_sync(ctx_[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
            QSM_HIP(hipSetDevice(dev_of(g)));
            (void)hipFree(send_[g]); send_[g] = nullptr;
        }
    }

    // `--reduce p2p`: the same reduction without a communicator. This process owns all N devices, so every GPU maps its
    // peers' memory (hipDeviceEnablePeerAccess) and sums ITS chunk of every peer's table with plain loads over xGMI
    // (qs_sum_words: 16 bytes per lane, N - 1 sources): 7 links x their read rate per GPU, each byte crossing one link
    // once -- the traffic of a reduce-scatter. With -q (full table on GPU 0) GPU 0 sums the whole tables of its peers.
    void reduce_p2p(size_t m) {
        const uint32_t both = QS_BATCH_ALL_TAXA | QS_BATCH_BINARY;
        const bool two_cell = !full_ && bits_ == 32 && G_ > 1 && (flags_and_.load() & both) == both && (opt_.algo & 0xFFu) != QS_ALGO_SCATTER;
        const uint64_t words2 = chunk_tuples_ * 2;
        for (int g = 0; g < G_; ++g) {
            QSM_HIP(hipSetDevice(dev_of(g)));
            for (int p = 0; p < G_; ++p) {
                if (p == g || dev_of(p) == dev_of(g)) continue;
                int can = 0;
                QSM_HIP(hipDeviceCanAccessPeer(&can, dev_of(g), dev_of(p)));
                if (!can) throw std::runtime_error("--reduce p2p: device " + std::to_string(dev_of(g)) + " cannot access device " + std::to_string(dev_of(p)) + " (use --reduce rccl)");
                const hipError_t e = hipDeviceEnablePeerAccess(dev_of(p), 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw std::runtime_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            }
        }
        pack_two_cell(two_cell, words2);          // (every GPU's counting has been awaited by its worker: qs_sync)
        for (int g = 0; g < (full_ ? 1 : G_); ++g) {
            std::vector<const void *> src;
            char *dst;
            uint64_t nw;
            if (full_) { dst = (char *)table_[0]; nw = chunk_words_ * G_; for (int p = 1; p < G_; ++p) src.push_back(table_[p]); }
            else if (two_cell) { dst = (char *)send_[g] + (size_t)g * words2 * 4; nw = words2; for (int p = 0; p < G_; ++p) if (p != g) src.push_back((const char *)send_[p] + (size_t)g * words2 * 4); }
            else { dst = (char *)table_[g] + (size_t)g * chunk_words_ * 4; nw = chunk_words_; for (int p = 0; p < G_; ++p) if (p != g) src.push_back((const char *)table_[p] + (size_t)g * chunk_words_ * 4); }
            if (qs_sum_words(ctx_[g], dst, src.data(), (uint32_t)src.size(), nw) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
        }
        for (int g = 0; g < (full_ ? 1 : G_); ++g) if (qs_sync(ctx_[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
        unpack_two_cell(two_cell, words2, m);
        for (int g = 0; g < G_; ++g) (void)qs_set_tuning(ctx_[g], QS_TUNE_TABLE_TREES, (uint64_t)m);
    }

    void score(Tree const &refTree) {
        qs_ref_tree rt;
        rt.n_nodes = (uint32_t)refTree.node_count(); rt.n_taxa = (uint32_t)ref_.names.size();
        rt.parent = ref_.parent.data(); rt.leaf_node = ref_.leaf_node.data();
        const uint32_t flags = (opt_.qp_exact64 ? QS_SCORE_QP_EXACT64 : QS_SCORE_QP_WRAP32) | (opt_.root_as_edge ? QS_SCORE_ROOT_AS_EDGE : 0u) |
                                   (opt_.savemem_lookups ? QS_SCORE_SAVEMEM_LOOKUPS : 0u);
        std::vector<double> lq(rt.n_nodes), qp(rt.n_nodes), eqp(rt.n_nodes);
        int bif = 0;
        if (full_) {   // every GPU holds the whole table: GPU 0 scores alone
            if (qs_score(ctx_[0], &rt, flags, lq.data(), qp.data(), eqp.data(), &bif) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[0]));
        } else {
            const size_t P = (size_t)qs_score_pair_slots(&rt);
            if (P == 0) throw std::runtime_error("bad reference tree");
            std::vector<int64_t *> d_sums(G_, nullptr), d_min(G_, nullptr), d_cand(G_, nullptr);
            std::vector<int64_t> sums(P * 3, 0), mins(P, INT64_MAX), cand((size_t)G_ * P * QS_SCORE_CAND_SLOTS), extra;
            auto free_all = [&]() {
                for (int g = 0; g < G_; ++g) { (void)hipSetDevice(dev_of(g)); (void)hipFree(d_sums[g]); (void)hipFree(d_min[g]); (void)hipFree(d_cand[g]); }
            };
            try {
                std::vector<int64_t> part_s(P * 3), part_m(P);
                for (int g = 0; g < G_; ++g) {   // pass 1 on every GPU (asynchronous), on its shard of the reduced table
                    QSM_HIP(hipSetDevice(dev_of(g)));
                    QSM_HIP(hipMalloc((void **)&d_sums[g], P * 3 * 8)); QSM_HIP(hipMalloc((void **)&d_min[g], P * 8)); QSM_HIP(hipMalloc((void **)&d_cand[g], P * QS_SCORE_CAND_SLOTS * 8));
                    const uint64_t lo = std::min<uint64_t>((uint64_t)g * chunk_tuples_, tuples_), cnt = std::min<uint64_t>(lo + chunk_tuples_, tuples_) - lo;
                    const char *shard = (const char *)table_[g] + (size_t)g * chunk_words_ * 4;
                    if (qs_score_set_view(ctx_[g], shard, bits_, lo, cnt) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                    if (qs_score_pass1(ctx_[g], &rt, d_sums[g], d_min[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                }
                for (int g = 0; g < G_; ++g) {   // SUM / MIN over the shards on the host (a few MB)
                    QSM_HIP(hipSetDevice(dev_of(g)));
                    QSM_HIP(hipMemcpy(part_s.data(), d_sums[g], P * 3 * 8, hipMemcpyDeviceToHost));
                    QSM_HIP(hipMemcpy(part_m.data(), d_min[g], P * 8, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < P * 3; ++i) sums[i] = (int64_t)((uint64_t)sums[i] + (uint64_t)part_s[i]);
                    for (size_t i = 0; i < P; ++i) mins[i] = std::min(mins[i], part_m[i]);
                }
                for (int g = 0; g < G_; ++g) {
                    QSM_HIP(hipSetDevice(dev_of(g)));
                    QSM_HIP(hipMemcpy(d_min[g], mins.data(), P * 8, hipMemcpyHostToDevice));
                    if (qs_score_pass2(ctx_[g], &rt, d_min[g], d_cand[g]) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                }
                for (int g = 0; g < G_; ++g) {
                    QSM_HIP(hipSetDevice(dev_of(g)));
                    int64_t *list = nullptr;
                    uint64_t k = 0;
                    if (qs_score_overflow(ctx_[g], &rt, d_min[g], d_cand[g], &list, &k) != QS_OK) throw std::runtime_error(qs_last_error(ctx_[g]));
                    if (k) { extra.insert(extra.end(), list, list + 4 * k); qs_free_host(list); }
                    QSM_HIP(hipMemcpy(cand.data() + (size_t)g * P * QS_SCORE_CAND_SLOTS, d_cand[g], P * QS_SCORE_CAND_SLOTS * 8, hipMemcpyDeviceToHost));
                    (void)qs_score_set_view(ctx_[g], nullptr, 0, 0, 0);
                }
            } catch (...) { free_all(); throw; }
            free_all();
            if (qs_score_finish(ctx_[0], &rt, flags, sums.data(), cand.data(), (uint32_t)G_, extra.empty() ? nullptr : extra.data(), extra.size() / 4,
                                lq.data(), qp.data(), eqp.data(), &bif) != QS_OK)
                throw std::runtime_error(qs_last_error(ctx_[0]));
        }
        scores.bifurcating = bif != 0;
        scores.lq.assign(lq.begin() + 1, lq.end());
        if (bif) { scores.qp.assign(qp.begin() + 1, qp.end()); scores.eqp.assign(eqp.begin() + 1, eqp.end()); }
    }
};

} // namespace qsh
