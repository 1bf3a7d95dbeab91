// table_shards.hpp -- the count table cut into K shards by the largest taxon id, on ONE GPU or spread over N GPUs
// (`QuartetScores --table-shards K [--gpus N]`).
//
// The reference keeps its table in host RAM and once planned an STXXL external-memory vector for tables beyond it
// (quartet_lookup_table.hpp:3,11-13,218-222); its disabled quartet-major scoring loop
// (QuartetScoreComputer.hpp:212-371, bucket key :266-273) is the blueprint of scoring by OWNED quartet. Here the table is
// cut into K shards by the largest taxon id (contiguous rank ranges, because the leading term of the rank is C(s3,4):
// quartet_lookup_table.hpp:161-165). Shard s lives on GPU s mod N; every GPU receives ALL evaluation trees (they are
// small) and counts only the quartets whose largest id falls into its shard -- no table collective at all:
//   round 1, per shard: count all trees into the shard (the flattened batches stay in host memory, shared by the
//            GPUs' host threads), score pass 1, fold the per-node-pair sums / minima into the GPU's host accumulators;
//            a GPU that owns ONE shard keeps it resident, otherwise the shard is copied to host memory
//            (spill = host) or dropped (spill = recount);
//   between the rounds: SUM of the sums, MIN of the minima over the GPUs (host, a few MB);
//   round 2, per shard: bring the shard back (still resident, upload, or count again), score pass 2 against the GLOBAL
//            minima, collect the candidate slots and overflow lists;
//   qs_score_finish once on the host.
// BASELINE configs[4] (1024 taxa, u16, 273 GB) is `--gpus 8 --table-shards 8`; a table larger than one device's memory on
// one GPU is `--table-shards K` alone (1200 taxa = 515 GB through one 288 GB MI355X in 3 shards). Same scores as the
// unsharded run and as the oracle (tests/test_cli.py).
#pragma once

#include "QuartetScoreComputer.hpp"

#include <hip/hip_runtime_api.h>

#include <fstream>
#include <mutex>

namespace qsh {

struct ShardedTableScores {
    std::vector<double> lq, qp, eqp; // per edge (edge e = edge above node e + 1), qp / eqp empty for a multifurcating reference
    bool bifurcating = false;
};

class ShardedTableQuartetScoreComputer {
public:
    enum Spill { SPILL_AUTO, SPILL_HOST, SPILL_RECOUNT };
    // bytes of host memory that can still be taken (MemAvailable of /proc/meminfo; 0 if unknown)
    static uint64_t host_mem_available() {
        std::ifstream f("/proc/meminfo");
        std::string key;
        uint64_t kb = 0;
        while (f >> key) {
            if (key == "MemAvailable:") { f >> kb; return kb * 1024; }
            std::getline(f, key);
        }
        return 0;
    }
    // number of shards so that one shard (plus panel and work space) fits the free device memory
    static int shards_needed(uint64_t table_bytes, int device) {
        size_t freeb = 0, total = 0;
        if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&freeb, &total) != hipSuccess || freeb == 0) return 1;
        const double room = 0.70 * (double)freeb;
        return table_bytes <= (uint64_t)(0.85 * (double)freeb) ? 1 : (int)std::ceil((double)table_bytes / room);
    }

    // Tree- or table-sharded on N GPUs when the table fits each of them (bench.py auto_mode holds the same arithmetic; measured basis:
    // profiles/r06_scaling_model.json). The count work per GPU is equal; the tree-sharded mode adds ONE collective on the table --
    // priced at the ring bound, reduce-scatter bytes per GPU over one 153 GB/s xGMI link, plus ~1.7 s when RCCL communicators have to
    // be created first --, the table-sharded mode the replicated panel build (1.6e-9 ms per tree and taxon pair), ~8 % of imbalance
    // and launch tails, and 1 ms per launch. true = table-sharded is the cheaper one.
    static bool prefer_table_shards(uint32_t n, size_t m, int gpus, bool rccl, double &coll_ms, double &extra_ms) {
        const double nq = (double)n * (n - 1) * (n - 2) * (n - 3) / 24.0, npairs = (double)n * (n - 1) / 2.0, G = (double)std::max(gpus, 1);
        const double bpt = m < 65536 ? 4.0 : 8.0;                     // two-cell wire formats of binary trees holding all taxa
        coll_ms = nq * bpt * (G - 1.0) / G / 153e9 * 1e3 + (rccl ? 1700.0 : 0.0);
        const double count_ms = (double)m * nq / 9.4e13 * 1e3 / G, panel_ms = 1.6e-9 * (double)m * npairs;
        const double groups = std::ceil((double)m / 32.0), slice_groups = std::max(256.0, std::floor(350e6 / (npairs * 16.0)));
        extra_ms = panel_ms * (G - 1.0) / G + 0.08 * count_ms + std::ceil(groups / slice_groups);
        return extra_ms < coll_ms;
    }

    // n_gpus devices opt.device .. opt.device + n_gpus - 1; n_shards >= 1 (fewer shards than GPUs leave GPUs idle)
    ShardedTableQuartetScoreComputer(Tree const &refTree, const std::string &evalTreesPath, size_t m, uint32_t count_bits, int n_shards,
                                     Spill spill, DeviceOptions opt, int n_gpus = 1)
        : ref_(flatten_reference(refTree)), opt_(opt), bits_(count_bits), G_(n_gpus) {
        const uint32_t n = (uint32_t)ref_.names.size();
        if (n_shards < 1) throw std::runtime_error("--table-shards needs a positive number");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw std::runtime_error("no HIP device (this program has no CPU fallback)");
        // (--gpus-on-one-device, a test hook: the N "GPUs" are N host threads with their own contexts on device opt.device)
        if (G_ < 1 || opt_.device < 0 || opt_.device + (opt_.gpus_on_one_device ? 1 : G_) > ndev)
            throw std::runtime_error("--gpus " + std::to_string(G_) + " from --device " + std::to_string(opt_.device) + ": " + std::to_string(ndev) + " device(s) visible");
        std::cout << "There are " << m << " evaluation trees.\n";
        std::cout << "The reference tree has " << n << " taxa.\n";
        const auto t0 = std::chrono::steady_clock::now();
        // shard bounds in the largest id d (qs_shard_bounds): one resident shard per GPU -> balanced by the count kernel's work;
        // more shards than GPUs (a table that passes through the devices shard by shard) -> by the tuples held, C(d,4): memory
        // decides there. Empty shards (small n) are dropped
        auto c4 = [](uint64_t x) { return x < 4 ? (uint64_t)0 : x * (x - 1) * (x - 2) * (x - 3) / 24; };
        const uint64_t total = c4(n);
        std::vector<uint32_t> bounds((size_t)n_shards + 1, 0);
        const uint32_t by = (n_shards > 1 && n_shards <= n_gpus) ? QS_SHARDS_BY_COST : QS_SHARDS_BY_TUPLES;
        if (qs_shard_bounds(n, (uint32_t)n_shards, by, bounds.data()) != QS_OK) throw std::runtime_error("qs_shard_bounds failed");
        for (size_t k = 0; k + 1 < bounds.size(); ++k)
            if (c4(bounds[k + 1]) > c4(bounds[k])) shards_.push_back({bounds[k], bounds[k + 1]});
        const size_t K = shards_.size();
        const uint64_t table_bytes = total * 3 * (bits_ / 8);
        // a GPU with one shard keeps it on the device between the rounds; only GPUs with several need a spill policy
        const bool any_multi = K > (size_t)G_;
        if (spill == SPILL_AUTO) {
            // Keeping a finished shard costs a round trip through (pageable) host memory, ~8 GB/s each way; counting it again
            // costs m x C(n,4) quartet-tree units at ~7e13 per second plus one pass over the shard per 4096 trees. Measured:
            // 1024 taxa x 500 trees in 8 shards on one GPU: 75 s with the spill, 2.5 s with the recount. Host memory is used
            // only when it is the cheaper of the two AND the table fits what is available.
            const uint64_t avail = host_mem_available();
            const double t_spill = 2.0 * (double)table_bytes / 8e9;
            const double t_recount = (double)m * (double)total / 7e13 + (double)((m + 4095) / 4096) * (double)table_bytes / 2e12;
            spill = (avail && table_bytes + (table_bytes >> 3) < avail && t_spill < t_recount) ? SPILL_HOST : SPILL_RECOUNT;
        }
        spill_host_ = spill == SPILL_HOST;
        std::cout << "Counting in " << K << " table shard(s) by largest taxon id on " << G_ << " GPU(s): every GPU counts all trees into its shard(s), no table collective; ";
        if (any_multi) std::cout << "finished shards are " << (spill_host_ ? "kept in host memory" : "dropped and counted again for the second scoring pass") << ".\n";
        else std::cout << "every shard stays on its GPU.\n";
        // the evaluation trees, flattened once (read-only afterwards: shared by the GPUs' host threads)
        {
            auto ef = loadEvalFile(evalTreesPath);
            if (ef->spans.size() != m) throw std::runtime_error("evaluation file changed while running");
            const bool want_ranges = (opt_.algo & 0xFFu) == QS_ALGO_SCATTER;
            const unsigned threads = opt_.ingest_threads ? opt_.ingest_threads : std::max(1u, std::thread::hardware_concurrency());
            for (size_t i0 = 0; i0 < m; i0 += opt_.batch_trees)
                batches_.push_back(flatten_parallel(ef->text, ef->spans, i0, std::min(m, i0 + opt_.batch_trees), ref_.name_to_id, threads, want_ranges));
            loadEvalFile(std::string(), true);
        }
        qs_ref_tree rt;
        rt.n_nodes = (uint32_t)refTree.node_count(); rt.n_taxa = n;
        rt.parent = ref_.parent.data(); rt.leaf_node = ref_.leaf_node.data();
        const size_t P = (size_t)qs_score_pair_slots(&rt);
        if (P == 0) throw std::runtime_error("bad reference tree");
        P_ = P; rt_ = &rt; m_ = m;

        std::vector<int64_t> sums(P * 3, 0), mins(P, INT64_MAX), cand(K * P * QS_SCORE_CAND_SLOTS), extra;
        gpu_.assign((size_t)G_, PerGpu());
        for (int g = 0; g < G_; ++g) {
            gpu_[g].dev = opt_.gpus_on_one_device ? opt_.device : opt_.device + g;
            for (size_t s = (size_t)g; s < K; s += (size_t)G_) gpu_[g].shards.push_back(s);
            gpu_[g].sums.assign(P * 3, 0); gpu_[g].mins.assign(P, INT64_MAX);
            gpu_[g].spilled.resize(gpu_[g].shards.size());
        }
        try {
            run_on_all([&](PerGpu &w) { round1(w); });
            for (const PerGpu &w : gpu_) {                     // SUM / MIN over the GPUs (host, a few MB)
                for (size_t i = 0; i < P * 3; ++i) sums[i] = (int64_t)((uint64_t)sums[i] + (uint64_t)w.sums[i]);
                for (size_t i = 0; i < P; ++i) mins[i] = std::min(mins[i], w.mins[i]);
            }
            const auto t1 = std::chrono::steady_clock::now();
            std::cout << "lookup table size in bytes: " << table_bytes << "\n";
            std::cout << "Finished counting quartets.\nIt took: " << std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() << " microseconds." << std::endl;
            run_on_all([&](PerGpu &w) { round2(w, mins, cand); });
            for (PerGpu &w : gpu_) extra.insert(extra.end(), w.extra.begin(), w.extra.end());
            release();
            const uint32_t flags = (opt_.qp_exact64 ? QS_SCORE_QP_EXACT64 : QS_SCORE_QP_WRAP32) | (opt_.root_as_edge ? QS_SCORE_ROOT_AS_EDGE : 0u) |
                                   (opt_.savemem_lookups ? QS_SCORE_SAVEMEM_LOOKUPS : 0u);
            std::vector<double> lq(rt.n_nodes), qp(rt.n_nodes), eqp(rt.n_nodes);
            int bif = 0;
            if (qs_score_finish(nullptr, &rt, flags, sums.data(), cand.data(), (uint32_t)K, extra.empty() ? nullptr : extra.data(), extra.size() / 4,
                                lq.data(), qp.data(), eqp.data(), &bif) != QS_OK)
                throw std::runtime_error(qs_last_error(nullptr));
            scores.bifurcating = bif != 0;
            scores.lq.assign(lq.begin() + 1, lq.end());
            if (bif) { scores.qp.assign(qp.begin() + 1, qp.end()); scores.eqp.assign(eqp.begin() + 1, eqp.end()); }
            const auto t2 = std::chrono::steady_clock::now();
            std::cout << (scores.bifurcating ? "The reference tree is bifurcating.\n" : "The reference tree is multifurcating.\n");
            std::cout << "Finished computing scores.\nIt took: " << std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count() << " microseconds." << std::endl;
        } catch (...) {
            release();
            rt_ = nullptr;
            throw;
        }
        rt_ = nullptr;
    }
    ~ShardedTableQuartetScoreComputer() { release(); }
    ShardedTableQuartetScoreComputer(const ShardedTableQuartetScoreComputer &) = delete;
    ShardedTableQuartetScoreComputer &operator=(const ShardedTableQuartetScoreComputer &) = delete;

    ShardedTableScores scores;

private:
    struct PerGpu {
        int dev = 0;
        std::vector<size_t> shards;            // shard numbers this GPU owns (s mod N == g), in order
        std::vector<int64_t> sums, mins, extra; // host accumulators of this GPU's shards
        std::vector<std::string> spilled;      // spill = host: the finished shards
        qs_ctx *resident = nullptr;            // the one shard of a GPU that owns exactly one: stays on the device
        int64_t *d_sums = nullptr, *d_min = nullptr, *d_cand = nullptr;
        void *table_buf = nullptr;             // a GPU with several shards: ONE allocation of the largest, attached to each in turn
        uint64_t table_buf_bytes = 0;          // (allocating and freeing 34 GB per shard cost ~1 s of the ~1.05 s a shard took)
    };
    RefFlat ref_;
    DeviceOptions opt_;
    uint32_t bits_;
    int G_ = 1;
    bool spill_host_ = false;
    size_t P_ = 0, m_ = 0;
    const qs_ref_tree *rt_ = nullptr;
    std::vector<std::pair<uint32_t, uint32_t>> shards_;
    std::vector<BatchFlat> batches_;
    std::vector<PerGpu> gpu_;
    std::mutex io_;

    // one host thread per GPU (the calling thread takes the last one); the first exception is rethrown after all have ended
    template <typename F> void run_on_all(F f) {
        std::vector<std::exception_ptr> errs(gpu_.size());
        auto body = [&](size_t g) { try { f(gpu_[g]); } catch (...) { errs[g] = std::current_exception(); } };
        std::vector<std::thread> pool;
        for (size_t g = 0; g + 1 < gpu_.size(); ++g) pool.emplace_back(body, g);
        body(gpu_.size() - 1);
        for (auto &th : pool) th.join();
        for (auto &e : errs) if (e) std::rethrow_exception(e);
    }
    void release() {
        for (PerGpu &w : gpu_) {
            (void)hipSetDevice(w.dev);
            if (w.resident) { qs_destroy(w.resident); w.resident = nullptr; }
            (void)hipFree(w.d_sums); (void)hipFree(w.d_min); (void)hipFree(w.d_cand); (void)hipFree(w.table_buf);
            w.d_sums = w.d_min = w.d_cand = nullptr; w.table_buf = nullptr; w.table_buf_bytes = 0;
        }
    }
    static void hip_ok(hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }

    void round1(PerGpu &w) {
        if (w.shards.empty()) return;
        hip_ok(hipSetDevice(w.dev), "hipSetDevice");
        if (hipMalloc((void **)&w.d_sums, P_ * 3 * 8) != hipSuccess || hipMalloc((void **)&w.d_min, P_ * 8) != hipSuccess ||
            hipMalloc((void **)&w.d_cand, P_ * QS_SCORE_CAND_SLOTS * 8) != hipSuccess)
            throw std::runtime_error("Insufficient memory!");
        std::vector<int64_t> part_s(P_ * 3), part_m(P_);
        for (size_t i = 0; i < w.shards.size(); ++i) {
            const size_t k = w.shards[i];
            qs_ctx *ctx = open_shard(k, w);
            trace_mark(opt_, "shard: context + table ready");
            try {
                count_all(ctx);
                trace_mark(opt_, "shard: counted");
                if (qs_score_pass1(ctx, rt_, w.d_sums, w.d_min) != QS_OK || qs_sync(ctx) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                hip_ok(hipMemcpy(part_s.data(), w.d_sums, P_ * 3 * 8, hipMemcpyDeviceToHost), "copy of the score sums");
                hip_ok(hipMemcpy(part_m.data(), w.d_min, P_ * 8, hipMemcpyDeviceToHost), "copy of the score minima");
                for (size_t j = 0; j < P_ * 3; ++j) w.sums[j] = (int64_t)((uint64_t)w.sums[j] + (uint64_t)part_s[j]);
                for (size_t j = 0; j < P_; ++j) w.mins[j] = std::min(w.mins[j], part_m[j]);
                trace_mark(opt_, "shard: score pass 1 done");
                const uint64_t bytes = qs_table_bytes(ctx);
                if (w.shards.size() == 1) { w.resident = ctx; ctx = nullptr; }
                else if (spill_host_) {
                    w.spilled[i].resize((size_t)bytes);
                    if (qs_table_download(ctx, &w.spilled[i][0], w.spilled[i].size()) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                }
                std::lock_guard<std::mutex> lk(io_);
                std::cout << "shard " << k << " on GPU " << w.dev << ": largest id in [" << shards_[k].first << ", " << shards_[k].second << "), " << bytes << " bytes" << std::endl;
            } catch (...) { if (ctx) qs_destroy(ctx); throw; }
            if (ctx) qs_destroy(ctx);
        }
    }

    void round2(PerGpu &w, const std::vector<int64_t> &mins, std::vector<int64_t> &cand) {
        if (w.shards.empty()) return;
        hip_ok(hipSetDevice(w.dev), "hipSetDevice");
        hip_ok(hipMemcpy(w.d_min, mins.data(), P_ * 8, hipMemcpyHostToDevice), "copy of the minima");
        for (size_t i = 0; i < w.shards.size(); ++i) {
            const size_t k = w.shards[i];
            qs_ctx *ctx = w.resident;
            w.resident = nullptr;
            if (!ctx) {
                ctx = open_shard(k, w);
                try {
                    if (spill_host_) {
                        if (qs_table_upload(ctx, w.spilled[i].data(), w.spilled[i].size()) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                        std::string().swap(w.spilled[i]);
                        // the uploaded table holds the counts of all m trees: size the device QIC's log table for them
                        (void)qs_set_tuning(ctx, QS_TUNE_TABLE_TREES, (uint64_t)m_);
                    } else count_all(ctx);
                } catch (...) { qs_destroy(ctx); throw; }
            }
            try {
                if (qs_score_pass2(ctx, rt_, w.d_min, w.d_cand) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                int64_t *list = nullptr;
                uint64_t cnt = 0;
                if (qs_score_overflow(ctx, rt_, w.d_min, w.d_cand, &list, &cnt) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                if (cnt) { w.extra.insert(w.extra.end(), list, list + 4 * cnt); qs_free_host(list); }
                hip_ok(hipMemcpy(cand.data() + k * P_ * QS_SCORE_CAND_SLOTS, w.d_cand, P_ * QS_SCORE_CAND_SLOTS * 8, hipMemcpyDeviceToHost), "copy of the candidates");
            } catch (...) { qs_destroy(ctx); throw; }
            qs_destroy(ctx);
        }
    }

    // A GPU that owns one shard lets the context allocate it (it stays resident); a GPU with several allocates the largest
    // of them once and attaches that buffer to one shard context after the other.
    qs_ctx *open_shard(size_t k, PerGpu &w) {
        qs_ctx *ctx = nullptr;
        if (qs_create(&ctx, (uint32_t)ref_.names.size(), bits_, QS_FLAG_NONE, w.dev, nullptr, shards_[k].first, shards_[k].second) != QS_OK)
            throw std::runtime_error(qs_last_error(nullptr));
        int rc;
        if (w.shards.size() == 1) rc = qs_table_alloc(ctx);
        else {
            if (!w.table_buf) {
                auto c4 = [](uint64_t x) { return x < 4 ? (uint64_t)0 : x * (x - 1) * (x - 2) * (x - 3) / 24; };
                uint64_t most = 0;
                for (size_t s : w.shards) most = std::max(most, (c4(shards_[s].second) - c4(shards_[s].first)) * 3 * (bits_ / 8));
                most = (most + 3) & ~(uint64_t)3;
                if (hipMalloc(&w.table_buf, (size_t)most) != hipSuccess) { (void)hipGetLastError(); qs_destroy(ctx); throw std::runtime_error("Insufficient memory!"); }
                w.table_buf_bytes = most;
            }
            rc = qs_table_attach(ctx, w.table_buf, w.table_buf_bytes);
            if (rc == QS_OK) rc = qs_table_clear(ctx);
        }
        if (rc != QS_OK) { std::string e = qs_last_error(ctx); qs_destroy(ctx); throw std::runtime_error(e); }
        return ctx;
    }
    void count_all(qs_ctx *ctx) {
        const bool want_ranges = (opt_.algo & 0xFFu) == QS_ALGO_SCATTER;
        std::vector<qs_device_batch *> in_flight;
        try {
            for (const BatchFlat &b : batches_) {
                qs_tree_batch hb;
                hb.n_trees = b.n_trees; hb.leaf_off = b.leaf_off.data(); hb.leaf_ids = b.leaf_ids.data(); hb.adj_depth = b.adj_depth.data();
                hb.node_off = want_ranges ? b.node_off.data() : nullptr; hb.rng_off = want_ranges ? b.rng_off.data() : nullptr;
                hb.ranges = b.ranges.data();
                if (in_flight.size() == 2) { qs_batch_free(ctx, in_flight.front()); in_flight.erase(in_flight.begin()); }
                qs_device_batch *db = nullptr;
                if (qs_batch_upload(ctx, &hb, &db) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                in_flight.push_back(db);
                if (qs_count_batch(ctx, db, opt_.algo) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
            }
            if (qs_sync(ctx) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
        } catch (...) {
            (void)qs_sync(ctx);
            for (auto *db : in_flight) qs_batch_free(ctx, db);
            throw;
        }
        for (auto *db : in_flight) qs_batch_free(ctx, db);
    }
};

} // namespace qsh
