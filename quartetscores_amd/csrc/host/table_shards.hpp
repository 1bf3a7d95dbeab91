// table_shards.hpp -- a count table larger than the device's memory on ONE GPU (`QuartetScores --table-shards K`).
//
// The reference keeps its table in host RAM and once planned an STXXL external-memory vector for tables beyond it
// (quartet_lookup_table.hpp:3,11-13,218-222). Here the table is cut into K shards by the largest
// taxon id (contiguous rank ranges, because the leading term of the rank is C(s3,4): quartet_lookup_table.hpp:161-165,
// the same cut the 8-GPU table-sharded mode uses) and the shards pass through the device one after the other:
//   round 1, per shard: count ALL evaluation trees into the shard (the flattened batches stay in host memory), score
//            pass 1, add the per-node-pair sums / take the minima on the host, then either copy the shard to host memory
//            (spill = host) or drop it (spill = recount);
//   round 2, per shard: bring the shard back (upload, or count again), score pass 2 against the GLOBAL minima, collect the
//            candidate slots and overflow lists;
//   qs_score_finish on the host, exactly as for shards on several GPUs.
// Same scores as the unsharded run (tests/test_cli.py compares the output files).
#pragma once

#include "QuartetScoreComputer.hpp"

#include <hip/hip_runtime_api.h>

#include <fstream>

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

    ShardedTableQuartetScoreComputer(Tree const &refTree, const std::string &evalTreesPath, size_t m, uint32_t count_bits, int n_shards,
                                     Spill spill, DeviceOptions opt)
        : ref_(flatten_reference(refTree)), opt_(opt), bits_(count_bits) {
        const uint32_t n = (uint32_t)ref_.names.size();
        if (n_shards < 1) throw std::runtime_error("--table-shards needs a positive number");
        std::cout << "There are " << m << " evaluation trees.\n";
        std::cout << "The reference tree has " << n << " taxa.\n";
        const auto t0 = std::chrono::steady_clock::now();
        // shard bounds in the largest id d, balanced by C(d,4); empty shards (small n) are dropped
        auto c4 = [](uint64_t x) { return x < 4 ? (uint64_t)0 : x * (x - 1) * (x - 2) * (x - 3) / 24; };
        const uint64_t total = c4(n);
        std::vector<uint32_t> bounds(1, 0);
        for (int r = 1; r < n_shards; ++r) {
            const uint64_t target = total / (uint64_t)n_shards * (uint64_t)r;
            uint32_t d = bounds.back();
            while (d < n && c4(d) < target) ++d;
            bounds.push_back(d);
        }
        bounds.push_back(n);
        for (size_t k = 0; k + 1 < bounds.size(); ++k)
            if (c4(bounds[k + 1]) > c4(bounds[k])) shards_.push_back({bounds[k], bounds[k + 1]});
        const uint64_t table_bytes = total * 3 * (bits_ / 8);
        if (spill == SPILL_AUTO) {
            const uint64_t avail = host_mem_available();
            spill = (avail && table_bytes + (table_bytes >> 3) < avail) ? SPILL_HOST : SPILL_RECOUNT;
        }
        spill_host_ = spill == SPILL_HOST;
        std::cout << "Counting in " << shards_.size() << " table shard(s) by largest taxon id on one GPU; finished shards are "
                  << (spill_host_ ? "kept in host memory" : "dropped and counted again for the second scoring pass") << ".\n";
        // the evaluation trees, flattened once
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
        const size_t K = shards_.size();
        std::vector<int64_t> sums(P * 3, 0), mins(P, INT64_MAX), part_s(P * 3), part_m(P), cand(K * P * QS_SCORE_CAND_SLOTS), extra;
        std::vector<std::string> spilled(spill_host_ ? K : 0);
        int64_t *d_sums = nullptr, *d_min = nullptr, *d_cand = nullptr;
        if (hipSetDevice(opt_.device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        auto dev_free = [&]() { (void)hipFree(d_sums); (void)hipFree(d_min); (void)hipFree(d_cand); d_sums = d_min = d_cand = nullptr; };
        qs_ctx *ctx = nullptr;
        try {
            if (hipMalloc((void **)&d_sums, P * 3 * 8) != hipSuccess || hipMalloc((void **)&d_min, P * 8) != hipSuccess ||
                hipMalloc((void **)&d_cand, P * QS_SCORE_CAND_SLOTS * 8) != hipSuccess)
                throw std::runtime_error("Insufficient memory!");
            for (size_t k = 0; k < K; ++k) {       // round 1
                ctx = open_shard(k);
                count_all(ctx);
                if (qs_score_pass1(ctx, &rt, d_sums, d_min) != QS_OK || qs_sync(ctx) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                if (hipMemcpy(part_s.data(), d_sums, P * 3 * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(part_m.data(), d_min, P * 8, hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("copy of the score accumulators failed");
                for (size_t i = 0; i < P * 3; ++i) sums[i] = (int64_t)((uint64_t)sums[i] + (uint64_t)part_s[i]);
                for (size_t i = 0; i < P; ++i) mins[i] = std::min(mins[i], part_m[i]);
                if (spill_host_) {
                    spilled[k].resize((size_t)qs_table_bytes(ctx));
                    if (qs_table_download(ctx, &spilled[k][0], spilled[k].size()) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                }
                std::cout << "shard " << k << ": largest id in [" << shards_[k].first << ", " << shards_[k].second << "), " << qs_table_bytes(ctx) << " bytes" << std::endl;
                qs_destroy(ctx); ctx = nullptr;
            }
            const auto t1 = std::chrono::steady_clock::now();
            std::cout << "lookup table size in bytes: " << table_bytes << "\n";
            std::cout << "Finished counting quartets.\nIt took: " << std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() << " microseconds." << std::endl;
            if (hipMemcpy(d_min, mins.data(), P * 8, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("copy of the minima failed");
            for (size_t k = 0; k < K; ++k) {       // round 2
                ctx = open_shard(k);
                if (spill_host_) {
                    if (qs_table_upload(ctx, spilled[k].data(), spilled[k].size()) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                    std::string().swap(spilled[k]);
                } else count_all(ctx);
                if (qs_score_pass2(ctx, &rt, d_min, d_cand) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                int64_t *list = nullptr;
                uint64_t cnt = 0;
                if (qs_score_overflow(ctx, &rt, d_min, d_cand, &list, &cnt) != QS_OK) throw std::runtime_error(qs_last_error(ctx));
                if (cnt) { extra.insert(extra.end(), list, list + 4 * cnt); qs_free_host(list); }
                if (hipMemcpy(cand.data() + k * P * QS_SCORE_CAND_SLOTS, d_cand, P * QS_SCORE_CAND_SLOTS * 8, hipMemcpyDeviceToHost) != hipSuccess)
                    throw std::runtime_error("copy of the candidates failed");
                qs_destroy(ctx); ctx = nullptr;
            }
            dev_free();
            const uint32_t flags = (opt_.qp_exact64 ? QS_SCORE_QP_EXACT64 : QS_SCORE_QP_WRAP32) | (opt_.root_as_edge ? QS_SCORE_ROOT_AS_EDGE : 0u);
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
            if (ctx) qs_destroy(ctx);
            dev_free();
            throw;
        }
    }
    ShardedTableQuartetScoreComputer(const ShardedTableQuartetScoreComputer &) = delete;
    ShardedTableQuartetScoreComputer &operator=(const ShardedTableQuartetScoreComputer &) = delete;

    ShardedTableScores scores;

private:
    RefFlat ref_;
    DeviceOptions opt_;
    uint32_t bits_;
    bool spill_host_ = false;
    std::vector<std::pair<uint32_t, uint32_t>> shards_;
    std::vector<BatchFlat> batches_;

    qs_ctx *open_shard(size_t k) {
        qs_ctx *ctx = nullptr;
        if (qs_create(&ctx, (uint32_t)ref_.names.size(), bits_, QS_FLAG_NONE, opt_.device, nullptr, shards_[k].first, shards_[k].second) != QS_OK)
            throw std::runtime_error(qs_last_error(nullptr));
        if (qs_table_alloc(ctx) != QS_OK) { std::string e = qs_last_error(ctx); qs_destroy(ctx); throw std::runtime_error(e); }
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
