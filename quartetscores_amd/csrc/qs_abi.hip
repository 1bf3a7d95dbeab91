// qs_abi.hip -- implementation of the C-ABI declared in include/quartetscores_hip.h.
//
// Host-side orchestration only: validation of the flattened trees, device memory, kernel
// dispatch, and the O(#node pairs) finalisation of the scores. All O(m*C(n,4)) and O(C(n,4))
// work runs in the HIP kernels of qs_count.hip / qs_score.hip. There is no CPU fallback.
#include "../../include/quartetscores_hip.h"
#include "qs_common.hpp"
#include "qs_internal.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <string>
#include <thread>
#include <vector>

using namespace qs;

struct EarlyPerm { std::vector<uint32_t> perm; uint32_t chunk = 0, cblock_in = 0, cgroup = 0; bool ok = false; };

struct qs_device_batch {
    DeviceBatch d;
    std::vector<uint32_t> fix_slot;   // depth clamp: slot (in the class-ordered batch) of the tree behind every correction unit, ascending
};

// the reference tree, flattened and indexed for scoring (build_ref)
struct RefHost {
    uint32_t n_nodes = 0, n = 0, n_inner = 0, root = 0;
    std::vector<int32_t> parent;
    std::vector<uint32_t> depth, nchild, inner_id, inner_node;
    std::vector<uint32_t> lca; // n*n
    std::vector<uint16_t> next; // n*n: for a < b the first a' > a with lca(a',b) != lca(a,b) (b if none): run ends of the score scan
    std::vector<uint32_t> leaf_node_in; // the caller's leaf_node array (cache key)
    std::vector<uint32_t> leaf_lo, leaf_cnt; // per node: its leaves are the lookup ids [leaf_lo, leaf_lo + leaf_cnt)
    bool bifurcating = false;
    bool root_deg2 = false;             // rooted Newick: the root has exactly two children
    std::vector<RootPairHost> root_pairs; // the node pairs (root, v) of such a tree, as the reference enumerates them
    uint32_t root_split = 0;            // ... and the number of taxa under the root's first child: lookup ids [0, root_split)
    uint64_t root_items = 0;
};

struct qs_ctx {
    uint32_t n = 0, count_bits = 32, flags = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t d_lo = 0, d_hi = 0;
    uint64_t rank_lo = 0, n_tuples = 0;
    void *table = nullptr;
    bool table_owned = false;
    uint64_t trees_counted = 0;
    uint32_t *wire_out = nullptr;   // qs_wire_attach: destination of QS_COUNT_WIRE16X2 (one word per tuple)
    uint64_t wire_trees = 0;        // trees accumulated in it
    // geometry
    uint32_t *dprefix = nullptr, *cprefix = nullptr;
    uint32_t n_dblk = 0, total_tiles = 0;
    uint32_t *dprefix3 = nullptr, *cprefix3 = nullptr; // tiling of count_bitslice3_kernel, binary batches (16x8 tiles, d-blocks counted down from d_hi)
    uint32_t total_tiles3 = 0;
    // (a,b)-major launch order of count_bitslice3_kernel's tiles (tile_order below): launch slot -> tile id, built on
    // first use; [0] binary tiling (16x8), [1] general / partial tiling (8x8). NULL = (d,c)-major (identity).
    uint32_t *perm[2] = {nullptr, nullptr};
    bool perm_built[2] = {false, false};
    std::future<EarlyPerm> perm_early;                 // the binary tiling's launch order, started by qs_create before its first HIP call
    uint32_t tile_chunk = 2, tile_cblock = 32;         // a-block (pairs) per chunk / c values per c-block; chunk 0 = (d,c)-major (round 3: 4 | 16 -> 2 | 32, -1.5 %)
    uint32_t tile_cgroup = 0;                          // > 1: c innermost in groups of this many (the waves of a workgroup share M[ab], M[bd])
    // binary tiling, cooperative workgroups (count_bitslice4_kernel): launch slots in groups of 4 tiles of one (a-blocks,
    // b-block, d-block) with consecutive c (bit 31 = shadow tile: takes part, does not store), and the list of the tiles
    // that stay with count_bitslice3_kernel (diagonal tiles, tiles without a second a-block)
    uint32_t *perm_coop = nullptr, *perm_rest = nullptr;
    uint32_t n_coop = 0, n_rest = 0;
    uint32_t tune_coop = 0;                            // QS_TUNE_COOP: 1 = cooperative workgroups on; 0 / 2 = off (the default: measured slower)
    std::vector<uint32_t> h_cp3, h_dp3, h_cp, h_dp1t;  // host copies of the prefix arrays
    uint32_t *dprefix1t = nullptr;                     // the same kernel on general / partial batches: 8x8 tiles (cprefix), d-blocks counted down
    uint32_t total_tiles1t = 0;
    // workspace
    void *panel = nullptr;
    size_t panel_bytes = 0;
    uint32_t *dev_flags = nullptr; // [0] counter overflow, [1] score flags, [3] two-cell wire format: tuple sum mismatch
    std::vector<hipEvent_t> evs;   // QS_COUNT_TIMED: evs[0] = start, then one event after every kernel launch
    std::vector<uint8_t> ev_kind;  // per event after evs[0]: 0 = panel build, 1 = count kernel, 2 = depth-clamp corrections (clamp_fix_kernel)
    uint32_t ev_used = 0;
    bool last_timed = false;
    // qs_set_tuning
    uint64_t tune_slice_bytes = 0; // 0 = automatic
    uint32_t tune_gather_impl = 0; // QS_IMPL_*
    uint32_t tune_panel_kernel = 0;
    uint32_t tune_cand_slots = kCand;  // candidate slots pass 2 fills per node pair (tests force overflows with fewer)
    int tune_score_kernel = 0;         // 0 = bundle kernel, 1 = scan kernel (QS_TUNE_SCORE_KERNEL)
    double tune_score_tol = 1e-12;     // pass 2 keeps triples whose device QIC is within this of the pair's minimum
    double *dev_logk = nullptr;    // log(k) table of the device QIC (qs_score.hip), tbl_n entries
    uint32_t tbl_n = 0;
    // scoring view (qs_score_set_view): tuples [view_rank_lo, view_rank_lo + view_n) in caller-owned device memory
    const void *view_table = nullptr;
    uint32_t view_bits = 0;
    uint64_t view_rank_lo = 0, view_n = 0;
    std::string variant;
    std::string err;
    // the flattened reference tree of the last scoring call, its LCA matrix on the device (qs_score_pass1 / pass2 /
    // raw_qic / finish of one scoring run all get the same tree: built once, not three times)
    RefHost *ref_cache = nullptr;
    uint32_t *ref_lca_dev = nullptr;
    uint16_t *ref_next_dev = nullptr;
    void *root_pairs_dev = nullptr;
    BundlePlan bundle[2];              // rounds of the score bundle kernel in pass 1 / pass 2 for [bundle_r0, bundle_r1) (plan_bundles)
    uint64_t bundle_r0[2] = {~0ull, ~0ull}, bundle_r1[2] = {~0ull, ~0ull};
    uint32_t *bundle_dev[2] = {nullptr, nullptr};    // plo[n] | pcnt[n] | rounds[2 R]
    int n_cu = 0;
    // tree batches go to the device through two pinned staging buffers on a copy stream of their own:
    // qs_batch_upload returns once the batch is in pinned memory, the copy of batch k+1 overlaps the counting of batch k
    hipStream_t copy_stream = nullptr;
    hipStream_t prep_stream = nullptr;           // qs_score_prepare: uploads of the scoring set-up go here instead of `stream` (which may hold count kernels)
    void *pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};   // recorded after the last copy out of the buffer
    unsigned pin_next = 0;
    std::vector<BatchSlab> slabs;                // device slabs of freed batches, for the next uploads
    // qs_last_score_ms: phases of the last qs_score call (host clock, passes 1 / 2 by HIP events on the stream)
    float score_ms[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t score_ev[3] = {nullptr, nullptr, nullptr};
    uint64_t table_trees_hint = 0;               // QS_TUNE_TABLE_TREES: trees behind an attached / uploaded / viewed table
    // single-read scoring (qs_score): pass 1 logs candidate (node pair, triple) records, score_log_kernel filters them
    unsigned long long *score_log = nullptr;     // log_cap records of 4 words + 1 word counter behind them
    uint64_t score_log_cap = 0, score_log_pairs = 0;
    bool log_valid = false;                      // the candidate log holds the last qs_score_pass1 of (log_table, log_rank_lo, log_n_tuples, log_ref)
    const void *log_table = nullptr, *log_ref = nullptr;
    uint64_t log_rank_lo = 0, log_n_tuples = 0;
    double log_tol = 0.0;
    uint32_t tune_score_passes = 0;              // QS_TUNE_SCORE_PASSES: 0 = automatic (default), 1 = two passes over the table, 2 = single read
    void *score_acc = nullptr, *score_acc_host = nullptr;   // qs_score's accumulators on the device + their pinned host copy (cached)
    size_t score_acc_cap = 0, score_acc_host_cap = 0;
    uint64_t last_score_estimate = 0;            // automatic single-read mode: predicted log records of the last qs_score (sample x S)
    uint64_t last_score_log = 0;                 // records the last single-read qs_score logged (0 = two passes were used)
    uint32_t tune_class_min = 1024;              // QS_TUNE_CLASS_MIN_TREES
    uint32_t tune_class_pct = 10;                // QS_TUNE_CLASS_PCT: a depth class below this share of the batch's trees is merged into the next one
    uint32_t tune_fuse = 1;                      // QS_TUNE_FUSE_CLASSES: classes of equal depth bits share ONE launch of the count kernel (qs_count_fused.hip)
    uint32_t tune_clamp_ppm = 20;                // QS_TUNE_DEPTH_CLAMP: corrections a tree may cost per depth bit it saves, in millionths of C(n,4); 0 = no clamp
    uint32_t tune_score_load = 0;                // QS_TUNE_SCORE_LOAD: 0 = a lane loads its row in 16-byte pieces, 1 = eight lanes load a row's chunk (LDS hand-over)
    uint32_t tune_score_dedupe = 1;              // QS_TUNE_SCORE_DEDUPE: the logging pass skips a quartet that repeats its node pair's last logged triple
    uint32_t tune_score_sample = 64u | 65536u;             // QS_TUNE_SCORE_SAMPLE: pre-pass of the single-read scoring (0 = none; S | by-round bit 16)
    uint64_t tune_score_log_cap = 0;             // QS_TUNE_SCORE_LOG_CAP: records the log may hold (0 = 8 M); tests force overflows
};

static thread_local std::string g_create_err;   // per thread: qs_create of several contexts may run concurrently (multi_gpu.hpp)
// Tree groups (panel elements along the tree axis) per sub-batch of a batch of n_total groups.
// A batch is counted slice by slice: panel build + count kernel per slice, the first slice stores into the table, the
// others read-modify-write it. qs_set_tuning(QS_TUNE_PANEL_SLICE_BYTES) fixes the slice size in bytes (tests, sweeps).
//  * bit-sliced kernel in (a,b)-major tile order (the default): what must stay in an XCD's 4 MB L2 is the working set of
//    its concurrent tiles, which grows with the number of groups G of the slice (about 40 KB x G at 20-byte elements), not
//    with the panel's size; against that every extra slice costs one read + one write of the table. Measured optimum
//    (profiles/r02_experiments.md): 128-150 groups at 256 taxa (2 GB table), ~105 at 512 taxa (34 GB table; 3 slices
//    for 10000 trees: 401 ms, 2: 415, 5: 432, unsliced 440), >= 80 at 1024 taxa (34 GB table shard). Rule: 128 groups,
//    slices balanced (313 groups -> 3 x 105, not 128 + 128 + 57), at most 2 GiB of panel.
//  * (d,c)-major order and the byte-SWAR kernel: the whole slice has to stay in the 256 MiB Infinity Cache while every
//    wave streams through it: 96 MiB, but at least 32 groups (a tile's fixed cost needs that many steps to amortise),
//    capped at 384 MiB (round-1 measurements: 512 taxa 0.60 s at 100 MB, 0.72 s at 50 MB, 0.80 s unsliced).
static uint32_t slice_groups(const qs_ctx *c, size_t group_bytes, uint32_t n_total, bool ab_major) {
    size_t g;
    bool balance = false;
    if (c->tune_slice_bytes) g = (size_t)c->tune_slice_bytes / group_bytes;
    else if (ab_major) {
        // Every slice costs a pass over the table (read + write of every tuple), a larger slice a larger L2 working set. Measured
        // on the round-3 kernel (profiles/r03_experiments.md 14): 512 taxa x 30000 trees best at 256 groups per slice (670 MB;
        // 128: +3.5 %, 512: +0.7 %), 256 taxa x 100000 best at 512 groups (334 MB; 128: +3 %, 1024: +1.5 %), a 34 GB shard of 1024
        // taxa x 5000 trees best in ONE slice of 157 groups (1.65 GB; two slices: +5 %): 256 groups, but at least 350 MB worth.
        g = std::max<size_t>(256, (350ull << 20) / group_bytes);
        g = std::min<size_t>(g, std::max<size_t>(1, (4096ull << 20) / group_bytes));
        balance = true;
    }
    else g = std::min<size_t>(std::max<size_t>(96ull << 20, 32 * group_bytes), 384ull << 20) / group_bytes;
    g = std::min<size_t>(std::max<size_t>(g, 1), std::max<uint32_t>(n_total, 1));
    // (round 5: up to 1.5 slices' worth stays ONE slice -- 313 groups at configs[2] since the depth clamp put all 10000 trees in one
    // class: 301.5 ms in one slice against 308.1 ms in two of 157, the second pass over the table costs more than the larger L2 set)
    if (balance && !c->tune_slice_bytes && n_total <= g + g / 2) g = std::max<uint32_t>(n_total, 1);
    if (balance) { const size_t slices = (n_total + g - 1) / g; g = (n_total + slices - 1) / std::max<size_t>(slices, 1); }
    return (uint32_t)std::max<size_t>(g, 1);
}

// QS_COUNT_TIMED: an event after every kernel launch of the call (created on demand, re-used by later calls)
static hipError_t mark(qs_ctx *c, int kind) {
    if (c->ev_used == c->evs.size()) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return rc;
        c->evs.push_back(e); c->ev_kind.push_back(0);
    }
    c->ev_kind[c->ev_used] = (uint8_t)kind;
    return hipEventRecord(c->evs[c->ev_used++], c->stream);
}

static int fail(qs_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}

#define QS_HIP(c, expr)                                                                                 \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail((c), e__ == hipErrorOutOfMemory ? QS_ERR_OOM : QS_ERR_HIP,                      \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                            \
    } while (0)

extern "C" const char *qs_version(void) { return "quartetscores_amd 0.1.0 (gfx950)"; }

extern "C" const char *qs_last_error(const qs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// (a,b)-major launch order of the tiles of count_bitslice3_kernel. The kernel decodes a tile id as (d-block k, c, tile
// inside c) -- the order in which the TABLE is contiguous. Walked in that order, the waves that run at the same time
// cover every (a,b) pair, so the private M[ab] loads (60 % of a wave's panel bytes) never hit in L2 and the launch is
// bound by the Infinity Cache (knock-out at 512 taxa: no panel loads = 45 % less time; profiles/r02_experiments.md).
// Here the launch slots are permuted: outermost the b-block Bk and a chunk of `chunk` a-blocks (pairs of a-blocks in
// the binary tiling) under it, then blocks of `cblock` values of c, then the d-blocks, then c inside the block,
// innermost the a-blocks of the chunk. Concurrent waves of an XCD (with xcd_remap each XCD walks a contiguous eighth of
// the slots) then share one chunk's M[ab] elements (8 b-rows x 16*chunk a's x the tree groups of the slice), the
// M[bd] elements of (Bk, d-block) and the M[xc] rows of the c-block, which stay in its 4 MB L2; per (c,d) a chunk still
// writes 8 contiguous runs of the table. Diagonal tiles follow at the end. 512 taxa x 10000 trees: 568 -> 401 ms,
// 256 taxa: -35 %. Shards with more than 2^26 tiles keep the (d,c)-major order (the slot array would be > 256 MB).
// Host part of the launch order: the (a,b)-major enumeration of the tiles of one tiling (bin: 16x8 tiles of count_bitslice3_kernel's
// two-column instances; else the 8x8 tiling) as launch slot -> tile id, checked to be a bijection. Pure host code (no context, no HIP):
// qs_create starts it on a helper thread BEFORE its first HIP call, so that in a fresh process the ~25 ms it takes at 512 taxa pass
// while the HIP runtime comes up (round 5: the CLI's first launch 25 ms earlier).
struct TilePermParams { uint32_t chunk, cblock_in, cgroup, d_hi, n_dblk, total; bool bin; };
static bool build_tile_perm(const TilePermParams &P, const std::vector<uint32_t> &cp, const std::vector<uint32_t> &dp, std::vector<uint32_t> &perm,
                            std::string &err) {
    const bool bin = P.bin;
    const uint32_t total = P.total, chunk = P.chunk, d_hi = P.d_hi, n_dblk = P.n_dblk;
    perm.clear();
    perm.reserve(total);
    auto T_of = [](uint32_t cc) { return (cc + 7) / 8; };
    const uint32_t cmax = d_hi >= 2 ? d_hi - 2 : 0;          // largest c of any tile
    const uint32_t Tmax = T_of(cmax);
    const uint32_t cblock = P.cblock_in ? P.cblock_in : cmax + 1;
    // off-diagonal tiles under b-block Bk. (Building the lists of several Bk on helper threads was tried: in the CLI, where 8 host
    // threads flatten the first batch at the same time, the set-up thread then was ready after 27-30 ms instead of 24.)
    auto emit_Bk = [&](uint32_t Bk, std::vector<uint32_t> &out) {
        // binary: tile Bk^2/4 + j = a-blocks (2j, 2j+1); general: tile C(Bk,2) + j = a-block j (unrank2 in the kernel)
        const uint32_t base = bin ? (Bk * Bk) / 4 : Bk * (Bk - 1) / 2;
        const uint32_t nj = bin ? ((Bk + 1) * (Bk + 1)) / 4 - base : Bk;
        const uint32_t c_lo = std::max(2u, 8 * Bk + 1);      // T(c) > Bk
        for (uint32_t j0 = 0; j0 < nj; j0 += chunk) {
            const uint32_t j1 = std::min(nj, j0 + chunk);
            for (uint32_t cb = c_lo; cb <= cmax; cb += cblock)
                for (uint32_t k = 0; k < n_dblk; ++k) {
                    const uint32_t d1 = d_hi - k * kDB;
                    if (P.cgroup <= 1) {
                        for (uint32_t cc = cb; cc < cb + cblock && cc + 1 < d1; ++cc) {
                            const uint32_t id = dp[k] + cp[cc] + base;
                            for (uint32_t j = j0; j < j1; ++j) out.push_back(id + j);
                        }
                    } else {
                        // c innermost in groups of `cgroup`: the waves of a workgroup (consecutive slots) then hold the SAME
                        // (a-blocks, b-block, d-block) and consecutive c -- their M[ab] and M[bd] elements are identical
                        for (uint32_t c4 = cb; c4 < cb + cblock && c4 + 1 < d1; c4 += P.cgroup)
                            for (uint32_t j = j0; j < j1; ++j)
                                for (uint32_t cc = c4; cc < c4 + P.cgroup && cc < cb + cblock && cc + 1 < d1; ++cc)
                                    out.push_back(dp[k] + cp[cc] + base + j);
                    }
                }
        }
    };
    for (uint32_t Bk = 1; Bk < Tmax; ++Bk) emit_Bk(Bk, perm);
    for (uint32_t kd = 0; kd < (Tmax + 1) / 2; ++kd)          // diagonal tiles (two diagonal blocks each)
        for (uint32_t k = 0; k < n_dblk; ++k) {
            const uint32_t d1 = d_hi - k * kDB;
            for (uint32_t cc = 2; cc + 1 < d1; ++cc) {
                const uint32_t T = T_of(cc);
                if (kd < (T + 1) / 2) perm.push_back(dp[k] + cp[cc] + (bin ? (T * T) / 4 : T * (T - 1) / 2) + kd);
            }
        }
    if (perm.size() != total) { err = "tile order: enumeration does not match the tiling"; return false; }
    {   // every tile exactly once: a tile listed twice would be counted by two waves (a race on its tuples), one left out never
        std::vector<uint64_t> seen((total + 63) / 64, 0);   // (a bit per tile: 350 KB at 512 taxa, stays in the host's L2)
        for (uint32_t id : perm) {
            if (id >= total || ((seen[id >> 6] >> (id & 63)) & 1ull)) { err = "tile order: launch permutation is not a bijection"; return false; }
            seen[id >> 6] |= 1ull << (id & 63);
        }
    }
    return true;
}

static int tile_order(qs_ctx *c, int which, const uint32_t **out) {
    *out = nullptr;
    if (c->perm_built[which]) { *out = c->perm[which]; return QS_OK; }
    c->perm_built[which] = true;
    const bool bin = which == 0;
    const uint32_t total = bin ? c->total_tiles3 : c->total_tiles1t;
    const uint32_t chunk = c->tile_chunk;
    if (chunk == 0 || total == 0 || total > (1u << 26)) {
        if (bin && c->perm_early.valid()) c->perm_early.wait();
        return QS_OK;
    }
    const uint32_t d_hi = c->d_hi, n_dblk = c->n_dblk;
    const std::vector<uint32_t> &cp = bin ? c->h_cp3 : c->h_cp, &dp = bin ? c->h_dp3 : c->h_dp1t;
    const TilePermParams P{chunk, c->tile_cblock, c->tile_cgroup, d_hi, n_dblk, total, bin};
    std::vector<uint32_t> perm;
    std::string perr;
    bool have = false;
    if (bin && c->perm_early.valid()) {       // started by qs_create beside the HIP start-up
        EarlyPerm ep = c->perm_early.get();
        if (ep.ok && ep.chunk == P.chunk && ep.cblock_in == P.cblock_in && ep.cgroup == P.cgroup) { perm.swap(ep.perm); have = true; }
    }
    if (!have && !build_tile_perm(P, cp, dp, perm, perr)) return fail(c, QS_ERR_STATE, perr);
    auto T_of = [](uint32_t cc) { return (cc + 7) / 8; };
    const uint32_t cmax = d_hi >= 2 ? d_hi - 2 : 0;
    const uint32_t Tmax = T_of(cmax);
    const uint32_t cblock = c->tile_cblock ? c->tile_cblock : cmax + 1;
    // Off unless asked for (QS_TUNE_COOP = 1): measured on MI355X the real barrier per 32-tree step costs more than the
    // shared loads save (512 taxa x 10000 trees: 368 ms against 354 ms; profiles/r03_experiments.md)
    if (bin && c->tune_coop == 1) {
        // Cooperative launch order: same nesting ((a,b)-major: b-block, chunk of a-block pairs, c-block, d-block), but
        // innermost FOUR consecutive c of one a-block pair -- one workgroup of count_bitslice4_kernel. A group short of
        // four valid c is filled with shadow copies of its first tile.
        std::vector<uint32_t> coop, rest;
        coop.reserve(total + total / 8);
        for (uint32_t Bk = 1; Bk < Tmax; ++Bk) {
            const uint32_t base = (Bk * Bk) / 4;
            const uint32_t nj = ((Bk + 1) * (Bk + 1)) / 4 - base;     // tiles under this b-block; the last one lacks a2 when Bk is odd
            const uint32_t nj2 = Bk / 2;                              // pairs (2j, 2j+1) with 2j+1 < Bk
            const uint32_t c_lo = std::max(2u, 8 * Bk + 1);
            for (uint32_t j0 = 0; j0 < nj2; j0 += chunk) {
                const uint32_t j1 = std::min(nj2, j0 + chunk);
                for (uint32_t cb = c_lo; cb <= cmax; cb += cblock)
                    for (uint32_t k = 0; k < n_dblk; ++k) {
                        const uint32_t d1 = d_hi - k * kDB;
                        for (uint32_t j = j0; j < j1; ++j)
                            for (uint32_t c4 = cb; c4 < cb + cblock && c4 + 1 < d1; c4 += 4) {
                                const uint32_t first = dp[k] + cp[c4] + base + j;
                                for (uint32_t cc = c4; cc < c4 + 4; ++cc)
                                    coop.push_back((cc < cb + cblock && cc + 1 < d1) ? dp[k] + cp[cc] + base + j : (first | 0x80000000u));
                            }
                    }
            }
            if (nj > nj2)                                            // the tile with a single a-block
                for (uint32_t k = 0; k < n_dblk; ++k) {
                    const uint32_t d1 = d_hi - k * kDB;
                    for (uint32_t cc = c_lo; cc + 1 < d1; ++cc) rest.push_back(dp[k] + cp[cc] + base + nj2);
                }
        }
        for (uint32_t kd = 0; kd < (Tmax + 1) / 2; ++kd)
            for (uint32_t k = 0; k < n_dblk; ++k) {
                const uint32_t d1 = d_hi - k * kDB;
                for (uint32_t cc = 2; cc + 1 < d1; ++cc) {
                    const uint32_t T = T_of(cc);
                    if (kd < (T + 1) / 2) rest.push_back(dp[k] + cp[cc] + (T * T) / 4 + kd);
                }
            }
        {   // coop (without shadows) and rest together: every tile exactly once
            std::vector<uint8_t> seen(total, 0);
            size_t real = 0;
            auto take = [&](uint32_t id) { if (id >= total || seen[id]) return false; seen[id] = 1; ++real; return true; };
            for (uint32_t id : coop) if (!(id & 0x80000000u) && !take(id)) return fail(c, QS_ERR_STATE, "tile order: cooperative list is not a partition");
            for (uint32_t id : rest) if (!take(id)) return fail(c, QS_ERR_STATE, "tile order: rest list is not a partition");
            if (real != total || coop.size() % 4) return fail(c, QS_ERR_STATE, "tile order: cooperative + rest lists do not cover the tiling");
        }
        if (!coop.empty()) {
            if (hipMalloc(&c->perm_coop, coop.size() * 4) != hipSuccess || hipMalloc(&c->perm_rest, std::max<size_t>(rest.size(), 1) * 4) != hipSuccess)
                return fail(c, QS_ERR_OOM, "hipMalloc tile order (cooperative lists)");
            if (hipMemcpyAsync(c->perm_coop, coop.data(), coop.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                (!rest.empty() && hipMemcpyAsync(c->perm_rest, rest.data(), rest.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) ||
                hipStreamSynchronize(c->stream) != hipSuccess) return fail(c, QS_ERR_HIP, "memcpy tile order (cooperative lists)");
            c->n_coop = (uint32_t)coop.size(); c->n_rest = (uint32_t)rest.size();
        }
    }
    if (hipMalloc(&c->perm[which], perm.size() * 4) != hipSuccess) return fail(c, QS_ERR_OOM, "hipMalloc tile order");
    if (hipMemcpyAsync(c->perm[which], perm.data(), perm.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return fail(c, QS_ERR_HIP, "memcpy tile order");
    *out = c->perm[which];
    return QS_OK;
}

extern "C" int qs_set_tuning(qs_ctx *c, uint32_t key, uint64_t value) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c) return QS_ERR_ARG;
    switch (key) {
        case QS_TUNE_SCORE_CAND_SLOTS:
            if (value < 1 || value > (uint64_t)kCand) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_CAND_SLOTS takes 1..8");
            c->tune_cand_slots = (uint32_t)value; return QS_OK;
        case QS_TUNE_SCORE_KERNEL:
            if (value > 1) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_KERNEL takes 0 (bundle kernel) or 1 (scan kernel)");
            c->tune_score_kernel = (int)value; return QS_OK;
        case QS_TUNE_SCORE_TOL_EXP:
            if (value < 1 || value > 15) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_TOL_EXP takes 1..15 (tolerance 10^-value)");
            c->tune_score_tol = std::pow(10.0, -(double)value); return QS_OK;
        case QS_TUNE_TILE_ORDER:
            QS_HIP(c, hipSetDevice(c->device));
            QS_HIP(c, hipStreamSynchronize(c->stream));
            for (int w = 0; w < 2; ++w) { if (c->perm[w]) (void)hipFree(c->perm[w]); c->perm[w] = nullptr; c->perm_built[w] = false; }
            if (c->perm_coop) (void)hipFree(c->perm_coop);
            if (c->perm_rest) (void)hipFree(c->perm_rest);
            c->perm_coop = c->perm_rest = nullptr; c->n_coop = c->n_rest = 0;
            c->tile_chunk = (uint32_t)(value & 0xFFFF); c->tile_cblock = (uint32_t)((value >> 16) & 0xFFFF); c->tile_cgroup = (uint32_t)((value >> 32) & 0xFF);
            return QS_OK;
        case QS_TUNE_PANEL_SLICE_BYTES: c->tune_slice_bytes = value; return QS_OK;
        case QS_TUNE_TABLE_TREES: c->table_trees_hint = value; return QS_OK;
        case QS_TUNE_SCORE_SAMPLE: {
            const uint64_t S = value & 0xFFFFu;
            if (value >> 17 || (value && (S < 2 || (S & (S - 1))))) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_SAMPLE takes 0 or a power of two in [2, 32768], optionally | 65536 (whole rounds)");
            c->tune_score_sample = (uint32_t)value; return QS_OK;
        }
        case QS_TUNE_CLASS_MIN_TREES: c->tune_class_min = (uint32_t)std::min<uint64_t>(value, 0xFFFFFFFFull); return QS_OK;
        case QS_TUNE_DEPTH_CLAMP: c->tune_clamp_ppm = (uint32_t)std::min<uint64_t>(value, 1000000ull); return QS_OK;
        case QS_TUNE_FUSE_CLASSES: c->tune_fuse = value ? 1u : 0u; return QS_OK;
        case QS_TUNE_CLASS_PCT: if (value > 100) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_CLASS_PCT takes 0 .. 100"); c->tune_class_pct = (uint32_t)value; return QS_OK;
        case QS_TUNE_SCORE_LOAD:
            if (value > 3) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_LOAD takes 0 .. 3");
            c->tune_score_load = (uint32_t)value;
            c->bundle_r0[0] = c->bundle_r0[1] = c->bundle_r1[0] = c->bundle_r1[1] = ~0ull;   // the rounds are planned per wave count, which the load mode sets
            return QS_OK;
        case QS_TUNE_SCORE_DEDUPE: c->tune_score_dedupe = value ? 1u : 0u; return QS_OK;
        case QS_TUNE_SCORE_LOG_CAP:
            if (value > (1ull << 26)) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_LOG_CAP takes at most 2^26 records");
            c->tune_score_log_cap = value; return QS_OK;
        case QS_TUNE_SCORE_PASSES:
            if (value > 2) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_SCORE_PASSES takes 0 (automatic), 1 (two passes) or 2 (single read)");
            c->tune_score_passes = (uint32_t)value; return QS_OK;
        case QS_TUNE_COOP:
            if (value > 2) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_COOP takes 0 (default: off), 1 (on) or 2 (off)");
            if (c->tune_coop != (uint32_t)value) {   // the launch lists depend on it: rebuild on next use
                QS_HIP(c, hipSetDevice(c->device));
                QS_HIP(c, hipStreamSynchronize(c->stream));
                if (c->perm[0]) (void)hipFree(c->perm[0]);
                c->perm[0] = nullptr; c->perm_built[0] = false;
                if (c->perm_coop) (void)hipFree(c->perm_coop);
                if (c->perm_rest) (void)hipFree(c->perm_rest);
                c->perm_coop = c->perm_rest = nullptr; c->n_coop = c->n_rest = 0;
            }
            c->tune_coop = (uint32_t)value; return QS_OK;
        case QS_TUNE_GATHER_IMPL:
            if (value > QS_IMPL_BITSLICE) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_GATHER_IMPL takes QS_IMPL_AUTO / _SWAR / _BITSLICE");
            c->tune_gather_impl = (uint32_t)value; return QS_OK;
        case QS_TUNE_PANEL_KERNEL:
            if (value > 1) return fail(c, QS_ERR_ARG, "qs_set_tuning: QS_TUNE_PANEL_KERNEL takes 0 (automatic) or 1 (general builder)");
            c->tune_panel_kernel = (uint32_t)value; return QS_OK;
        default: return fail(c, QS_ERR_ARG, "qs_set_tuning: unknown key");
    }
}

extern "C" int qs_create(qs_ctx **out, uint32_t n_taxa, uint32_t count_bits, uint32_t flags, int device, void *stream,
                         uint32_t d_lo, uint32_t d_hi) {
    if (!out) return fail(nullptr, QS_ERR_ARG, "qs_create: out is NULL");
    *out = nullptr;
    if (n_taxa < 4 || n_taxa > 4096) return fail(nullptr, QS_ERR_ARG, "qs_create: n_taxa must be in [4, 4096]");
    if (count_bits != 16 && count_bits != 32) return fail(nullptr, QS_ERR_ARG, "qs_create: count_bits must be 16 or 32");
    if (d_lo == 0 && d_hi == 0) d_hi = n_taxa;
    if (d_hi > n_taxa || d_lo >= d_hi) return fail(nullptr, QS_ERR_ARG, "qs_create: bad shard [d_lo, d_hi)");
    // Everything that needs no device first: the tile geometry and -- for large tilings, on a helper thread -- the launch order of the
    // binary tiling (build_tile_perm: ~25 ms at 512 taxa), so that in a fresh process it is computed while the first HIP call below
    // waits for the runtime to come up.
    qs_ctx *c = new qs_ctx();
    c->n = n_taxa; c->count_bits = count_bits; c->flags = flags; c->device = device;
    c->stream = (hipStream_t)stream;
    c->d_lo = d_lo; c->d_hi = d_hi;
    c->rank_lo = binom4(d_lo);
    c->n_tuples = binom4(d_hi) - binom4(d_lo);
    // tile geometry of the gather kernel
    std::vector<uint32_t> cp(n_taxa + 2, 0);
    for (uint32_t cc = 2; cc <= n_taxa; ++cc)
        cp[cc + 1] = cp[cc] + gather_tiles_for_c(cc);
    // note: cp[c] = tiles of all c' < c (c' >= 2)
    const uint32_t d_start = std::max(d_lo, 3u);
    c->n_dblk = d_hi > d_start ? (d_hi - d_start + kDB - 1) / kDB : 0;
    std::vector<uint32_t> dp(c->n_dblk + 1, 0);
    uint64_t tiles64 = 0;
    for (uint32_t k = 0; k < c->n_dblk; ++k) {
        uint32_t d0 = d_start + k * kDB, d1 = std::min(d0 + (uint32_t)kDB, d_hi);
        tiles64 += cp[d1 - 1]; // c in [2, d1-1)
        dp[k + 1] = (uint32_t)tiles64;
    }
    if (tiles64 >= (1ull << 31)) { delete c; return fail(nullptr, QS_ERR_UNSUPPORTED, "qs_create: shard too large for one launch; use a narrower [d_lo, d_hi)"); }
    c->total_tiles = dp[c->n_dblk];
    // the 16x8 tiles of count_bitslice3_kernel; block k = [max(d_start, d1 - 8), d1) with d1 = d_hi - 8k
    std::vector<uint32_t> cp3(n_taxa + 2, 0), dp3(c->n_dblk + 1, 0), dp1(c->n_dblk + 1, 0);
    {
        for (uint32_t cc = 2; cc <= n_taxa; ++cc) cp3[cc + 1] = cp3[cc] + bitslice3_tiles_for_c(cc);
        uint64_t t3 = 0;
        for (uint32_t k = 0; k < c->n_dblk; ++k) { t3 += cp3[d_hi - k * kDB - 1]; dp3[k + 1] = (uint32_t)t3; }
        if (t3 >= (1ull << 31)) { delete c; return fail(nullptr, QS_ERR_UNSUPPORTED, "qs_create: shard too large for one launch; use a narrower [d_lo, d_hi)"); }
        c->total_tiles3 = dp3[c->n_dblk];
        for (uint32_t k = 0; k < c->n_dblk; ++k) dp1[k + 1] = dp1[k] + cp[d_hi - k * kDB - 1]; // total equals total_tiles (< 2^31, checked above)
        c->h_cp3 = cp3; c->h_dp3 = dp3; c->h_cp = cp; c->h_dp1t = dp1;
        c->total_tiles1t = dp1[c->n_dblk];
    }
    if (c->total_tiles3 >= (1u << 20) && c->total_tiles3 <= (1u << 26) && c->tile_chunk) {
        const TilePermParams P{c->tile_chunk, c->tile_cblock, c->tile_cgroup, c->d_hi, c->n_dblk, c->total_tiles3, true};
        try {
            c->perm_early = std::async(std::launch::async, [P, cp3, dp3] {
                EarlyPerm ep; ep.chunk = P.chunk; ep.cblock_in = P.cblock_in; ep.cgroup = P.cgroup;
                std::string err;
                ep.ok = build_tile_perm(P, cp3, dp3, ep.perm, err);
                return ep;
            });
        } catch (...) { c->perm_early = std::future<EarlyPerm>(); }   // no thread to be had: tile_order builds it when asked
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        delete c;   // (waits for the helper thread: a std::async future blocks in its destructor)
        return fail(nullptr, QS_ERR_NO_DEVICE,
                    "qs_create: no HIP device (this library has no CPU fallback; it needs a gfx950 GPU)");
    }
    if (device < 0 || device >= ndev) { delete c; return fail(nullptr, QS_ERR_ARG, "qs_create: bad device ordinal"); }
    if (hipSetDevice(device) != hipSuccess) { delete c; return fail(nullptr, QS_ERR_HIP, "qs_create: hipSetDevice"); }
    // the kernels take d_lo as the first d of block 0
    c->d_lo = d_lo; // shard boundary for ranks
    auto cleanup = [&](int code, const std::string &m) { qs_destroy(c); return fail(nullptr, code, m); };
    if (hipMalloc(&c->cprefix, cp.size() * 4) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc cprefix");
    if (hipMalloc(&c->dprefix, dp.size() * 4) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc dprefix");
    if (hipMemcpy(c->cprefix, cp.data(), cp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(QS_ERR_HIP, "memcpy cprefix");
    if (hipMemcpy(c->dprefix, dp.data(), dp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(QS_ERR_HIP, "memcpy dprefix");
    if (hipMalloc(&c->dprefix1t, dp1.size() * 4) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc dprefix1t");
    if (hipMemcpy(c->dprefix1t, dp1.data(), dp1.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(QS_ERR_HIP, "memcpy dprefix1t");
    if (hipMalloc(&c->cprefix3, cp3.size() * 4) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc cprefix3");
    if (hipMalloc(&c->dprefix3, dp3.size() * 4) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc dprefix3");
    if (hipMemcpy(c->cprefix3, cp3.data(), cp3.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(QS_ERR_HIP, "memcpy cprefix3");
    if (hipMemcpy(c->dprefix3, dp3.data(), dp3.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(QS_ERR_HIP, "memcpy dprefix3");
    if (hipMalloc(&c->dev_flags, 16) != hipSuccess) return cleanup(QS_ERR_OOM, "hipMalloc flags");
    if (hipMemset(c->dev_flags, 0, 16) != hipSuccess) return cleanup(QS_ERR_HIP, "memset flags");
    *out = c;
    return QS_OK;
}

extern "C" void qs_destroy(qs_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->table && c->table_owned) (void)hipFree(c->table);
    if (c->panel) (void)hipFree(c->panel);
    if (c->dprefix) (void)hipFree(c->dprefix);
    if (c->cprefix) (void)hipFree(c->cprefix);
    if (c->dprefix3) (void)hipFree(c->dprefix3);
    for (int w = 0; w < 2; ++w) if (c->perm[w]) (void)hipFree(c->perm[w]);
    if (c->perm_coop) (void)hipFree(c->perm_coop);
    if (c->perm_rest) (void)hipFree(c->perm_rest);
    if (c->dev_logk) (void)hipFree(c->dev_logk);
    if (c->dprefix1t) (void)hipFree(c->dprefix1t);
    if (c->cprefix3) (void)hipFree(c->cprefix3);
    if (c->dev_flags) (void)hipFree(c->dev_flags);
    for (hipEvent_t e : c->evs) (void)hipEventDestroy(e);
    if (c->ref_lca_dev) (void)hipFree(c->ref_lca_dev);
    if (c->ref_next_dev) (void)hipFree(c->ref_next_dev);
    if (c->root_pairs_dev) (void)hipFree(c->root_pairs_dev);
    for (int w = 0; w < 2; ++w) if (c->bundle_dev[w]) (void)hipFree(c->bundle_dev[w]);
    for (int w = 0; w < 2; ++w) { if (c->pin_ev[w]) { (void)hipEventSynchronize(c->pin_ev[w]); (void)hipEventDestroy(c->pin_ev[w]); } if (c->pin[w]) (void)hipHostFree(c->pin[w]); }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (BatchSlab &sl : c->slabs) { (void)hipFree(sl.p); if (sl.last_use) (void)hipEventDestroy(sl.last_use); }
    for (hipEvent_t e : c->score_ev) if (e) (void)hipEventDestroy(e);
    if (c->score_log) (void)hipFree(c->score_log);
    if (c->score_acc) (void)hipFree(c->score_acc);
    if (c->score_acc_host) (void)hipHostFree(c->score_acc_host);
    delete c->ref_cache;
    delete c;
}

// Work the first qs_count_batch would otherwise do before its first launch, done ahead of time (the host calls this on its
// GPU-initialisation thread while the evaluation trees are still being parsed): the launch order of the bit-sliced count
// kernel for binary batches (12 M slots at 512 taxa: ~60 ms of host work + a 47 MB copy) and the pair-depth panel for a
// batch of n_trees_hint trees. Purely an optimisation: qs_count_batch builds whatever is missing. The reference does the
// equivalent set-up in the table's constructor (QuartetCounterLookup.hpp:245-273).
static size_t Stager_padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }   // = Stager::padded (batches, below)
static int prepare_staging(qs_ctx *c, uint64_t n_trees_hint);
extern "C" int qs_prepare(qs_ctx *c, uint64_t n_trees_hint) {
    if (!c) return QS_ERR_ARG;
    QS_HIP(c, hipSetDevice(c->device));
    // the staging allocations (two pinned buffers + a device slab: ~10 ms) on a helper thread beside the launch order (~15 ms of
    // host work): they touch different members of the context
    std::thread staging;
    if (n_trees_hint) {
        try { staging = std::thread([c, n_trees_hint] { (void)prepare_staging(c, n_trees_hint); }); }
        catch (...) { (void)prepare_staging(c, n_trees_hint); }   // (no thread to be had: do it here; nothing throws across the C boundary)
    }
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_staging{staging};
    const uint32_t *order = nullptr;
    int rc = tile_order(c, 0, &order);
    if (rc != QS_OK) return rc;
    if (n_trees_hint && !c->panel) {
        const size_t group_bytes = (size_t)binom2(c->n) * 5 * 4;   // 5 planes: the common depth class
        const uint32_t groups = slice_groups(c, group_bytes, (uint32_t)std::min<uint64_t>((n_trees_hint + 31) / 32, 1u << 20), order != nullptr);
        if (hipMalloc(&c->panel, (size_t)groups * group_bytes) == hipSuccess) c->panel_bytes = (size_t)groups * group_bytes;
        else { c->panel = nullptr; (void)hipGetLastError(); }       // not fatal here: the count reports it if it persists
    }
    return QS_OK;
}
static int prepare_staging(qs_ctx *c, uint64_t n_trees_hint) {
    if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); return QS_ERR_HIP; }
    {
        // both pinned staging buffers and one device slab for batches of that many full trees (qs_batch_upload's Stager
        // would allocate them on first use: hipHostMalloc + hipMalloc, tens of ms in front of the first count)
        const uint64_t nt = std::min<uint64_t>(n_trees_hint, 1u << 22);
        const size_t need = Stager_padded(((size_t)nt + 1) * 4) + 2 * Stager_padded((size_t)nt * c->n * 2) + Stager_padded((size_t)nt * 4);
        if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { c->copy_stream = nullptr; (void)hipGetLastError(); return QS_ERR_HIP; }
        for (int slot = 0; slot < 2; ++slot)
            if (c->pin_cap[slot] < need) {
                if (c->pin[slot]) { (void)hipHostFree(c->pin[slot]); c->pin[slot] = nullptr; c->pin_cap[slot] = 0; }
                const size_t cap = need + need / 4 + 4096;
                if (hipHostMalloc(&c->pin[slot], cap, hipHostMallocDefault) == hipSuccess) c->pin_cap[slot] = cap;
                else { c->pin[slot] = nullptr; (void)hipGetLastError(); }
            }
        bool have = false;
        for (const BatchSlab &sl : c->slabs) have = have || sl.cap >= need;
        if (!have && c->slabs.size() < 4) {
            BatchSlab sl;
            sl.cap = need + need / 4;
            if (hipMalloc(&sl.p, sl.cap) == hipSuccess) c->slabs.push_back(sl); else (void)hipGetLastError();
        }
    }
    // first use of the count kernels' code object (~5 ms of loading on MI355X: tools/modload_probe.py) here, beside the launch
    // order, instead of in front of the first count: one lookup of the quartet {0,1,2,3} on the copy stream
    if (c->table && c->n >= 4 && c->d_lo == 0) {
        void *scratch = nullptr;
        if (hipMalloc(&scratch, 64) == hipSuccess) {
            const uint16_t ids[4] = {0, 1, 2, 3};
            if (hipMemcpyAsync(scratch, ids, sizeof ids, hipMemcpyHostToDevice, c->copy_stream) == hipSuccess)
                (void)launch_lookup(c->copy_stream, c->n, c->d_lo, c->d_hi, c->rank_lo, c->table, (int)c->count_bits, 1, (const uint16_t *)scratch,
                                    (uint64_t *)((char *)scratch + 16));
            (void)hipStreamSynchronize(c->copy_stream);
            (void)hipFree(scratch);
        }
        (void)hipGetLastError();
    }
    return QS_OK;
}

// ---- table -------------------------------------------------------------------------------

extern "C" uint64_t qs_table_tuples(const qs_ctx *c) { return c ? c->n_tuples : 0; }
extern "C" uint64_t qs_table_bytes(const qs_ctx *c) { return c ? c->n_tuples * 3 * (c->count_bits / 8) : 0; }
extern "C" void *qs_table_device_ptr(const qs_ctx *c) { return c ? c->table : nullptr; }
extern "C" uint64_t qs_trees_counted(const qs_ctx *c) { return c ? c->trees_counted : 0; }

extern "C" int qs_table_alloc(qs_ctx *c) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c) return QS_ERR_ARG;
    QS_HIP(c, hipSetDevice(c->device));
    if (c->table && c->table_owned) { (void)hipFree(c->table); c->table = nullptr; }
    size_t bytes = (size_t)qs_table_bytes(c);
    size_t freeb = 0, total = 0;
    QS_HIP(c, hipMemGetInfo(&freeb, &total));
    if (bytes + (64u << 20) > freeb) return fail(c, QS_ERR_OOM, "Insufficient memory!");
    hipError_t e = hipMalloc(&c->table, bytes + 16);
    if (e != hipSuccess) { c->table = nullptr; return fail(c, QS_ERR_OOM, "Insufficient memory!"); }
    c->table_owned = true;
    c->trees_counted = 0;
    QS_HIP(c, hipMemsetAsync(c->table, 0, bytes + 16, c->stream));
    return QS_OK;
}

extern "C" int qs_table_attach(qs_ctx *c, void *device_ptr, uint64_t bytes) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c) return QS_ERR_ARG;
    if (!device_ptr) {
        // detach: the caller takes its buffer back (e.g. to free a full table once its reduce-scattered shard is the scoring view)
        if (bytes != 0) return fail(c, QS_ERR_ARG, "qs_table_attach: NULL pointer with a size (detach = NULL, 0)");
        if (c->table && c->table_owned) return fail(c, QS_ERR_STATE, "qs_table_attach: detach of a table the context owns (qs_table_alloc)");
        QS_HIP(c, hipSetDevice(c->device));
        QS_HIP(c, hipStreamSynchronize(c->stream));   // nothing of this context may still be writing the buffer
        c->table = nullptr; c->table_owned = false;
        return QS_OK;
    }
    // 16-bit cells are updated through their 32-bit word (packed half-word atomics of the scatter kernel, word-wise
    // collectives): the buffer must cover whole words
    const uint64_t need = (qs_table_bytes(c) + 3) & ~3ull;
    if (bytes < need) return fail(c, QS_ERR_ARG, "qs_table_attach: buffer smaller than qs_table_bytes() rounded up to a multiple of 4");
    if ((uintptr_t)device_ptr & 3) return fail(c, QS_ERR_ARG, "qs_table_attach: pointer must be 4-byte aligned");
    if (c->table && c->table_owned) (void)hipFree(c->table);
    c->table = device_ptr;
    c->table_owned = false;
    return QS_OK;
}

extern "C" int qs_table_clear(qs_ctx *c) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c || !c->table) return fail(c, QS_ERR_STATE, "qs_table_clear: no table");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, hipMemsetAsync(c->table, 0, (size_t)qs_table_bytes(c), c->stream));
    c->trees_counted = 0;
    return QS_OK;
}

extern "C" int qs_table_download(qs_ctx *c, void *host_dst, uint64_t bytes) {
    if (!c || !c->table) return fail(c, QS_ERR_STATE, "qs_table_download: no table");
    if (bytes > qs_table_bytes(c)) return fail(c, QS_ERR_ARG, "qs_table_download: too many bytes");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, hipMemcpyAsync(host_dst, c->table, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    QS_HIP(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

extern "C" int qs_table_upload(qs_ctx *c, const void *host_src, uint64_t bytes) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c || !c->table) return fail(c, QS_ERR_STATE, "qs_table_upload: no table");
    if (bytes > qs_table_bytes(c)) return fail(c, QS_ERR_ARG, "qs_table_upload: too many bytes");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, hipMemcpyAsync(c->table, host_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    QS_HIP(c, hipStreamSynchronize(c->stream));
    return QS_OK;
}

extern "C" int qs_table_pack16(qs_ctx *c, void *dst_device, uint64_t dst_bytes) {
    if (!c || !c->table || !dst_device) return fail(c, QS_ERR_STATE, "qs_table_pack16: no table / NULL destination");
    if (c->count_bits != 32) return fail(c, QS_ERR_ARG, "qs_table_pack16: the table already has 16-bit cells");
    const uint64_t cells = c->n_tuples * 3, need = ((cells + 1) / 2) * 4;
    if (dst_bytes < need) return fail(c, QS_ERR_ARG, "qs_table_pack16: destination smaller than " + std::to_string(need) + " bytes");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, launch_pack16(c->stream, c->table, dst_device, cells, c->dev_flags));
    return QS_OK;
}

extern "C" int qs_wire_attach(qs_ctx *c, void *dst_device, uint64_t dst_bytes) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c) return QS_ERR_ARG;
    if (!dst_device) { c->wire_out = nullptr; c->wire_trees = 0; return QS_OK; }
    if (dst_bytes < c->n_tuples * 4) return fail(c, QS_ERR_ARG, "qs_wire_attach: destination smaller than " + std::to_string(c->n_tuples * 4) + " bytes");
    c->wire_out = (uint32_t *)dst_device;
    c->wire_trees = 0;
    return QS_OK;
}

extern "C" int qs_table_pack16x2(qs_ctx *c, void *dst_device, uint64_t dst_bytes) {
    if (!c || !c->table || !dst_device) return fail(c, QS_ERR_STATE, "qs_table_pack16x2: no table / NULL destination");
    if (c->count_bits != 32) return fail(c, QS_ERR_ARG, "qs_table_pack16x2: needs a 32-bit table");
    if (c->trees_counted > 0xFFFFull) return fail(c, QS_ERR_OVERFLOW, "qs_table_pack16x2: more than 65535 trees counted");
    if (dst_bytes < c->n_tuples * 4) return fail(c, QS_ERR_ARG, "qs_table_pack16x2: destination smaller than " + std::to_string(c->n_tuples * 4) + " bytes");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, launch_pack16x2(c->stream, c->table, dst_device, c->n_tuples, (uint32_t)c->trees_counted, c->dev_flags, c->dev_flags + 3));
    return QS_OK;
}

extern "C" int qs_unpack16x2(qs_ctx *c, const void *src_device, uint64_t n_tuples, uint32_t total_trees, void *dst_device) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c || !src_device || !dst_device) return fail(c, QS_ERR_ARG, "qs_unpack16x2: NULL argument");
    if (total_trees > 0xFFFFu) return fail(c, QS_ERR_OVERFLOW, "qs_unpack16x2: more than 65535 trees");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, launch_unpack16x2(c->stream, src_device, dst_device, n_tuples, total_trees, c->dev_flags + 3));
    return QS_OK;
}

// Two-cell wire format with 32-bit cells: (n0, n1) per tuple, for batches of binary trees holding all taxa whose total
// over the ranks reaches 65536 trees (8 bytes per quartet on the wire instead of 12).
extern "C" int qs_table_pack32x2(qs_ctx *c, void *dst_device, uint64_t dst_bytes) {
    if (!c || !c->table || !dst_device) return fail(c, QS_ERR_STATE, "qs_table_pack32x2: no table / NULL destination");
    if (c->count_bits != 32) return fail(c, QS_ERR_ARG, "qs_table_pack32x2: needs a 32-bit table");
    if (c->trees_counted > 0xFFFFFFFFull) return fail(c, QS_ERR_OVERFLOW, "qs_table_pack32x2: more than 2^32 - 1 trees counted");
    if (dst_bytes < c->n_tuples * 8) return fail(c, QS_ERR_ARG, "qs_table_pack32x2: destination smaller than " + std::to_string(c->n_tuples * 8) + " bytes");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, launch_pack32x2(c->stream, c->table, dst_device, c->n_tuples, (uint32_t)c->trees_counted, c->dev_flags + 3));
    return QS_OK;
}

extern "C" int qs_unpack32x2(qs_ctx *c, const void *src_device, uint64_t n_tuples, uint64_t total_trees, void *dst_device) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c || !src_device || !dst_device) return fail(c, QS_ERR_ARG, "qs_unpack32x2: NULL argument");
    if (total_trees > 0xFFFFFFFFull) return fail(c, QS_ERR_OVERFLOW, "qs_unpack32x2: more than 2^32 - 1 trees");
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, launch_unpack32x2(c->stream, src_device, dst_device, n_tuples, (uint32_t)total_trees, c->dev_flags + 3));
    return QS_OK;
}

extern "C" int qs_sum_words(qs_ctx *c, void *dst_device, const void *const *src_device, uint32_t n_src, uint64_t n_words) {
    if (c) c->log_valid = false;   // (a table may change under a logged pass 1)
    if (!c || (n_words && n_src && (!dst_device || !src_device))) return fail(c, QS_ERR_ARG, "qs_sum_words: NULL argument");
    if (n_src > 15) return fail(c, QS_ERR_ARG, "qs_sum_words: at most 15 sources");
    if ((uintptr_t)dst_device & 15) return fail(c, QS_ERR_ARG, "qs_sum_words: the destination must be 16-byte aligned");
    for (uint32_t k = 0; k < n_src; ++k)
        if (!src_device[k] || ((uintptr_t)src_device[k] & 15)) return fail(c, QS_ERR_ARG, "qs_sum_words: sources must be non-NULL and 16-byte aligned");
    QS_HIP(c, hipSetDevice(c->device));
    if (c->n_cu == 0) { int v = 0; QS_HIP(c, hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device)); c->n_cu = std::max(1, v); }
    QS_HIP(c, launch_sum_words(c->stream, dst_device, src_device, n_src, n_words, c->n_cu));
    return QS_OK;   // asynchronous on the context's stream
}

// ---- issue probe: how fast does THIS device run the count kernel's instruction mix? ------------------------------------
// MI355X devices of one pool differ by several per cent in the clock they hold under a VALU-dense load (the guide measures up
// to 12 % between devices), more than a round of kernel work moves the headline. The probe runs the bare slot of the hot
// instance at B = 5 -- two magnitude comparisons of 6 planes sharing one operand (24 v_bitop3) + 4 v_bcnt with accumulate, all
// operands in registers -- at 4 waves per SIMD on every CU and returns nanoseconds per wave instruction and SIMD, the figure
// profiles/r03_valu_yardstick.txt calls the slot's cost (1.39-1.40 ns on the boxes it was written on). bench.py prints it
// beside its line so that lines from different boxes can be compared.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void issue_probe_kernel(const uint32_t *__restrict__ seed, uint32_t *__restrict__ out, uint32_t iters) {
    uint32_t l1[6], l2[6], r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        l1[k] = seed[(threadIdx.x * 19u + k) & 1023u];
        l2[k] = seed[(threadIdx.x * 23u + k + 6) & 1023u];
        r[k] = seed[(threadIdx.x * 29u + k + 12) & 1023u];
    }
    uint32_t g1 = 0, t1 = 0, g2 = 0, t2 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {       // the truth tables of cmp_planes (qs_count.hip): greater-than / less-than steps
                g1 = __builtin_amdgcn_bitop3_b32(l1[k], r[(k + u) % 6], g1, 0xb2);
                t1 = __builtin_amdgcn_bitop3_b32(l1[k], r[(k + u) % 6], t1, 0x8e);
                g2 = __builtin_amdgcn_bitop3_b32(l2[k], r[(k + u) % 6], g2, 0xb2);
                t2 = __builtin_amdgcn_bitop3_b32(l2[k], r[(k + u) % 6], t2, 0x8e);
            }
            c0 += (uint32_t)__builtin_popcount(g1); c1 += (uint32_t)__builtin_popcount(t1);
            c2 += (uint32_t)__builtin_popcount(g2); c3 += (uint32_t)__builtin_popcount(t2);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
}

extern "C" int qs_issue_probe(qs_ctx *c, uint32_t iterations, float *ns_per_instruction) {
    if (!c || !ns_per_instruction || iterations == 0) return fail(c, QS_ERR_ARG, "qs_issue_probe: bad argument");
    QS_HIP(c, hipSetDevice(c->device));
    if (c->n_cu == 0) { int v = 0; QS_HIP(c, hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device)); c->n_cu = std::max(1, v); }
    const uint32_t blocks = (uint32_t)c->n_cu * 4u;              // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    uint32_t *buf = nullptr;
    QS_HIP(c, hipMalloc((void **)&buf, (1024 + (size_t)blocks * 256) * 4));
    std::vector<uint32_t> host(1024);
    uint32_t x = 0x9E3779B9u;
    for (uint32_t &w : host) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w = x; }   // random operands: all-zero words let the clock rise
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMemcpyAsync(buf, host.data(), 4096, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.f;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(issue_probe_kernel, dim3(blocks), dim3(256), 0, c->stream, buf, buf + 1024, iterations);   // warm-up: clocks, code object
        e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) { hipLaunchKernelGGL(issue_probe_kernel, dim3(blocks), dim3(256), 0, c->stream, buf, buf + 1024, iterations); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(c, QS_ERR_HIP, std::string("qs_issue_probe: ") + hipGetErrorString(e));
    // per SIMD: 4 waves x iterations x 4 slots x 28 instructions
    *ns_per_instruction = ms * 1e6f / ((float)iterations * 4.f * 28.f * 4.f);
    return QS_OK;
}

// ---- batches -----------------------------------------------------------------------------

extern "C" void qs_batch_free(qs_ctx *c, qs_device_batch *b) {
    if (!b) return;
    if (c) (void)hipSetDevice(c->device);
    DeviceBatch &d = b->d;
    if (d.ready) { (void)hipEventSynchronize(d.ready); (void)hipEventDestroy(d.ready); }
    if (d.slab.p) {
        // kernels that read the batch may still be queued: remember where the compute stream stands and keep the slab
        bool kept = false;
        if (c && c->slabs.size() < 4) {
            if (!d.slab.last_use && hipEventCreateWithFlags(&d.slab.last_use, hipEventDisableTiming) != hipSuccess) d.slab.last_use = nullptr;
            if (d.slab.last_use && hipEventRecord(d.slab.last_use, c->stream) == hipSuccess) { c->slabs.push_back(d.slab); kept = true; }
        }
        if (!kept) { (void)hipFree(d.slab.p); if (d.slab.last_use) (void)hipEventDestroy(d.slab.last_use); }   // (hipFree waits for the device)
    }
    delete b;
}

// One batch's arrays are packed (256-byte aligned pieces) into a pinned staging buffer of the context and go to a device
// slab of the same layout with ONE copy on the copy stream. Slabs come from a small cache of the context: qs_batch_free
// records where the compute stream stands and hands the slab back, the next upload makes the copy stream wait for that
// point -- no hipMalloc / hipFree (a device-wide synchronisation) per batch, and no host thread ever waits for a kernel.
struct Stager {
    qs_ctx *c = nullptr; int slot = 0; size_t off = 0, need = 0; hipError_t err = hipSuccess;
    BatchSlab slab;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t begin(qs_ctx *ctx, size_t bytes) {
        c = ctx; need = std::max<size_t>(bytes, 256); slot = (int)(c->pin_next++ & 1u);
        hipError_t e = hipSuccess;
        if (!c->copy_stream) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !c->pin_ev[slot]) e = hipEventCreateWithFlags(&c->pin_ev[slot], hipEventDisableTiming);
        else if (e == hipSuccess) e = hipEventSynchronize(c->pin_ev[slot]);      // its previous contents are on the device
        if (e == hipSuccess && c->pin_cap[slot] < need) {
            if (c->pin[slot]) { (void)hipHostFree(c->pin[slot]); c->pin[slot] = nullptr; c->pin_cap[slot] = 0; }
            const size_t cap = need + need / 4 + 4096;
            e = hipHostMalloc(&c->pin[slot], cap, hipHostMallocDefault);
            if (e == hipSuccess) c->pin_cap[slot] = cap;
        }
        if (e != hipSuccess) return err = e;
        // device slab: the smallest cached one that is large enough, else a new one
        int best = -1;
        for (size_t i = 0; i < c->slabs.size(); ++i)
            if (c->slabs[i].cap >= need && (best < 0 || c->slabs[i].cap < c->slabs[(size_t)best].cap)) best = (int)i;
        if (best >= 0) { slab = c->slabs[(size_t)best]; c->slabs.erase(c->slabs.begin() + best); }
        else {
            slab = BatchSlab();
            slab.cap = need + need / 4;
            e = hipMalloc(&slab.p, slab.cap);
            if (e != hipSuccess) { slab = BatchSlab(); return err = e; }
        }
        if (slab.last_use) e = hipStreamWaitEvent(c->copy_stream, slab.last_use, 0);   // kernels of its previous batch
        return err = e;
    }
    template <typename T> T *put(const T *src, size_t count) {
        if (err != hipSuccess) return nullptr;
        const size_t bytes = count * sizeof(T);
        if (count) std::memcpy(static_cast<unsigned char *>(c->pin[slot]) + off, src, bytes);
        T *dev = reinterpret_cast<T *>(static_cast<unsigned char *>(slab.p) + off);
        off += padded(bytes);
        return dev;
    }
    // `ready` fires when the batch is on the device; the staging buffer is free again at the same moment
    hipError_t finish(hipEvent_t *ready) {
        if (err != hipSuccess) return err;
        err = hipMemcpyAsync(slab.p, c->pin[slot], std::max<size_t>(off, 1), hipMemcpyHostToDevice, c->copy_stream);
        if (err == hipSuccess) err = hipEventCreateWithFlags(ready, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventRecord(*ready, c->copy_stream);
        if (err == hipSuccess) err = hipEventRecord(c->pin_ev[slot], c->copy_stream);
        return err;
    }
};

// ---- depth clamp (qs_count.hip clamp_fix_kernel): the plan -- "plan_depth_clamp" in qs_count.hip's comments and DESIGN.md = clamp_cost +
// clamp_choice below and the class rules of qs_batch_upload ------------------------------------------------------------------------
// (tree, quartet) corrections a tree costs when its LCA depths are cut at `cut`: the quartets with at least three leaves in one
// maximal run of tour-adjacent LCA depths >= cut (a subtree below a node of depth `cut`): C(s,3)(L - s) + C(s,4) per run of s
// leaves. ~0 when a run is longer than the kernel's LDS table takes. runs (optional): (first position, leaves) of every run.
constexpr uint32_t kClampMaxRun = 128;  // = kFixMaxRun of qs_count.hip
static uint64_t clamp_cost(const uint16_t *adj, uint32_t L, uint32_t cut, std::vector<std::pair<uint32_t, uint32_t>> *runs) {
    uint64_t cost = 0;
    for (uint32_t i = 0; i + 1 < L;) {
        if (adj[i] < cut) { ++i; continue; }
        uint32_t j = i;
        while (j + 1 < L && adj[j] >= cut) ++j;
        const uint64_t s_ = (uint64_t)(j - i) + 1;
        if (s_ >= 3) {
            if (s_ > kClampMaxRun) return ~0ull;
            cost += binom3(s_) * (L - s_) + binom4(s_);
            if (runs) runs->emplace_back(i, (uint32_t)s_);
        }
        i = j;
    }
    return cost;
}
static uint32_t depth_class_of(uint32_t depth) {   // depth bits of a tree's class: 4 .. 10, 11 = beyond the bit-sliced instances
    uint32_t bb = 1;
    while ((1u << bb) <= depth) ++bb;
    return bb <= 4 ? 4u : (bb <= 10 ? bb : 11u);
}
// The cheapest class (depth bits) a tree may be counted in: its own, or a lower one when the corrections of the cut stay within
// `budget_per_bit` (tree, quartet) pairs per bit saved -- soft: the rule for every tree; hard (16 x): for the trees of a class
// that would otherwise be too small to pay for its own pass over the table.
static void clamp_choice(const uint16_t *adj, uint32_t L, uint32_t own_bits, uint64_t budget_per_bit, uint8_t &soft, uint8_t &hard) {
    soft = hard = (uint8_t)own_bits;
    if (budget_per_bit == 0 || own_bits <= 4) return;
    for (uint32_t tb = 4; tb < own_bits; ++tb) {
        const uint64_t cost = clamp_cost(adj, L, (1u << tb) - 1u, nullptr);
        if (cost == ~0ull) continue;
        const uint64_t bits_saved = own_bits - tb;
        if (hard == own_bits && cost <= 16 * budget_per_bit * bits_saved) hard = (uint8_t)tb;
        if (cost <= budget_per_bit * bits_saved) { soft = (uint8_t)tb; return; }
    }
}

// ---- classes of a batch (host-only: qs_batch_upload and qs_class_plan) --------------------------------------------------------------
// What validation knows about every tree: its deepest LCA, the cheapest kernel mode that is exact for it, and the lowest depth class
// the clamp budget allows it (soft: the rule for every tree; hard: 16 x, for the trees of a class too small for a pass of its own).
struct TreeFacts { std::vector<uint16_t> depth; std::vector<uint8_t> mode, soft, hard; };
// deepest adjacent LCA and "fully resolved" of one tree from its adjacent-LCA depth sequence
static void tree_shape(const uint16_t *adj, uint32_t L, std::vector<uint32_t> &stack, uint32_t &depth, bool &binary) {
    stack.clear();
    uint32_t nodes = 0, zeros = 0;
    depth = 0;
    for (uint32_t i = 0; i + 1 < L; ++i) {
        const uint32_t dd = adj[i];
        depth = std::max(depth, dd);
        if (dd == 0) ++zeros;
        while (!stack.empty() && stack.back() > dd) stack.pop_back();
        if (stack.empty() || stack.back() < dd) { stack.push_back(dd); ++nodes; }
    }
    binary = L >= 3 && (zeros == 1 || zeros == 2) && (L - 1 - zeros) == nodes - 1;
}
struct ClassPlan {
    uint32_t n_classes = 0;
    uint32_t class_mode[DeviceBatch::kMaxClasses] = {0}, class_bits[DeviceBatch::kMaxClasses] = {0}, class_end[DeviceBatch::kMaxClasses] = {0},
             class_max_depth[DeviceBatch::kMaxClasses] = {0};
    std::vector<uint32_t> order;       // slot -> tree (empty = identity: one class)
    std::vector<uint8_t> bits, mode;   // per tree: depth bits and kernel mode of the class it is counted in
};
// Classes = (kernel mode, depth bits) per TREE. The bit-sliced kernel's work per (quartet, 32 trees) grows with the bits B
// of the class (2(B+1)+2 instructions for full binary trees, B >= 4) and with what the trees may contain: binary trees with
// missing taxa 2(B+1)+5, multifurcating trees 3(B+1)+3, both 3(B+1)+7. One deep, one multifurcating or one incomplete tree
// must not put the whole batch on the dearest instance (the reference's loop is shape-independent,
// QuartetCounterLookup.hpp:65-106), so trees are counted class by class: B = 4 .. 10 and "deeper" (byte-SWAR kernel, 16-bit
// depths) within each of the four modes. Every class costs at least one more panel slice = one more pass over the table, so a
// class that holds fewer than max(class_min, class_pct %) of the trees joins one that is exact for it: first a whole mode joins
// a more general mode that is present (binary_full -> binary_partial, general_full or partial; binary_partial, general_full
// -> partial), then -- depth clamp -- a small deep class goes DOWN where the corrections allow it, else a depth class joins
// the next deeper one of its mode.
// fuse (QS_TUNE_FUSE_CLASSES, round 6): classes of equal depth bits (up to kFusedMaxBits) run as segments of ONE launch (qs_count_fused.hip),
// so what costs a table pass is a depth-bits GROUP over all modes, not a class: no mode joins a dearer mode any more, and "too small
// for a pass of its own" is asked of the group.
static void plan_classes(uint32_t n, uint32_t nt, const uint32_t *leaf_off, const uint16_t *adj_depth, const TreeFacts &F, uint32_t class_min,
                         uint32_t class_pct, uint32_t clamp_ppm, ClassPlan &P, bool fuse = false) {
    {
        constexpr uint32_t top_bits = 10, deep = 11;   // depth class id = depth bits; `deep` = beyond the bit-sliced instances
        constexpr uint32_t kModes = 4, kBits = 12;
        // Depth clamp: a tree may sit in a class BELOW its own depth bits -- the panel builders cut its depths at the class's
        // largest value and clamp_fix_kernel adds the quartets the cut ties (qs_count.hip) -- when that costs less than the
        // instructions it saves (eff_soft: 2 of 2(B+1)+2 per bit and quartet against a few thousand atomics per tree).
        uint32_t cnt[kModes][kBits] = {{0}}, mx[kModes][kBits] = {{0}};
        std::vector<uint8_t> cls(nt);
        for (uint32_t t = 0; t < nt; ++t) {
            cls[t] = F.soft[t];
            cnt[F.mode[t]][cls[t]]++;
        }
        const uint32_t small = std::max<uint32_t>(class_min, (uint32_t)((uint64_t)nt * class_pct / 100));
        uint32_t mode_map[kModes] = {0, 1, 2, 3};
        auto total = [&](uint32_t mo) { uint32_t s_ = 0; for (uint32_t bb = 0; bb < kBits; ++bb) s_ += cnt[mo][bb]; return s_; };
        auto join_mode = [&](uint32_t from, std::initializer_list<uint32_t> into) {
            const uint32_t tf = total(from);
            if (tf == 0 || tf >= small) return;
            for (uint32_t to : into) {
                if (total(to) == 0) continue;
                for (uint32_t bb = 0; bb < kBits; ++bb) { cnt[to][bb] += cnt[from][bb]; mx[to][bb] = std::max(mx[to][bb], mx[from][bb]); cnt[from][bb] = 0; }
                mode_map[from] = to;
                return;
            }
        };
        if (!fuse) {
            join_mode(MODE_BINARY_FULL, {MODE_BINARY_PARTIAL, MODE_GENERAL_FULL, MODE_PARTIAL});
            join_mode(MODE_BINARY_PARTIAL, {MODE_PARTIAL});
            join_mode(MODE_GENERAL_FULL, {MODE_PARTIAL});
        }
        auto final_mode = [&](uint32_t mo) { while (mode_map[mo] != mo) mo = mode_map[mo]; return mo; };
        // trees that share the launch of class (mo, bb): the class itself, or -- fused launches -- every class with these depth bits
        auto size_at = [&](uint32_t mo, uint32_t bb) {
            if (!fuse || bb > (uint32_t)kFusedMaxBits) return cnt[mo][bb];
            uint32_t s_ = 0;
            for (uint32_t m2 = 0; m2 < kModes; ++m2) s_ += cnt[m2][bb];
            return s_;
        };
        // a class too small for a pass of its own: its trees go DOWN to a class that is large enough where the 16-fold budget allows
        // it (what is left joins the next deeper class below, as before)
        if (clamp_ppm)
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t mo = final_mode(F.mode[t]), bb = cls[t];
                if (size_at(mo, bb) >= small || F.hard[t] >= bb) continue;
                for (uint32_t lo = bb - 1; lo >= F.hard[t] && lo >= 4; --lo)
                    if (size_at(mo, lo) >= small) { cnt[mo][bb]--; cnt[mo][lo]++; cls[t] = (uint8_t)lo; break; }
            }
        // ... and a class that is still small after that costs a launch of its own = one more pass over the table (10-20 ms at
        // 34 GB) plus the waves' fixed cost for a handful of trees: its trees go down at ANY finite price, as long as the sum stays
        // below what the pass would cost (5 % of C(n,4) corrections ~ 1.4e8 at 512 taxa)
        if (clamp_ppm)
            for (uint32_t mo = 0; mo < kModes; ++mo)
                for (uint32_t bb = top_bits; bb > 4; --bb) {
                    if (cnt[mo][bb] == 0 || size_at(mo, bb) >= small) continue;
                    // the class below to join: the nearest one that is large enough for a pass of its own, or -- in a batch whose
                    // classes are ALL small -- the nearest non-empty one that is at least as large as this one (joining the larger
                    // neighbour downwards beats the rule below, which would send the larger class up to this one's depth bits)
                    uint32_t lo = bb - 1;
                    while (lo > 4 && size_at(mo, lo) < small) --lo;
                    if (size_at(mo, lo) < small) {
                        lo = bb - 1;
                        while (lo > 4 && size_at(mo, lo) == 0) --lo;
                        if (size_at(mo, lo) == 0 || size_at(mo, lo) < size_at(mo, bb)) continue;   // nothing below that is worth joining
                    }
                    uint64_t total = 0;
                    bool finite = true;
                    std::vector<uint32_t> members;
                    for (uint32_t t = 0; t < nt && finite; ++t) {
                        if (final_mode(F.mode[t]) != mo || cls[t] != bb) continue;
                        const uint32_t base = leaf_off[t], L = leaf_off[t + 1] - base;
                        const uint64_t cost = clamp_cost(adj_depth + base, L, (1u << lo) - 1u, nullptr);
                        if (cost == ~0ull) finite = false; else { total += cost; members.push_back(t); }
                    }
                    if (!finite || total > binom4(n) / 20) continue;
                    for (uint32_t t : members) cls[t] = (uint8_t)lo;
                    cnt[mo][lo] += cnt[mo][bb]; cnt[mo][bb] = 0;
                }
        for (uint32_t t = 0; t < nt; ++t) { const uint32_t mo = final_mode(F.mode[t]); mx[mo][cls[t]] = std::max<uint32_t>(mx[mo][cls[t]], F.depth[t]); }
        uint32_t remap[kModes][kBits];
        for (uint32_t mo = 0; mo < kModes; ++mo) {
            for (uint32_t bb = 0; bb < kBits; ++bb) remap[mo][bb] = bb;
            for (uint32_t bb = 4; bb < top_bits; ++bb) {
                if (cnt[mo][bb] == 0 || size_at(mo, bb) >= small) continue;
                uint32_t up = bb + 1;
                while (up <= top_bits && size_at(mo, up) == 0) ++up;
                if (up > top_bits) continue;               // nothing above it among the bit-sliced classes of this mode (fused: of any mode)
                cnt[mo][up] += cnt[mo][bb]; mx[mo][up] = std::max(mx[mo][up], mx[mo][bb]); cnt[mo][bb] = 0; remap[mo][bb] = up;
            }
        }
        auto final_cls = [&](uint32_t mo, uint32_t k) { while (remap[mo][k] != k) k = remap[mo][k]; return k; };
        P.n_classes = 0;
        uint32_t run = 0, start[kModes][kBits] = {{0}};
        // launch order: the dearest instances first? No: by mode, then depth -- binary_full first (its first slice stores
        // instead of accumulating when the caller asks for QS_COUNT_OVERWRITE; any class may be the first)
        static const uint32_t mode_seq[kModes] = {MODE_BINARY_FULL, MODE_BINARY_PARTIAL, MODE_GENERAL_FULL, MODE_PARTIAL};
        for (uint32_t mi = 0; mi < kModes; ++mi) {
            const uint32_t mo = mode_seq[mi];
            for (uint32_t k = 4; k <= deep; ++k) {
                if (cnt[mo][k] == 0) continue;
                start[mo][k] = run; run += cnt[mo][k];
                P.class_mode[P.n_classes] = mo; P.class_bits[P.n_classes] = k; P.class_end[P.n_classes] = run; P.class_max_depth[P.n_classes] = mx[mo][k];
                P.n_classes++;
            }
        }
        P.bits.resize(nt); P.mode.resize(nt);
        for (uint32_t t = 0; t < nt; ++t) { const uint32_t mo = final_mode(F.mode[t]); P.mode[t] = (uint8_t)mo; P.bits[t] = (uint8_t)final_cls(mo, cls[t]); }
        P.order.clear();
        if (P.n_classes > 1) {
            P.order.resize(nt);
            for (uint32_t t = 0; t < nt; ++t) P.order[start[P.mode[t]][P.bits[t]]++] = t;
        }
    }
}

/* Host-only (no device call): the class plan of qs_batch_upload for the trees of `hb` on n_taxa taxa with the depth-clamp budget
 * `ppm` (QS_TUNE_DEPTH_CLAMP): per tree its own depth bits, the depth bits of the cheapest class the budget allows, and the
 * (tree, quartet) corrections that class costs. For tests and tools; qs_batch_upload applies the same functions. */
extern "C" int qs_depth_clamp_plan(uint32_t n_taxa, const qs_tree_batch *hb, uint32_t ppm, uint8_t *own_bits, uint8_t *class_bits,
                                   uint64_t *corrections) {
    if (!hb || !hb->leaf_off || (hb->n_trees && !hb->adj_depth) || n_taxa < 4) return QS_ERR_ARG;
    const uint64_t budget = (uint64_t)((double)binom4(n_taxa) * (double)ppm * 1e-6);
    for (uint32_t t = 0; t < hb->n_trees; ++t) {
        const uint32_t base = hb->leaf_off[t], L = hb->leaf_off[t + 1] - base;
        uint32_t depth = 0;
        for (uint32_t i = 0; i + 1 < L; ++i) depth = std::max<uint32_t>(depth, hb->adj_depth[base + i]);
        const uint32_t own = depth_class_of(depth);
        uint8_t soft, hard;
        clamp_choice(hb->adj_depth + base, L, own, budget, soft, hard);
        if (own_bits) own_bits[t] = (uint8_t)own;
        if (class_bits) class_bits[t] = soft;
        if (corrections) corrections[t] = soft < own ? clamp_cost(hb->adj_depth + base, L, (1u << soft) - 1u, nullptr) : 0;
    }
    return QS_OK;
}

/* Host-only: the classes qs_batch_upload forms for the trees of `hb` (which it does not validate) with the given floors and clamp
 * budget: per tree the kernel mode (0 binary_full, 1 general_full, 2 partial, 3 binary_partial) and depth bits of the class it is
 * counted in, and its slot in the class-ordered batch. */
extern "C" int qs_class_plan(uint32_t n_taxa, const qs_tree_batch *hb, uint32_t class_min_trees, uint32_t class_pct, uint32_t clamp_ppm,
                             uint8_t *mode_of_tree, uint8_t *bits_of_tree, uint32_t *slot_of_tree) {
    if (!hb || !hb->leaf_off || (hb->n_trees && !hb->adj_depth) || n_taxa < 4 || (class_pct & 0xFFu) > 100 || (class_pct & ~(0xFFu | QS_CLASS_PLAN_FUSED))) return QS_ERR_ARG;
    const uint32_t nt = hb->n_trees;
    const uint64_t budget = (uint64_t)((double)binom4(n_taxa) * (double)clamp_ppm * 1e-6);
    TreeFacts F;
    F.depth.resize(nt); F.mode.resize(nt); F.soft.resize(nt); F.hard.resize(nt);
    std::vector<uint32_t> stack;
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t base = hb->leaf_off[t], L = hb->leaf_off[t + 1] - base;
        uint32_t depth = 0;
        bool binary = false;
        tree_shape(hb->adj_depth + base, L, stack, depth, binary);
        F.depth[t] = (uint16_t)depth;
        F.mode[t] = (uint8_t)(L == n_taxa ? (binary ? MODE_BINARY_FULL : MODE_GENERAL_FULL) : (binary ? MODE_BINARY_PARTIAL : MODE_PARTIAL));
        clamp_choice(hb->adj_depth + base, L, depth_class_of(depth), budget, F.soft[t], F.hard[t]);
    }
    ClassPlan P;
    plan_classes(n_taxa, nt, hb->leaf_off, hb->adj_depth, F, class_min_trees, class_pct & 0xFFu, clamp_ppm, P, (class_pct & QS_CLASS_PLAN_FUSED) != 0);
    for (uint32_t t = 0; t < nt; ++t) {
        if (mode_of_tree) mode_of_tree[t] = P.mode[t];
        if (bits_of_tree) bits_of_tree[t] = P.bits[t];
    }
    if (slot_of_tree) {
        if (P.order.empty()) for (uint32_t t = 0; t < nt; ++t) slot_of_tree[t] = t;
        else for (uint32_t sl = 0; sl < nt; ++sl) slot_of_tree[P.order[sl]] = sl;
    }
    return QS_OK;
}

/* Host-only: bounds[0..n_shards] of the largest taxon id d that cut the table into n_shards contiguous shards [bounds[k], bounds[k+1])
 * (contiguous rank ranges: the rank's leading term is C(s3,4), quartet_lookup_table.hpp:161-165).
 *   by = QS_SHARDS_BY_TUPLES: balanced by the tuples a shard HOLDS, C(d,4) (memory: configs[4], out-of-core runs);
 *   by = QS_SHARDS_BY_COST:   balanced by what the count kernel SPENDS on a shard -- the wave tiles of its d-blocks (blocks of 8
 *        largest ids aligned to the shard's top, the partial block at its bottom), each tile priced at (kShardTileOverhead + live
 *        d slots): fitted to the per-shard timings of profiles/r06_scaling_model.json within 2 %. The largest shard is minimal
 *        over all contiguous cuts (bisection on the bound, greedy from the top). */
constexpr double kShardTileOverhead = 4.0;
extern "C" int qs_shard_bounds(uint32_t n_taxa, uint32_t n_shards, uint32_t by, uint32_t *bounds) {
    if (!bounds || n_shards == 0 || n_taxa < 4 || n_taxa > 65535) return QS_ERR_ARG;
    const uint32_t n = n_taxa, K = n_shards;
    if (by == QS_SHARDS_BY_TUPLES) {
        const uint64_t total = binom4(n);
        bounds[0] = 0;
        for (uint32_t r = 1; r < K; ++r) {
            const uint64_t target = total / K * r;
            uint32_t d = bounds[r - 1];
            while (d < n && binom4(d) < target) ++d;
            bounds[r] = d;
        }
        bounds[K] = n;
        return QS_OK;
    }
    if (by != QS_SHARDS_BY_COST) return QS_ERR_ARG;
    // T0[c] = sum of tiles(c') over c' < c, T1[c] = sum of tiles(c') * c'
    std::vector<double> T0(n + 2, 0.0), T1(n + 2, 0.0);
    for (uint32_t c = 0; c <= n; ++c) {
        const double t = c >= 2 ? (double)bitslice3_tiles_for_c(c) : 0.0;
        T0[c + 1] = T0[c] + t; T1[c + 1] = T1[c] + t * c;
    }
    auto cost = [&](uint32_t d_lo, uint32_t d_hi) {   // count-kernel cost of the shard [d_lo, d_hi)
        d_lo = std::max(d_lo, 3u);
        double tot = 0.0;
        for (uint32_t d1 = d_hi; d1 > d_lo;) {
            const uint32_t d0 = d1 > d_lo + kDB ? d1 - kDB : d_lo, p = d1 - d0;
            // third ids below the block see all p slots; c inside the block sees the slots above it: d1 - 1 - c
            tot += (kShardTileOverhead + p) * T0[d0] + (kShardTileOverhead + (double)(d1 - 1)) * (T0[d1 - 1] - T0[d0]) - (T1[d1 - 1] - T1[d0]);
            d1 = d0;
        }
        return tot;
    };
    auto cut = [&](double bound, uint32_t *out) {     // greedy from the top; true if K shards suffice (shards left over stay empty: [0, 0))
        uint32_t hi = n;
        if (out) out[K] = n;
        for (uint32_t k = K; k-- > 0;) {
            uint32_t lo = hi;
            while (lo > 0 && cost(lo - 1, hi) <= bound) --lo;   // (ids below 3 hold nothing: the cost stops growing, lo runs to 0)
            if (k == 0 && lo > 0) return false;
            if (out) out[k] = lo;
            hi = lo;
        }
        return true;
    };
    double lo_b = cost(0, n) / K, hi_b = cost(0, n);
    for (int it = 0; it < 60 && hi_b - lo_b > 1e-9 * hi_b; ++it) {
        const double mid = 0.5 * (lo_b + hi_b);
        if (cut(mid, nullptr)) hi_b = mid; else lo_b = mid;
    }
    if (!cut(hi_b, bounds)) return QS_ERR_STATE;
    // fewer useful shards than asked for (tiny n): the greedy cut leaves the EMPTY ones at the bottom as [0, 0); move them to the top as
    // [n, n) so that shard 0 (rank 0: the rank that writes the output) always holds quartets
    uint32_t e = 0;
    while (e < K && bounds[e + 1] == 0) ++e;
    if (e) {
        for (uint32_t k = 0; k + e <= K; ++k) bounds[k] = bounds[k + e];
        for (uint32_t k = K - e + 1; k <= K; ++k) bounds[k] = n;
    }
    return QS_OK;
}

extern "C" int qs_batch_upload(qs_ctx *c, const qs_tree_batch *hb, qs_device_batch **out) {
    if (!c || !hb || !out) return fail(c, QS_ERR_ARG, "qs_batch_upload: NULL argument");
    *out = nullptr;
    if (!hb->leaf_off || (hb->n_trees && (!hb->leaf_ids || !hb->adj_depth)))
        return fail(c, QS_ERR_ARG, "qs_batch_upload: leaf arrays missing");
    const uint32_t nt = hb->n_trees, n = c->n;
    // ---- validate (the reference dies on malformed input; we return a status) ----
    // Trees are independent: large batches are checked by a few host threads, the first error in tree order wins.
    struct Part { uint32_t max_depth = 0; bool all_full = true, all_binary = true; uint32_t err_tree = 0xFFFFFFFFu; std::string err; };
    std::vector<uint16_t> tree_depth(nt, 0); // deepest LCA of every tree
    std::vector<uint8_t> tree_mode(nt, 0);   // CountMode of every tree: the cheapest kernel instance that is exact for it
    std::vector<uint8_t> eff_soft(nt, 0), eff_hard(nt, 0);   // depth clamp: the lowest class (depth bits) the budget allows the tree (clamp_choice)
    const uint64_t clamp_budget = (uint64_t)((double)binom4(n) * (double)c->tune_clamp_ppm * 1e-6);
    auto check = [&](uint32_t t0, uint32_t t1, Part &P) {
        std::vector<uint32_t> stamp(n, 0xFFFFFFFFu);
        std::vector<uint32_t> stack;
        auto bad = [&](uint32_t t, const std::string &m) { P.err_tree = t; P.err = m; };
        for (uint32_t t = t0; t < t1; ++t) {
            if (hb->leaf_off[t + 1] < hb->leaf_off[t]) return bad(t, "qs_batch_upload: leaf_off not monotone");
            const uint32_t base = hb->leaf_off[t], L = hb->leaf_off[t + 1] - base;
            if (L > n) return bad(t, "qs_batch_upload: tree " + std::to_string(t) + " has more leaves than taxa");
            if (L != n) P.all_full = false;
            for (uint32_t i = 0; i < L; ++i) {
                const uint32_t id = hb->leaf_ids[base + i];
                if (id >= n) return bad(t, "qs_batch_upload: tree " + std::to_string(t) + ": taxon id out of range (unknown taxon)");
                if (stamp[id] == t) return bad(t, "qs_batch_upload: tree " + std::to_string(t) + ": duplicate taxon");
                stamp[id] = t;
            }
            // internal nodes from the adjacent-LCA depth sequence
            uint32_t depth = 0;
            bool binary = false;
            tree_shape(hb->adj_depth + base, L, stack, depth, binary);
            P.max_depth = std::max(P.max_depth, depth);
            tree_depth[t] = (uint16_t)depth;
            if (!binary) P.all_binary = false;
            tree_mode[t] = (uint8_t)(L == n ? (binary ? MODE_BINARY_FULL : MODE_GENERAL_FULL) : (binary ? MODE_BINARY_PARTIAL : MODE_PARTIAL));
            clamp_choice(hb->adj_depth + base, L, depth_class_of(tree_depth[t]), clamp_budget, eff_soft[t], eff_hard[t]);
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned workers = nt >= 2048 ? std::min(8u, std::min(hw, nt / 1024)) : 1u;
    std::vector<Part> parts(workers);
    if (workers == 1) check(0, nt, parts[0]);
    else {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < workers; ++w)
            pool.emplace_back([&, w] { check((uint32_t)((uint64_t)nt * w / workers), (uint32_t)((uint64_t)nt * (w + 1) / workers), parts[w]); });
        for (auto &th : pool) th.join();
    }
    uint32_t max_depth = 0;
    bool all_full = true, all_binary = true;
    for (const Part &P : parts) { // parts are in tree order: the first one with an error holds the earliest bad tree
        if (P.err_tree != 0xFFFFFFFFu) return fail(c, QS_ERR_ARG, P.err);
        max_depth = std::max(max_depth, P.max_depth);
        all_full = all_full && P.all_full; all_binary = all_binary && P.all_binary;
    }
    qs_device_batch *b = new qs_device_batch();
    DeviceBatch &d = b->d;
    d.n_trees = nt;
    d.total_leaves = nt ? hb->leaf_off[nt] : 0;
    d.max_depth = max_depth; d.all_full = all_full && nt > 0; d.all_binary = all_binary && nt > 0;
    // classes = (kernel mode, depth bits) per tree: plan_classes above
    std::vector<uint32_t> order_host;
    std::vector<FixUnit> fix_units;
    ClassPlan plan;
    {
        TreeFacts F;
        F.depth.swap(tree_depth); F.mode.swap(tree_mode); F.soft.swap(eff_soft); F.hard.swap(eff_hard);
        plan_classes(n, nt, hb->leaf_off, hb->adj_depth, F, c->tune_class_min, c->tune_class_pct, c->tune_clamp_ppm, plan,
                     c->tune_fuse && c->tune_gather_impl != QS_IMPL_SWAR);
        tree_depth.swap(F.depth); tree_mode.swap(F.mode);
        d.n_classes = plan.n_classes;
        for (uint32_t k = 0; k < plan.n_classes; ++k) { d.class_mode[k] = plan.class_mode[k]; d.class_bits[k] = plan.class_bits[k]; d.class_end[k] = plan.class_end[k]; d.class_max_depth[k] = plan.class_max_depth[k]; }
        order_host = plan.order;
        constexpr uint32_t top_bits = 10;
        // correction units of the trees whose class holds fewer depth bits than they need, by slot (a slice of a class = a range of
        // slots = a range of units). One unit = one workgroup = a stretch of a run's triples worth ~32 K fourth leaves.
        if (c->tune_clamp_ppm) {
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            auto emit = [&](uint32_t slot, uint32_t t) {
                const uint32_t fb = plan.bits[t];
                if (fb > top_bits || depth_class_of(tree_depth[t]) <= fb) return;
                const uint32_t base = hb->leaf_off[t], L = hb->leaf_off[t + 1] - base;
                runs.clear();
                const uint64_t cost = clamp_cost(hb->adj_depth + base, L, (1u << fb) - 1u, &runs);
                if (cost == ~0ull) return;   // cannot happen: the cut of a lower class was accepted, and runs only shrink with the cut
                d.fix_quartets += cost; d.clamped_trees++;
                const uint32_t per_unit = std::max<uint32_t>(1u, 32768u / std::max<uint32_t>(L, 1u));
                for (const auto &r : runs) {
                    const uint32_t triples = (uint32_t)binom3(r.second);
                    for (uint32_t t0 = 0; t0 < triples; t0 += per_unit) {
                        fix_units.push_back(FixUnit{t, r.first | (r.second << 16), t0, std::min(triples, t0 + per_unit)});
                        b->fix_slot.push_back(slot);
                    }
                }
            };
            if (order_host.empty()) for (uint32_t t = 0; t < nt; ++t) emit(t, t);
            else for (uint32_t sl = 0; sl < nt; ++sl) emit(sl, order_host[sl]);
            d.n_fix = (uint32_t)fix_units.size();
        }
    }
    // scatter batches: the tree of every inner node, and a bounds check of the leaf ranges (host side, before any copy)
    const bool with_nodes = hb->node_off && hb->rng_off && hb->ranges;
    std::vector<uint32_t> node_tree;
    if (with_nodes) {
        d.n_nodes = hb->node_off[nt];
        d.n_links = hb->rng_off[d.n_nodes];
        node_tree.resize(d.n_nodes);
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t L = hb->leaf_off[t + 1] - hb->leaf_off[t];
            for (uint32_t v = hb->node_off[t]; v < hb->node_off[t + 1]; ++v) {
                node_tree[v] = t;
                for (uint32_t k = hb->rng_off[v]; k < hb->rng_off[v + 1]; ++k)
                    if (hb->ranges[2 * k] >= std::max(L, 1u) || hb->ranges[2 * k + 1] >= std::max(L, 1u)) {
                        qs_batch_free(c, b);
                        return fail(c, QS_ERR_ARG, "qs_batch_upload: range position out of bounds");
                    }
            }
        }
    }
    hipError_t e = hipSetDevice(c->device);
    Stager st;
    if (e == hipSuccess) {
        size_t need = Stager::padded(((size_t)nt + 1) * 4) + 2 * Stager::padded((size_t)d.total_leaves * 2) + Stager::padded(order_host.size() * 4) +
                      Stager::padded(fix_units.size() * sizeof(FixUnit));
        if (with_nodes) need += Stager::padded(((size_t)nt + 1) * 4) + Stager::padded(((size_t)d.n_nodes + 1) * 4) + Stager::padded((size_t)d.n_nodes * 4) + Stager::padded((size_t)2 * d.n_links * 2);
        e = st.begin(c, need);
    }
    if (e == hipSuccess) {
        d.leaf_off = st.put(hb->leaf_off, (size_t)nt + 1);
        d.leaf_ids = st.put(hb->leaf_ids, d.total_leaves);
        d.adj_depth = st.put(hb->adj_depth, d.total_leaves);
        if (!order_host.empty()) d.tree_order = st.put(order_host.data(), order_host.size());
        if (!fix_units.empty()) d.fix_units = st.put(fix_units.data(), fix_units.size());
        if (with_nodes) {
            d.node_off = st.put(hb->node_off, (size_t)nt + 1);
            d.rng_off = st.put(hb->rng_off, (size_t)d.n_nodes + 1);
            d.node_tree = st.put(node_tree.data(), d.n_nodes);
            d.ranges = st.put(hb->ranges, (size_t)2 * d.n_links);
        }
        e = st.finish(&d.ready);
    }
    d.slab = st.slab;
    if (e != hipSuccess) {
        qs_batch_free(c, b);
        return fail(c, e == hipErrorOutOfMemory ? QS_ERR_OOM : QS_ERR_HIP, std::string("qs_batch_upload: ") + hipGetErrorString(e));
    }
    *out = b;
    return QS_OK;
}

extern "C" uint32_t qs_batch_flags(const qs_device_batch *b) {
    if (!b) return 0;
    return (b->d.all_full ? QS_BATCH_ALL_TAXA : 0u) | (b->d.all_binary ? QS_BATCH_BINARY : 0u);
}

// corrections of the depth clamp for the trees in slots [slot_lo, slot_hi) of the batch, after their count kernel
static hipError_t clamp_fix_slots(qs_ctx *c, const qs_device_batch *b, uint32_t slot_lo, uint32_t slot_hi, int mode, uint32_t *wire) {
    const auto lo = std::lower_bound(b->fix_slot.begin(), b->fix_slot.end(), slot_lo), hi = std::lower_bound(lo, b->fix_slot.end(), slot_hi);
    if (lo == hi) return hipSuccess;
    return launch_clamp_fix(c->stream, b->d, b->d.fix_units + (lo - b->fix_slot.begin()), (uint32_t)(hi - lo), std::max(c->d_lo, 3u), c->d_hi,
                            c->rank_lo, c->table, (int)c->count_bits, mode, wire);
}

// QS_COUNT_WIRE16X2: count a binary_full batch straight into the attached wire buffer (one word per tuple)
static int count_batch_wire(qs_ctx *c, const qs_device_batch *b, uint32_t algo) {
    const DeviceBatch &d = b->d;
    if (!c->wire_out) return fail(c, QS_ERR_STATE, "QS_COUNT_WIRE16X2: no wire buffer (qs_wire_attach first)");
    const bool overwrite = (algo & QS_COUNT_OVERWRITE) != 0;
    const uint32_t base_algo = algo & 0xFFu;
    if (base_algo != QS_ALGO_AUTO && base_algo != QS_ALGO_GATHER) return fail(c, QS_ERR_ARG, "QS_COUNT_WIRE16X2 needs the gather algorithm");
    const bool timed = (algo & QS_COUNT_TIMED) != 0;
    QS_HIP(c, hipSetDevice(c->device));
    if (d.n_trees == 0) {
        if (overwrite) { QS_HIP(c, hipMemsetAsync(c->wire_out, 0, c->n_tuples * 4, c->stream)); c->wire_trees = 0; }
        return QS_OK;
    }
    if (!(d.all_full && d.all_binary))
        return fail(c, QS_ERR_STATE, "QS_COUNT_WIRE16X2: the batch is not made of binary trees that hold all taxa (count into the table and use qs_table_pack16)");
    if (d.ready) QS_HIP(c, hipStreamWaitEvent(c->stream, d.ready, 0));
    if ((overwrite ? 0 : c->wire_trees) + d.n_trees > 0xFFFFull)
        return fail(c, QS_ERR_OVERFLOW, "QS_COUNT_WIRE16X2: more than 65535 trees do not fit 16-bit cells");
    if (d.class_bits[d.n_classes - 1] > 10) return fail(c, QS_ERR_UNSUPPORTED, "QS_COUNT_WIRE16X2: tree depth needs more than 10 bits; count into the table instead");
    CountGeometry g;
    g.n = c->n; g.d_lo = std::max(c->d_lo, 3u); g.d_hi = c->d_hi; g.rank_lo = c->rank_lo; g.n_dblk = c->n_dblk;
    g.total_tiles = c->total_tiles3; g.dprefix = c->dprefix3; g.cprefix = c->cprefix3;
    { int rc_o = tile_order(c, 0, &g.perm); if (rc_o != QS_OK) return rc_o; }
    g.perm_coop = c->perm_coop; g.n_coop = c->n_coop; g.perm_rest = c->perm_rest; g.n_rest = c->n_rest;
    c->ev_used = 0;
    if (timed) QS_HIP(c, mark(c, 0));
    bool first = true;
    c->variant = "gather/binary_full/bitslice_";
    for (uint32_t k = 0; k < d.n_classes; ++k) {
        const uint32_t s_lo = k ? d.class_end[k - 1] : 0, s_hi = d.class_end[k], depth_bits = d.class_bits[k];
        const uint32_t compact_nw = std::max(depth_bits, 4u);
        const size_t chunk_bytes = (size_t)binom2(c->n) * compact_nw * 4;
        const uint32_t n_chunks_total = (s_hi - s_lo + 31) / 32;
        const uint32_t chunks_per_slice = slice_groups(c, chunk_bytes, n_chunks_total, g.perm != nullptr);
        const size_t need = (size_t)chunks_per_slice * chunk_bytes;
        if (need > c->panel_bytes) {
            if (c->panel) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->panel); c->panel = nullptr; c->panel_bytes = 0; }
            if (hipMalloc(&c->panel, need) != hipSuccess) return fail(c, QS_ERR_OOM, "Insufficient memory! (pair-depth panel)");
            c->panel_bytes = need;
        }
        for (uint32_t ch0 = 0; ch0 < n_chunks_total; ch0 += chunks_per_slice) {
            const uint32_t nch = std::min(chunks_per_slice, n_chunks_total - ch0);
            const uint32_t t0 = ch0 * 32, nt = std::min(nch * 32, s_hi - s_lo - t0);
            DeviceBatch sub = d;
            sub.slot0 = s_lo + t0;
            sub.n_trees = nt;
            QS_HIP(c, launch_build_bitpanel(c->stream, sub, c->n, false, c->panel, nch, compact_nw, c->tune_panel_kernel == 1));
            if (timed) QS_HIP(c, mark(c, 0));
            QS_HIP(c, launch_count_bitslice3(c->stream, g, c->panel, (int)depth_bits, MODE_BINARY_FULL, nch, nt, nullptr, 32, c->dev_flags,
                                             overwrite && first, c->wire_out));
            if (timed) QS_HIP(c, mark(c, 1));
            if (d.n_fix) { QS_HIP(c, clamp_fix_slots(c, b, sub.slot0, sub.slot0 + nt, MODE_BINARY_FULL, c->wire_out)); if (timed) QS_HIP(c, mark(c, 2)); }
            first = false;
        }
        c->variant += (k ? "+b" : "b") + std::to_string(depth_bits) + "x2" + (d.n_classes > 1 ? ":" + std::to_string(s_hi - s_lo) : "");
    }
    c->variant += "/wire_u16x2";
    if (d.n_fix) c->variant += "/clamp:" + std::to_string(d.clamped_trees);
    if (c->n_coop) c->variant += "/coop4";
    c->wire_trees = (overwrite ? 0 : c->wire_trees) + d.n_trees;
    c->last_timed = timed;
    return QS_OK;
}

extern "C" int qs_count_batch(qs_ctx *c, const qs_device_batch *b, uint32_t algo) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c || !b) return fail(c, QS_ERR_ARG, "qs_count_batch: NULL argument");
    if (algo & QS_COUNT_WIRE16X2) return count_batch_wire(c, b, algo);
    if (!c->table) return fail(c, QS_ERR_STATE, "qs_count_batch: no table (qs_table_alloc / qs_table_attach first)");
    const DeviceBatch &d = b->d;
    QS_HIP(c, hipSetDevice(c->device));
    if (d.n_trees == 0) {
        if (algo & QS_COUNT_OVERWRITE) { // "discard the previous contents" holds for an empty batch too
            QS_HIP(c, hipMemsetAsync(c->table, 0, c->n_tuples * 3 * (c->count_bits / 8), c->stream));
            c->trees_counted = 0;
        }
        c->last_timed = false;
        return QS_OK;
    }
    if (c->count_bits == 16 && ((algo & QS_COUNT_OVERWRITE) ? 0 : c->trees_counted) + d.n_trees > 0xFFFFull)
        return fail(c, QS_ERR_OVERFLOW, "qs_count_batch: more than 65535 trees need count_bits = 32");
    if (d.ready) QS_HIP(c, hipStreamWaitEvent(c->stream, d.ready, 0));   // the batch's copies run on the copy stream
    const bool overwrite = (algo & QS_COUNT_OVERWRITE) != 0;
    const bool timed = (algo & QS_COUNT_TIMED) != 0;
    algo &= ~(QS_COUNT_OVERWRITE | QS_COUNT_TIMED);
    if (algo == QS_ALGO_AUTO) algo = QS_ALGO_GATHER;
    if (overwrite && algo != QS_ALGO_GATHER) return fail(c, QS_ERR_ARG, "qs_count_batch: QS_COUNT_OVERWRITE needs the gather algorithm");
    // (an overwrite resets trees_counted only once nothing can fail any more: a refused call leaves table and count as they were)
    c->ev_used = 0;
    if (timed) QS_HIP(c, mark(c, 0));
    if (algo == QS_ALGO_GATHER) {
        static const char *mode_names[4] = {"binary_full", "general_full", "partial", "binary_partial"};
        CountGeometry g;
        g.n = c->n; g.d_lo = std::max(c->d_lo, 3u); g.d_hi = c->d_hi; g.rank_lo = c->rank_lo;
        g.n_dblk = c->n_dblk; g.total_tiles = c->total_tiles; g.dprefix = c->dprefix; g.cprefix = c->cprefix;
        // One (panel build + count kernel) per slice of every class of the batch (classes = kernel mode x depth bits per TREE:
        // qs_batch_upload; slices: slice_groups). With QS_IMPL_SWAR the whole batch is one class of the byte-SWAR kernel,
        // in the mode the batch as a whole needs.
        const uint32_t top_bits = 10u;
        const bool all_swar = c->tune_gather_impl == QS_IMPL_SWAR;
        if (c->tune_gather_impl == QS_IMPL_BITSLICE)
            for (uint32_t k = 0; k < d.n_classes; ++k)
                if (d.class_bits[k] > top_bits) return fail(c, QS_ERR_UNSUPPORTED, "QS_IMPL_BITSLICE: tree depth needs more than 10 bits");
        const uint32_t n_cls = all_swar ? 1u : d.n_classes;
        bool first = true, mixed = false;
        for (uint32_t k = 1; k < n_cls; ++k) mixed = mixed || d.class_mode[k] != d.class_mode[0];
        const int batch_mode = !d.all_full ? MODE_PARTIAL : (d.all_binary ? MODE_BINARY_FULL : MODE_GENERAL_FULL);
        std::string names;
        bool any_coop = false;
        // ---- fused launches (QS_TUNE_FUSE_CLASSES): the bit-sliced classes that share their depth bits run as segments of ONE launch
        // of count_bitslice3_fused_kernel -- one pass over the table, one wave prologue / epilogue, whatever the mix of modes ----
        std::vector<bool> done(n_cls, false);
        uint32_t fused_launch_groups = 0;
        if (c->tune_fuse && !all_swar) {
            const uint32_t *order = nullptr;
            for (uint32_t bb = 4; bb <= (uint32_t)kFusedMaxBits; ++bb) {
                std::vector<uint32_t> ks;
                for (uint32_t k = 0; k < n_cls; ++k) if (std::max(d.class_bits[k], 4u) == bb) ks.push_back(k);
                // a lone binary_partial class at 4 or 5 bits takes the fused binary kernel too (with an empty binary_full segment): under the one
                // dispatch it needs 123 VGPRs and spills nothing at 4 waves per SIMD, where the one-class instance is held to 128 with 6-24
                // spilled: 512 taxa x 1500 trees with 10 % of the taxa dropped 64.95 -> 63.7 ms (profiles/r06_experiments.md 3)
                const bool lone_bp4 = ks.size() == 1 && bb <= 5 && d.class_mode[ks[0]] == MODE_BINARY_PARTIAL;   // (5 bits: 4 waves instead of the one-class instance's 3)
                if (ks.size() < 2 && !lone_bp4) continue;
                if (!order) { int rc_o = tile_order(c, 0, &order); if (rc_o != QS_OK) return rc_o; }
                struct Seg { uint32_t k, s_lo, trees, groups, mode, nw; bool part; size_t chunk_bytes; };
                std::vector<Seg> segs;
                uint32_t g_total = 0;
                size_t max_chunk = 0;
                for (uint32_t k : ks) {
                    Seg sg;
                    sg.k = k; sg.s_lo = k ? d.class_end[k - 1] : 0u; sg.trees = d.class_end[k] - sg.s_lo; sg.groups = (sg.trees + 31) / 32;
                    sg.mode = d.class_mode[k]; sg.part = sg.mode == MODE_PARTIAL || sg.mode == MODE_BINARY_PARTIAL;
                    sg.nw = bb + (sg.part ? 1u : 0u); sg.chunk_bytes = (size_t)binom2(c->n) * sg.nw * 4;
                    g_total += sg.groups; max_chunk = std::max(max_chunk, sg.chunk_bytes);
                    segs.push_back(sg);
                }
                const uint32_t per_slice = slice_groups(c, max_chunk, g_total, order != nullptr);
                const size_t need = (size_t)per_slice * max_chunk + 256 * segs.size();
                if (need > c->panel_bytes) {
                    if (c->panel) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->panel); c->panel = nullptr; c->panel_bytes = 0; }
                    if (hipMalloc(&c->panel, need) != hipSuccess) return fail(c, QS_ERR_OOM, "Insufficient memory! (pair-depth panel)");
                    c->panel_bytes = need;
                }
                CountGeometry g3 = g;
                g3.total_tiles = c->total_tiles3; g3.dprefix = c->dprefix3; g3.cprefix = c->cprefix3; g3.perm = order;
                // slices of the concatenated group list [0, g_total): slice [g0, g1) takes from every class the groups that fall into it
                for (uint32_t g0 = 0; g0 < g_total; g0 += per_slice) {
                    const uint32_t g1 = std::min(g_total, g0 + per_slice);
                    const void *seg_panel[4] = {nullptr, nullptr, nullptr, nullptr};
                    uint32_t seg_groups[4] = {0, 0, 0, 0}, seg_trees[4] = {0, 0, 0, 0}, seg_slot[4] = {0, 0, 0, 0};
                    size_t off = 0;
                    uint32_t base = 0;
                    for (const Seg &sg : segs) {
                        const uint32_t a = std::max(g0, base), e_ = std::min(g1, base + sg.groups);
                        base += sg.groups;
                        if (a >= e_) continue;
                        const uint32_t ga = a - (base - sg.groups), nch = e_ - a;
                        const uint32_t t0 = ga * 32, nt = std::min(nch * 32, sg.trees - t0);
                        DeviceBatch sub = d;
                        sub.slot0 = sg.s_lo + t0;
                        sub.n_trees = nt;
                        void *pp = (char *)c->panel + off;
                        QS_HIP(c, launch_build_bitpanel(c->stream, sub, c->n, sg.part, pp, nch, sg.nw, c->tune_panel_kernel == 1));
                        seg_panel[sg.mode] = pp; seg_groups[sg.mode] = nch; seg_trees[sg.mode] = nt; seg_slot[sg.mode] = sub.slot0;
                        off += ((size_t)nch * sg.chunk_bytes + 255) & ~(size_t)255;
                    }
                    if (timed) QS_HIP(c, mark(c, 0));
                    QS_HIP(c, launch_count_bitslice3_fused(c->stream, g3, seg_panel, seg_groups, seg_trees, (int)bb, c->table, (int)c->count_bits, c->dev_flags, overwrite && first));
                    first = false;
                    if (timed) QS_HIP(c, mark(c, 1));
                    if (d.n_fix) {
                        bool any_fix = false;
                        for (int mo = 0; mo < 4; ++mo)
                            if (seg_groups[mo]) {   // (the rule of the correction depends on the mode: a tied quartet sits in the third cell of a binary tree)
                                QS_HIP(c, clamp_fix_slots(c, b, seg_slot[mo], seg_slot[mo] + seg_trees[mo], mo, nullptr));
                                any_fix = true;
                            }
                        if (timed && any_fix) QS_HIP(c, mark(c, 2));
                    }
                }
                for (const Seg &sg : segs) {
                    done[sg.k] = true;
                    std::string nm = "bitslice_b" + std::to_string(bb) + "x2";          // (named as the class-by-class path names its classes)
                    if (mixed) nm = std::string(mode_names[sg.mode]) + "." + nm;
                    if (n_cls > 1) nm += ":" + std::to_string(sg.trees);
                    names += (names.empty() ? "" : "+") + nm;
                }
                ++fused_launch_groups;
            }
        }
        for (uint32_t k = 0; k < n_cls; ++k) {
            if (done[k]) continue;
            const uint32_t s_lo = (all_swar || k == 0) ? 0u : d.class_end[k - 1], s_hi = all_swar ? d.n_trees : d.class_end[k];
            const uint32_t depth_bits = all_swar ? 11u : d.class_bits[k];   // 11 = beyond the bit-sliced instances
            const uint32_t cls_max_depth = all_swar ? d.max_depth : d.class_max_depth[k];
            const bool use_bitslice = depth_bits <= top_bits;
            // the byte-SWAR kernel has no binary_partial instance: such trees are exact under its partial mode
            int mode = all_swar ? batch_mode : (int)d.class_mode[k];
            if (!use_bitslice && mode == MODE_BINARY_PARTIAL) mode = MODE_PARTIAL;
            const bool part = mode == MODE_PARTIAL || mode == MODE_BINARY_PARTIAL;           // panel elements carry a presence word
            const bool bin_tiles = true;   // every instance of the bit-sliced kernel owns two a-columns per lane (16 a x 8 b tiles) since round 4
            // bit-sliced classes run count_bitslice3_kernel on the compact panel
            int bits = 8;
            uint32_t tpc;            // trees per panel element
            size_t elem_bytes;       // bytes per (pair, element)
            const uint32_t compact_nw = std::max(depth_bits, 4u) + (part ? 1u : 0u); // words per compact panel element
            if (use_bitslice) { tpc = 32; elem_bytes = compact_nw * 4; }
            else {
                const uint32_t lim8 = part ? kMaxDepthU8Partial : kMaxDepthU8Full;
                const uint32_t lim16 = part ? kMaxDepthU16Partial : kMaxDepthU16Full;
                if (cls_max_depth <= lim8) bits = 8;
                else if (cls_max_depth <= lim16) bits = 16;
                else return fail(c, QS_ERR_UNSUPPORTED, "qs_count_batch: tree depth " + std::to_string(cls_max_depth) + " exceeds the panel range; re-root the tree at its centre");
                tpc = 16 / (bits / 8); elem_bytes = 16;
            }
            const size_t chunk_bytes = (size_t)binom2(c->n) * elem_bytes;
            const uint32_t n_chunks_total = (s_hi - s_lo + tpc - 1) / tpc;
            const uint32_t *order = nullptr;   // launch order of the bit-sliced kernel's tiles for this class's tiling
            if (use_bitslice) { int rc_o = tile_order(c, 0, &order); if (rc_o != QS_OK) return rc_o; }
            const uint32_t chunks_per_slice = slice_groups(c, chunk_bytes, n_chunks_total, order != nullptr);
            const size_t need = (size_t)chunks_per_slice * chunk_bytes;
            if (need > c->panel_bytes) {
                if (c->panel) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->panel); c->panel = nullptr; c->panel_bytes = 0; }
                hipError_t e = hipMalloc(&c->panel, need);
                if (e != hipSuccess) return fail(c, QS_ERR_OOM, "Insufficient memory! (pair-depth panel)");
                c->panel_bytes = need;
            }
            for (uint32_t ch0 = 0; ch0 < n_chunks_total; ch0 += chunks_per_slice) {
                const uint32_t nch = std::min(chunks_per_slice, n_chunks_total - ch0);
                const uint32_t t0 = ch0 * tpc, nt = std::min(nch * tpc, s_hi - s_lo - t0);
                DeviceBatch sub = d;
                sub.slot0 = s_lo + t0;        // slots [slot0, slot0 + nt) of the class-ordered batch
                sub.n_trees = nt;
                if (use_bitslice) QS_HIP(c, launch_build_bitpanel(c->stream, sub, c->n, part, c->panel, nch, compact_nw, c->tune_panel_kernel == 1));
                else QS_HIP(c, launch_build_panel(c->stream, sub, c->n, bits, part, c->panel, nch));
                if (timed) QS_HIP(c, mark(c, 0));
                if (use_bitslice) {
                    CountGeometry g3 = g;
                    if (bin_tiles) {
                        g3.total_tiles = c->total_tiles3; g3.dprefix = c->dprefix3; g3.cprefix = c->cprefix3;
                        if (mode == MODE_BINARY_FULL) { g3.perm_coop = c->perm_coop; g3.n_coop = c->n_coop; g3.perm_rest = c->perm_rest; g3.n_rest = c->n_rest; any_coop = any_coop || (c->n_coop && depth_bits <= 7); }   // (count_bitslice4_kernel carries at most 7 planes)
                    }
                    else { g3.total_tiles = c->total_tiles1t; g3.dprefix = c->dprefix1t; g3.cprefix = c->cprefix; }
                    g3.perm = order;
                    QS_HIP(c, launch_count_bitslice3(c->stream, g3, c->panel, (int)depth_bits, mode, nch, nt, c->table, (int)c->count_bits, c->dev_flags, overwrite && first, nullptr));
                }
                else QS_HIP(c, launch_count_gather(c->stream, g, c->panel, bits, mode, nch, nt, c->table, (int)c->count_bits, c->dev_flags, overwrite && first));
                first = false;
                if (timed) QS_HIP(c, mark(c, 1));
                if (use_bitslice && d.n_fix) {   // (never with QS_IMPL_SWAR: the byte panel holds the trees' own depths)
                    QS_HIP(c, clamp_fix_slots(c, b, sub.slot0, sub.slot0 + nt, mode, nullptr));
                    if (timed) QS_HIP(c, mark(c, 2));
                }
            }
            // kernel variant of the class; several classes: "a:trees+b:trees", classes of several modes: "mode.a:trees+..."
            std::string nm = use_bitslice ? "bitslice_b" + std::to_string(std::max(depth_bits, 4u)) + (bin_tiles ? "x2" : "")
                                          : "depth_u" + std::to_string(bits);
            if (mixed) nm = std::string(mode_names[mode]) + "." + nm;
            if (n_cls > 1) nm += ":" + std::to_string(s_hi - s_lo);
            names += (names.empty() ? "" : "+") + nm;
        }
        const int one_mode = all_swar ? batch_mode : (int)d.class_mode[0];
        c->variant = std::string("gather/") + (mixed ? "mixed" : mode_names[(!all_swar && d.class_bits[0] > top_bits && one_mode == MODE_BINARY_PARTIAL) ? (int)MODE_PARTIAL : one_mode]) + "/" + names + "/count_u" + std::to_string(c->count_bits);
        if (fused_launch_groups) c->variant += "/fused:" + std::to_string(fused_launch_groups);   // depth-bits groups whose classes shared a launch (count_bitslice3_fused_kernel)
        if (any_coop) c->variant += "/coop4";   // tiles with two a-blocks of binary_full classes: count_bitslice4_kernel
        if (!all_swar && d.n_fix) c->variant += "/clamp:" + std::to_string(d.clamped_trees);   // trees counted below their own depth bits + clamp_fix_kernel
    } else if (algo == QS_ALGO_SCATTER) {
        if (!d.node_off) return fail(c, QS_ERR_ARG, "qs_count_batch: QS_ALGO_SCATTER needs node_off/rng_off/ranges in the batch");
        if (c->n > 4096) return fail(c, QS_ERR_UNSUPPORTED, "scatter: n too large");
        QS_HIP(c, launch_count_scatter(c->stream, d, c->n, c->d_lo, c->d_hi, c->rank_lo, c->table, (int)c->count_bits));
        if (timed) QS_HIP(c, mark(c, 1));
        c->variant = std::string("scatter/atomic/count_u") + std::to_string(c->count_bits);
    } else {
        return fail(c, QS_ERR_ARG, "qs_count_batch: unknown algo");
    }
    c->last_timed = timed;
    c->trees_counted = (overwrite ? 0 : c->trees_counted) + d.n_trees;
    return QS_OK;
}

extern "C" int qs_sync(qs_ctx *c) {
    if (!c) return QS_ERR_ARG;
    QS_HIP(c, hipSetDevice(c->device));
    QS_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t fl[4] = {0, 0, 0, 0};
    QS_HIP(c, hipMemcpy(fl, c->dev_flags, 16, hipMemcpyDeviceToHost));
    if (fl[0]) {
        (void)hipMemset(c->dev_flags, 0, 4);
        return fail(c, QS_ERR_OVERFLOW, "count table overflow: a counter exceeded count_bits");
    }
    if (fl[3]) {
        (void)hipMemset(c->dev_flags + 3, 0, 4);
        return fail(c, QS_ERR_STATE, "two-cell wire format: a tuple does not sum to the number of trees (the batch was not binary with all taxa)");
    }
    return QS_OK;
}

extern "C" int qs_count_trees(qs_ctx *c, const qs_tree_batch *batch, uint32_t algo) {
    qs_device_batch *b = nullptr;
    int rc = qs_batch_upload(c, batch, &b);
    if (rc != QS_OK) return rc;
    rc = qs_count_batch(c, b, algo);
    int rc2 = qs_sync(c);
    qs_batch_free(c, b);
    return rc != QS_OK ? rc : rc2;
}

extern "C" int qs_last_count_ms(qs_ctx *c, float out_ms[3]) {
    if (!c || !c->last_timed || c->ev_used < 2) return fail(c, QS_ERR_STATE, "qs_last_count_ms: the last qs_count_batch did not carry QS_COUNT_TIMED");
    QS_HIP(c, hipEventSynchronize(c->evs[c->ev_used - 1]));
    out_ms[0] = out_ms[1] = 0.f;
    for (uint32_t i = 1; i < c->ev_used; ++i) {
        float ms = 0.f;
        QS_HIP(c, hipEventElapsedTime(&ms, c->evs[i - 1], c->evs[i]));
        out_ms[c->ev_kind[i] ? 1 : 0] += ms;   // (the depth-clamp corrections count as count-kernel time: qs_last_count_fix_ms has their share)
    }
    QS_HIP(c, hipEventElapsedTime(&out_ms[2], c->evs[0], c->evs[c->ev_used - 1]));
    return QS_OK;
}

extern "C" int qs_last_count_launches(const qs_ctx *c) {
    if (!c || !c->last_timed) return 0;
    int k = 0;
    for (uint32_t i = 1; i < c->ev_used; ++i) k += c->ev_kind[i] == 1 ? 1 : 0;
    return k;
}

extern "C" float qs_last_count_fix_ms(qs_ctx *c) {
    if (!c || !c->last_timed || c->ev_used < 2) return 0.f;
    if (hipEventSynchronize(c->evs[c->ev_used - 1]) != hipSuccess) return 0.f;
    float sum = 0.f;
    for (uint32_t i = 1; i < c->ev_used; ++i) {
        float ms = 0.f;
        if (c->ev_kind[i] == 2 && hipEventElapsedTime(&ms, c->evs[i - 1], c->evs[i]) == hipSuccess) sum += ms;
    }
    return sum;
}

extern "C" int qs_last_count_events(qs_ctx *c, float *ms, uint8_t *kind, int cap) {
    if (!c || !c->last_timed || c->ev_used < 2) return 0;
    if (hipEventSynchronize(c->evs[c->ev_used - 1]) != hipSuccess) return 0;
    int k = 0;
    for (uint32_t i = 1; i < c->ev_used && k < cap; ++i, ++k) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, c->evs[i - 1], c->evs[i]);
        if (ms) ms[k] = t;
        if (kind) kind[k] = c->ev_kind[i];
    }
    return k;
}

extern "C" int qs_batch_clamp_info(const qs_device_batch *b, uint64_t out[3]) {
    if (!b || !out) return QS_ERR_ARG;
    out[0] = b->d.clamped_trees; out[1] = b->d.fix_quartets; out[2] = b->d.n_fix;
    return QS_OK;
}

extern "C" const char *qs_last_count_variant(const qs_ctx *c) { return c ? c->variant.c_str() : ""; }

extern "C" int qs_lookup(qs_ctx *c, uint64_t nq, const uint16_t *abcd, uint64_t *out3) {
    if (!c || !c->table) return fail(c, QS_ERR_STATE, "qs_lookup: no table");
    if (nq == 0) return QS_OK;
    if (!abcd || !out3) return fail(c, QS_ERR_ARG, "qs_lookup: NULL");
    for (uint64_t i = 0; i < nq * 4; ++i)
        if (abcd[i] >= c->n) return fail(c, QS_ERR_ARG, "qs_lookup: id out of range");
    QS_HIP(c, hipSetDevice(c->device));
    uint16_t *dq = nullptr; uint64_t *dout = nullptr;
    QS_HIP(c, hipMalloc(&dq, nq * 8));
    hipError_t e = hipMalloc(&dout, nq * 24);
    if (e != hipSuccess) { (void)hipFree(dq); return fail(c, QS_ERR_OOM, "qs_lookup: hipMalloc"); }
    e = hipMemcpyAsync(dq, abcd, nq * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_lookup(c->stream, c->n, c->d_lo, c->d_hi, c->rank_lo, c->table, (int)c->count_bits, nq, dq, dout);
    if (e == hipSuccess) e = hipMemcpyAsync(out3, dout, nq * 24, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dq); (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, QS_ERR_HIP, std::string("qs_lookup: ") + hipGetErrorString(e));
    return QS_OK;
}

// ---- scoring -----------------------------------------------------------------------------

// QuartetScoreComputer.hpp:135-159, evaluated with the host's libm like the reference.
static double host_log_score(uint64_t q1, uint64_t q2, uint64_t q3) {
    if (q1 == 0 && q2 == 0 && q3 == 0) return 0;
    uint64_t sum = q1 + q2 + q3;
    double p1 = (double)q1 / sum, p2 = (double)q2 / sum, p3 = (double)q3 / sum;
    double qic = 1;
    if (p1 != 0) qic += p1 * std::log(p1) / std::log(3);
    if (p2 != 0) qic += p2 * std::log(p2) / std::log(3);
    if (p3 != 0) qic += p3 * std::log(p3) / std::log(3);
    return (q1 < q2 || q1 < q3) ? qic * -1 : qic;
}


static int build_ref(qs_ctx *c, const qs_ref_tree *ref, RefHost &R) {
    if (!ref || !ref->parent || !ref->leaf_node) return fail(c, QS_ERR_ARG, "reference tree: NULL arrays");
    if (c && ref->n_taxa != c->n) return fail(c, QS_ERR_ARG, "reference tree: n_taxa differs from the context");
    const uint32_t N = ref->n_nodes, n = ref->n_taxa;
    if (N < n + 1 || N > 65535) return fail(c, QS_ERR_ARG, "reference tree: bad node count");
    R.n_nodes = N; R.n = n;
    R.parent.assign(ref->parent, ref->parent + N);
    R.nchild.assign(N, 0); R.depth.assign(N, 0);
    int root = -1;
    for (uint32_t v = 0; v < N; ++v) {
        int p = R.parent[v];
        if (p < 0) { if (root >= 0) return fail(c, QS_ERR_ARG, "reference tree: several roots"); root = (int)v; }
        else if ((uint32_t)p >= N || (uint32_t)p == v) return fail(c, QS_ERR_ARG, "reference tree: bad parent");
        else R.nchild[p]++;
    }
    if (root < 0) return fail(c, QS_ERR_ARG, "reference tree: no root");
    R.root = (uint32_t)root;
    // depths (parents may come after children in the numbering: iterate with memo)
    std::vector<int> dep(N, -1);
    dep[root] = 0;
    std::vector<uint32_t> chain;
    for (uint32_t v = 0; v < N; ++v) {
        chain.clear();
        uint32_t x = v;
        while (dep[x] < 0) { chain.push_back(x); x = (uint32_t)R.parent[x]; if (chain.size() > N) return fail(c, QS_ERR_ARG, "reference tree: cycle"); }
        int dd = dep[x];
        for (size_t i = chain.size(); i-- > 0;) dep[chain[i]] = ++dd;
    }
    for (uint32_t v = 0; v < N; ++v) R.depth[v] = (uint32_t)dep[v];
    std::vector<uint8_t> is_leaf_taxon(N, 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = ref->leaf_node[i];
        if (v >= N || R.nchild[v] != 0 || is_leaf_taxon[v]) return fail(c, QS_ERR_ARG, "reference tree: leaf_node must list distinct leaves");
        is_leaf_taxon[v] = 1;
    }
    uint32_t n_leaf_nodes = 0;
    for (uint32_t v = 0; v < N; ++v) if (R.nchild[v] == 0) n_leaf_nodes++;
    if (n_leaf_nodes != n) return fail(c, QS_ERR_ARG, "reference tree: number of leaves differs from n_taxa");
    // depth-first order check: every node's leaves form one contiguous id interval
    std::vector<uint32_t> lo(N, 0xFFFFFFFFu), hi(N, 0), cnt(N, 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t x = ref->leaf_node[i];
        for (;;) {
            lo[x] = std::min(lo[x], i); hi[x] = std::max(hi[x], i); cnt[x]++;
            if (R.parent[x] < 0) break;
            x = (uint32_t)R.parent[x];
        }
    }
    for (uint32_t v = 0; v < N; ++v)
        if (cnt[v] && hi[v] - lo[v] + 1 != cnt[v])
            return fail(c, QS_ERR_ARG, "reference tree: lookup ids are not in depth-first leaf order");
    // is_bifurcating: max over nodes of (degree - 1) == 2 (QuartetScoreComputer.hpp:760)
    uint32_t max_rank = 0;
    for (uint32_t v = 0; v < N; ++v) {
        uint32_t deg = R.nchild[v] + (R.parent[v] >= 0 ? 1 : 0);
        if (deg >= 1) max_rank = std::max(max_rank, deg - 1);
    }
    R.bifurcating = (max_rank == 2);
    R.leaf_lo = lo; R.leaf_cnt = cnt;
    R.root_deg2 = R.nchild[R.root] == 2;
    R.inner_id.assign(N, 0xFFFFFFFFu);
    R.inner_node.clear();
    for (uint32_t v = 0; v < N; ++v)
        if (R.nchild[v] > 0) { R.inner_id[v] = (uint32_t)R.inner_node.size(); R.inner_node.push_back(v); }
    R.n_inner = (uint32_t)R.inner_node.size();
    // LCA of leaves with adjacent ids, then running minima per row
    std::vector<uint32_t> adj(n > 0 ? n - 1 : 0);
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint32_t x = ref->leaf_node[i], y = ref->leaf_node[i + 1];
        while (x != y) {
            if (R.depth[x] >= R.depth[y]) x = (uint32_t)R.parent[x]; else y = (uint32_t)R.parent[y];
        }
        adj[i] = x;
    }
    R.lca.assign((size_t)n * n, 0);
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint32_t cur = adj[i];
        for (uint32_t j = i + 1; j < n; ++j) {
            if (j > i + 1 && R.depth[adj[j - 1]] < R.depth[cur]) cur = adj[j - 1];
            uint32_t e = (R.depth[cur] << 16) | R.inner_id[cur];
            R.lca[(size_t)i * n + j] = e;
            R.lca[(size_t)j * n + i] = e;
        }
    }
    if (R.bifurcating && R.root_deg2) {
        // QuartetScoreComputer.hpp:390-410 for u = root: S1 = the root's other child subtree, S2 = the child subtree that
        // holds v, S3 / S4 = v's child subtrees in depth-first order; all of them intervals of lookup ids
        std::vector<std::vector<uint32_t>> kids(N);
        for (uint32_t v = 0; v < N; ++v) if (R.parent[v] >= 0) kids[(uint32_t)R.parent[v]].push_back(v);
        for (auto &k : kids) std::sort(k.begin(), k.end(), [&](uint32_t p_, uint32_t q_) { return lo[p_] < lo[q_]; });
        const uint32_t rx = kids[R.root][0], ry = kids[R.root][1];
        R.root_split = cnt[rx];
        for (uint32_t v = 0; v < N; ++v) {
            if (v == R.root || R.nchild[v] != 2) continue;
            const bool in_x = lo[v] >= lo[rx] && lo[v] < lo[rx] + cnt[rx];
            const uint32_t mine = in_x ? rx : ry, other = in_x ? ry : rx;
            RootPairHost P{};
            P.s1_lo = lo[other]; P.s1_n = cnt[other]; P.s2_lo = lo[mine]; P.s2_n = cnt[mine];
            P.s3_lo = lo[kids[v][0]]; P.s3_n = cnt[kids[v][0]]; P.s4_lo = lo[kids[v][1]]; P.s4_n = cnt[kids[v][1]];
            const uint32_t ir = R.inner_id[R.root], iv = R.inner_id[v];
            P.key = std::min(ir, iv) * R.n_inner + std::max(ir, iv);
            P.first = R.root_items;
            R.root_items += (uint64_t)P.s1_n * P.s2_n * P.s3_n * P.s4_n;
            R.root_pairs.push_back(P);
        }
    }
    R.next.assign((size_t)n * n, 0);
    for (uint32_t b = 1; b < n; ++b) {
        R.next[(size_t)b * n + (b - 1)] = (uint16_t)b;
        for (uint32_t a = b - 1; a-- > 0;)
            R.next[(size_t)b * n + a] = R.lca[(size_t)b * n + a] != R.lca[(size_t)b * n + a + 1] ? (uint16_t)(a + 1) : R.next[(size_t)b * n + a + 1];
    }
    return QS_OK;
}

// The context's cached reference tree (host arrays + LCA matrix on the device when want_dev); rebuilt only when the
// caller passes a different tree.
static int get_ref(qs_ctx *c, const qs_ref_tree *ref, bool want_dev, const RefHost **out) {
    if (!ref || !ref->parent || !ref->leaf_node) return fail(c, QS_ERR_ARG, "reference tree: NULL arrays");
    RefHost *R = c->ref_cache;
    const bool same = R && R->n_nodes == ref->n_nodes && R->n == ref->n_taxa &&
                      memcmp(R->parent.data(), ref->parent, (size_t)ref->n_nodes * 4) == 0 &&
                      memcmp(R->leaf_node_in.data(), ref->leaf_node, (size_t)ref->n_taxa * 4) == 0;
    if (!same) {
        RefHost *fresh = new RefHost();
        int rc = build_ref(c, ref, *fresh);
        if (rc != QS_OK) { delete fresh; return rc; }
        fresh->leaf_node_in.assign(ref->leaf_node, ref->leaf_node + ref->n_taxa);
        if (c->ref_lca_dev) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ref_lca_dev); (void)hipFree(c->ref_next_dev); (void)hipFree(c->root_pairs_dev); c->ref_lca_dev = nullptr; c->ref_next_dev = nullptr; c->root_pairs_dev = nullptr; }
        delete c->ref_cache;
        c->ref_cache = R = fresh;
        c->log_valid = false;
    }
    if (want_dev && !c->ref_lca_dev) {
        QS_HIP(c, hipMalloc(&c->ref_lca_dev, R->lca.size() * 4));
        QS_HIP(c, hipMalloc(&c->ref_next_dev, R->next.size() * 2));
        hipStream_t up = c->prep_stream ? c->prep_stream : c->stream;
        QS_HIP(c, hipMemcpyAsync(c->ref_lca_dev, R->lca.data(), R->lca.size() * 4, hipMemcpyHostToDevice, up));
        QS_HIP(c, hipMemcpyAsync(c->ref_next_dev, R->next.data(), R->next.size() * 2, hipMemcpyHostToDevice, up));
        if (!R->root_pairs.empty()) {
            QS_HIP(c, hipMalloc(&c->root_pairs_dev, R->root_pairs.size() * sizeof(RootPairHost)));
            QS_HIP(c, hipMemcpyAsync(c->root_pairs_dev, R->root_pairs.data(), R->root_pairs.size() * sizeof(RootPairHost), hipMemcpyHostToDevice, up));
        }
    }
    *out = R;
    return QS_OK;
}

struct DevPtr { // RAII for a hipMalloc'ed pointer
    void *p = nullptr;
    ~DevPtr() { if (p) (void)hipFree(p); }
};

// log(k) and 1/k for the integer arguments of the device QIC: every count and every tuple sum is at most the
// number of trees counted. Unknown (uploaded / attached tables, reduce-scattered shards): 65536 entries, larger
// values take the kernel's slow path.
static int ensure_score_tables(qs_ctx *c) {
    // every count and every tuple sum is at most the number of trees behind the table: what this context counted, or what
    // the caller says stands behind a reduced / uploaded / viewed table (QS_TUNE_TABLE_TREES)
    const uint64_t want64 = std::min<uint64_t>(std::max<uint64_t>(std::max(c->trees_counted, c->table_trees_hint) + 1, 65536), 1ull << 20);
    const uint32_t want = (uint32_t)want64;
    if (c->tbl_n >= want) return QS_OK;
    if (c->dev_logk) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->dev_logk); c->dev_logk = nullptr; c->tbl_n = 0; }
    std::vector<double> lk(want);
    lk[0] = 0.0;
    for (uint32_t k = 1; k < want; ++k) lk[k] = std::log((double)k);
    QS_HIP(c, hipMalloc(&c->dev_logk, (size_t)want * 8));
    {   // (not the null stream: that one waits for count kernels still running on `stream`)
        hipStream_t up = c->prep_stream ? c->prep_stream : c->stream;
        QS_HIP(c, hipMemcpyAsync(c->dev_logk, lk.data(), (size_t)want * 8, hipMemcpyHostToDevice, up));
        QS_HIP(c, hipStreamSynchronize(up));
    }
    c->tbl_n = want;
    return QS_OK;
}

static void fill_score_device(const qs_ctx *c, const RefHost &R, const uint32_t *lca_dev, ScoreDevice &sd) {
    sd.logk = c->dev_logk; sd.tbl_n = c->tbl_n;
    // as much of the log table as fits goes to LDS (every tuple sum of up to 15743 trees); larger arguments are range-checked
    sd.coop_load = c->tune_score_load;
    sd.lds_n = (uint32_t)std::min<uint64_t>(c->tbl_n, score_scan_max_lds_log((c->tune_score_load == 1 || c->tune_score_load == 3) && c->tune_score_kernel == 0));
    sd.ref_lca = lca_dev; sd.ref_next = c->ref_next_dev; sd.n = c->n; sd.n_inner = R.n_inner; sd.d_lo = c->d_lo; sd.d_hi = c->d_hi;
    sd.rank_lo = c->rank_lo; sd.n_tuples = c->n_tuples; sd.table = c->table; sd.count_bits = (int)c->count_bits;
    if (c->view_table) { sd.rank_lo = c->view_rank_lo; sd.n_tuples = c->view_n; sd.table = const_cast<void *>(c->view_table); sd.count_bits = (int)c->view_bits; }
    sd.pair_sums = nullptr; sd.pair_min = nullptr; sd.pair_cand = nullptr; sd.flags = c->dev_flags + 1;
    sd.cand_limit = c->tune_cand_slots; sd.list = nullptr; sd.list_count = nullptr; sd.list_cap = 0; sd.last_trip = nullptr;
    sd.frame = R.bifurcating ? 0 : 1;
    sd.root_split = (R.bifurcating && R.root_deg2) ? R.root_split : 0u;
    sd.bundle_plo = sd.bundle_pcnt = sd.bundle_rounds = nullptr; sd.n_rounds = 0; sd.sample = 0;
}

// the bundle kernel's rounds for the rank range sd covers (own table, table shard or view): planned on the host, cached
static int ensure_bundle_plan(qs_ctx *c, ScoreDevice &sd, int pass) {
    sd.bundle_plo = sd.bundle_pcnt = sd.bundle_rounds = nullptr; sd.n_rounds = 0;
    if (c->tune_score_kernel == 1) return QS_OK;
    if (c->n_cu == 0) { int v = 0; QS_HIP(c, hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device)); c->n_cu = std::max(1, v); }
    const int w = pass - 1;
    BundlePlan &plan = c->bundle[w];
    const uint64_t r0 = sd.rank_lo, r1 = sd.rank_lo + sd.n_tuples;
    if (c->bundle_r0[w] != r0 || c->bundle_r1[w] != r1 || !c->bundle_dev[w]) {
        if (c->bundle_dev[w]) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->bundle_dev[w]); c->bundle_dev[w] = nullptr; }
        plan_bundles(c->n, r0, r1, score_bundle_waves(pass, c->tune_score_load == 1 || c->tune_score_load == 3), plan);
        const size_t words = 2 * (size_t)c->n + plan.rounds.size();
        if (hipMalloc(&c->bundle_dev[w], std::max<size_t>(words, 1) * 4) != hipSuccess) return fail(c, QS_ERR_OOM, "qs_score: round table of the bundle kernel");
        hipStream_t up = c->prep_stream ? c->prep_stream : c->stream;
        QS_HIP(c, hipMemcpyAsync(c->bundle_dev[w], plan.plo.data(), (size_t)c->n * 4, hipMemcpyHostToDevice, up));
        QS_HIP(c, hipMemcpyAsync(c->bundle_dev[w] + c->n, plan.pcnt.data(), (size_t)c->n * 4, hipMemcpyHostToDevice, up));
        if (!plan.rounds.empty())
            QS_HIP(c, hipMemcpyAsync(c->bundle_dev[w] + 2 * (size_t)c->n, plan.rounds.data(), plan.rounds.size() * 4, hipMemcpyHostToDevice, up));
        QS_HIP(c, hipStreamSynchronize(up));   // (the host vectors may be re-planned by the next call)
        c->bundle_r0[w] = r0; c->bundle_r1[w] = r1;
    }
    sd.bundle_plo = c->bundle_dev[w]; sd.bundle_pcnt = c->bundle_dev[w] + c->n; sd.bundle_rounds = c->bundle_dev[w] + 2 * (size_t)c->n;
    sd.n_rounds = (uint32_t)(plan.rounds.size() / 2);
    return QS_OK;
}

extern "C" int qs_score_plan(uint32_t n_taxa, uint64_t rank_lo, uint64_t n_tuples, uint32_t *first_pair, uint32_t *n_pairs, uint64_t parts[4]) {
    if (!first_pair || !n_pairs || !parts || n_taxa < 4 || n_taxa > 65535) return QS_ERR_ARG;
    if (rank_lo > binom4(n_taxa) || n_tuples > binom4(n_taxa) - rank_lo) return QS_ERR_ARG;
    BundlePlan plan;
    plan_bundles(n_taxa, rank_lo, rank_lo + n_tuples, score_bundle_waves(1), plan);
    std::copy(plan.plo.begin(), plan.plo.end(), first_pair);
    std::copy(plan.pcnt.begin(), plan.pcnt.end(), n_pairs);
    for (int i = 0; i < 2; ++i) { parts[2 * i] = i < plan.n_parts ? plan.part_lo[i] : 0; parts[2 * i + 1] = i < plan.n_parts ? plan.part_n[i] : 0; }
    return QS_OK;
}

extern "C" uint64_t qs_score_pair_slots(const qs_ref_tree *ref) {
    if (!ref || !ref->parent || ref->n_nodes == 0) return 0;
    std::vector<uint32_t> nchild(ref->n_nodes, 0);
    for (uint32_t v = 0; v < ref->n_nodes; ++v)
        if (ref->parent[v] >= 0 && (uint32_t)ref->parent[v] < ref->n_nodes) nchild[ref->parent[v]]++;
    uint64_t ni = 0;
    for (uint32_t v = 0; v < ref->n_nodes; ++v) ni += nchild[v] > 0;
    return ni * ni;
}

extern "C" int qs_score_set_view(qs_ctx *c, const void *table_dev, uint32_t count_bits, uint64_t rank_lo, uint64_t n_tuples) {
    if (c) c->log_valid = false;   // (the table / view / tuning may change: a logged pass 1 no longer describes it)
    if (!c) return QS_ERR_ARG;
    if (!table_dev) { c->view_table = nullptr; c->view_bits = 0; c->view_rank_lo = c->view_n = 0; return QS_OK; }
    if (count_bits != 16 && count_bits != 32) return fail(c, QS_ERR_ARG, "qs_score_set_view: count_bits must be 16 or 32");
    if (rank_lo > binom4(c->n) || n_tuples > binom4(c->n) - rank_lo) return fail(c, QS_ERR_ARG, "qs_score_set_view: rank range outside C(n,4)");
    c->view_table = table_dev; c->view_bits = count_bits; c->view_rank_lo = rank_lo; c->view_n = n_tuples;
    return QS_OK;
}

// qs_score's accumulators on the device and their pinned host copy, cached in the context
static int ensure_score_accumulators(qs_ctx *c, size_t np) {
    const size_t need_dev = np * (size_t)(3 + 1 + kCand) * 8, need_host = np * (size_t)(3 + kCand) * 8;
    if (c->score_acc_cap < need_dev) {
        if (c->score_acc) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->score_acc); }
        c->score_acc = nullptr; c->score_acc_cap = 0;
        QS_HIP(c, hipMalloc(&c->score_acc, need_dev));
        c->score_acc_cap = need_dev;
    }
    if (c->score_acc_host_cap < need_host) {
        if (c->score_acc_host) { QS_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->score_acc_host); }
        c->score_acc_host = nullptr; c->score_acc_host_cap = 0;
        QS_HIP(c, hipHostMalloc(&c->score_acc_host, need_host, hipHostMallocDefault));
        c->score_acc_host_cap = need_host;
    }
    return QS_OK;
}

// the candidate log of the single-read scoring: records, the counter word, one word per node pair (tie filter)
static bool ensure_score_log(qs_ctx *c, size_t np) {
    const uint64_t want_cap = c->tune_score_log_cap ? c->tune_score_log_cap : (1ull << 23);   // 8 M records = 256 MB
    if (c->score_log && (c->score_log_cap != want_cap || c->score_log_pairs < np)) {
        (void)hipStreamSynchronize(c->stream); (void)hipFree(c->score_log); c->score_log = nullptr; c->score_log_cap = 0; c->score_log_pairs = 0;
    }
    if (!c->score_log) {
        if (hipMalloc((void **)&c->score_log, (want_cap * 4 + 1 + np) * 8) == hipSuccess) { c->score_log_cap = want_cap; c->score_log_pairs = np; }
        else (void)hipGetLastError();   // no room for the log: two passes
    }
    return c->score_log != nullptr;
}

extern "C" int qs_score_pass1(qs_ctx *c, const qs_ref_tree *ref, int64_t *sums_dev, int64_t *min_dev) {
    if (!c || !sums_dev || !min_dev) return fail(c, QS_ERR_ARG, "qs_score_pass1: NULL");
    if (!c->table && !c->view_table) return fail(c, QS_ERR_STATE, "qs_score_pass1: no table");
    QS_HIP(c, hipSetDevice(c->device));
    const RefHost *Rp = nullptr;
    int rc = get_ref(c, ref, true, &Rp);
    if (rc != QS_OK) return rc;
    const RefHost &R = *Rp;
    const size_t np = (size_t)R.n_inner * R.n_inner;
    QS_HIP(c, hipMemsetAsync(sums_dev, 0, np * 3 * 8, c->stream));
    QS_HIP(c, hipMemsetAsync(min_dev, 0x7F, np * 8, c->stream));
    { int rc_t = ensure_score_tables(c); if (rc_t != QS_OK) return rc_t; }
    ScoreDevice sd;
    fill_score_device(c, R, c->ref_lca_dev, sd);
    sd.pair_sums = (unsigned long long *)sums_dev; sd.pair_min = (long long *)min_dev;
    { int rc_b = ensure_bundle_plan(c, sd, 1); if (rc_b != QS_OK) return rc_b; }
    // Single-read scoring: this pass also LOGS every quartet whose device QIC is within the tolerance of the bound it knows for
    // its node pair (the pair's minimum in memory, lowered beforehand by a minima-only pre-pass over a sample of the table, and
    // the lane's own running minimum) -- a superset of the quartets within the tolerance of ANY later minimum <= this table's,
    // so the qs_score_pass2 that follows filters the log instead of reading the table again, also when its min_dev holds the
    // minima over several shards / GPUs (QuartetScoreComputer.hpp:417-469 evaluates log_score of every quartet once, too).
    // QS_TUNE_SCORE_PASSES: 0 = automatic (tables from 1 GB, whole rows, bundle kernel; a second sample predicts the log and
    // the pass runs plain when it would not hold), 1 = never, 2 = always (pass 2 reads the table if the log overflows).
    c->log_valid = false;
    c->last_score_estimate = 0;
    const uint64_t scored_bytes = sd.n_tuples * 3 * (uint64_t)(sd.count_bits / 8);
    const bool want_log = c->tune_score_kernel == 0 && c->bundle[0].n_parts == 0 && sd.n_rounds > 0 &&
                          (c->tune_score_passes == 2 || (c->tune_score_passes == 0 && c->tune_score_sample != 0 && scored_bytes >= (1ull << 30)));
    bool logging = false;
    if (want_log) {
        logging = ensure_score_log(c, np);
    }
    if (logging) {
        sd.list = c->score_log; sd.list_count = c->score_log + 4 * c->score_log_cap; sd.list_cap = c->score_log_cap;
        QS_HIP(c, hipMemsetAsync(sd.list_count, 0, 8, c->stream));
        if (c->tune_score_dedupe) {   // the tie filter (one word per node pair behind the log's counter)
            sd.last_trip = sd.list_count + 1;
            QS_HIP(c, hipMemsetAsync(sd.last_trip, 0xFF, np * 8, c->stream));
        }
        if (c->tune_score_sample) {   // minima-only pre-pass over a sample of the table: the bound the logging pass starts from
            ScoreDevice pre = sd;
            pre.list = nullptr; pre.list_count = nullptr; pre.list_cap = 0; pre.sample = c->tune_score_sample;
            QS_HIP(c, launch_score_pass1(c->stream, pre, 0, c->n_cu, nullptr, nullptr, 0, 0.0));
            if (c->tune_score_passes != 2) {
                // automatic mode: a second, disjoint sample counts what the logging pass would log of it
                pre.list_count = sd.list_count; pre.sample = c->tune_score_sample | (1u << 17);
                QS_HIP(c, launch_score_pass1(c->stream, pre, 0, c->n_cu, nullptr, nullptr, 0, c->tune_score_tol));
                unsigned long long hits = 0;
                QS_HIP(c, hipMemcpyAsync(&hits, sd.list_count, 8, hipMemcpyDeviceToHost, c->stream));
                QS_HIP(c, hipStreamSynchronize(c->stream));
                c->last_score_estimate = hits * (c->tune_score_sample & 0xFFFFu);
                // The static bounds of the sample over-predict: the full pass also lowers the minima as it goes, a lane keeps its
                // own running minimum, and the tie filter has seen 64 x more of every node pair (measured, predicted / logged:
                // 12.6 M / 1.6 M at 512 taxa x 10000 random trees, 3.7 M / 0.75 M at 256 taxa, 19 M / 1.0 M at 512 taxa x 10000
                // reference + NNI trees; without the tie filter those predict 90 M and do overflow 8 M): go ahead up to 6 x the log.
                if ((double)c->last_score_estimate > 6.0 * (double)c->score_log_cap) {
                    logging = false;
                    sd.list = nullptr; sd.list_count = nullptr; sd.list_cap = 0;
                } else {
                    QS_HIP(c, hipMemsetAsync(sd.list_count, 0, 8, c->stream));
                    if (sd.last_trip) QS_HIP(c, hipMemsetAsync(sd.last_trip, 0xFF, np * 8, c->stream));   // (the estimate logged nothing)
                }
            }
        }
        // waves reserve whole chunks of records: what they leave unwritten must read as "no record" (key = all ones)
        if (logging) {
            QS_HIP(c, hipMemsetAsync(c->score_log, 0xFF, (size_t)c->score_log_cap * 32, c->stream));
            c->log_valid = true;                 // until the table, the view or the tuning changes (invalidate_log)
            c->log_table = sd.table; c->log_rank_lo = sd.rank_lo; c->log_n_tuples = sd.n_tuples; c->log_ref = Rp; c->log_tol = c->tune_score_tol;
        }
    }
    QS_HIP(c, launch_score_pass1(c->stream, sd, c->tune_score_kernel, c->n_cu, c->bundle[0].part_lo, c->bundle[0].part_n, c->bundle[0].n_parts, c->tune_score_tol));
    // rooted reference (degree-2 root): the sums of the node pairs (root, v) as the reference enumerates them (quirk Q5)
    if (c->root_pairs_dev) QS_HIP(c, launch_root_pair_sums(c->stream, sd, c->root_pairs_dev, (uint32_t)R.root_pairs.size(), R.root_items));
    return QS_OK;   // asynchronous on the context's stream
}

// Everything a first qs_score pays before its kernels -- reference tree + LCA matrix on the device, log table, the bundle
// kernel's round tables, the accumulators with their pinned host copy, the candidate log, the kernels' code objects' first
// use of the device-to-host copy path -- done ahead of time. Safe while count kernels of this context are still running (same
// host thread): the uploads go through the copy stream, nothing waits for `stream`. The CLI calls it after the last batch
// is enqueued, where the host would only wait: the scoring phase of 512 taxa x 10000 trees drops from 28 to ~15 ms.
extern "C" int qs_score_prepare(qs_ctx *c, const qs_ref_tree *ref, uint64_t n_trees_total) {
    if (!c || !ref) return fail(c, QS_ERR_ARG, "qs_score_prepare: NULL");
    QS_HIP(c, hipSetDevice(c->device));
    if (!c->copy_stream) QS_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // every exit path -- also the error returns below -- leaves the copy stream drained: what was uploaded through it is marked
    // present in the context, and the kernels of a later qs_score run on `stream`, which has no ordering against `copy_stream`
    struct Restore { qs_ctx *c; ~Restore() { c->prep_stream = nullptr; (void)hipStreamSynchronize(c->copy_stream); } } restore{c};
    c->prep_stream = c->copy_stream;
    const bool had_ref = c->ref_lca_dev != nullptr;
    const RefHost *Rp = nullptr;
    int rc = get_ref(c, ref, !had_ref || true, &Rp);
    if (rc != QS_OK) return rc;
    const size_t np = (size_t)Rp->n_inner * Rp->n_inner;
    if (np == 0) return fail(c, QS_ERR_ARG, "qs_score_prepare: bad reference tree");
    const uint64_t keep_hint = c->table_trees_hint;
    c->table_trees_hint = std::max<uint64_t>(c->table_trees_hint, n_trees_total);   // size the log table for the finished table
    rc = ensure_score_tables(c);
    c->table_trees_hint = keep_hint;
    if (rc != QS_OK) return rc;
    ScoreDevice sd;
    fill_score_device(c, *Rp, c->ref_lca_dev, sd);
    rc = ensure_bundle_plan(c, sd, 1);
    if (rc == QS_OK) rc = ensure_bundle_plan(c, sd, 2);
    if (rc != QS_OK) return rc;
    rc = ensure_score_accumulators(c, np);
    if (rc != QS_OK) return rc;
    const uint64_t scored_bytes = sd.n_tuples * 3 * (uint64_t)(sd.count_bits / 8);
    if (c->tune_score_kernel == 0 && c->bundle[0].n_parts == 0 && (c->tune_score_passes == 2 || (c->tune_score_passes == 0 && c->tune_score_sample != 0 && scored_bytes >= (1ull << 30))))
        (void)ensure_score_log(c, np);
    for (hipEvent_t &e : c->score_ev) if (!e) QS_HIP(c, hipEventCreate(&e));
    // first use of the pinned accumulator copy by the device (the first device-to-host copy into freshly pinned memory took 7 ms)
    QS_HIP(c, hipMemcpyAsync(c->score_acc_host, c->score_acc, np * (size_t)(3 + kCand) * 8, hipMemcpyDeviceToHost, c->copy_stream));
    QS_HIP(c, hipStreamSynchronize(c->copy_stream));
    return QS_OK;
}

extern "C" int qs_score_pass2(qs_ctx *c, const qs_ref_tree *ref, const int64_t *min_dev, int64_t *cand_dev) {
    if (!c || !min_dev || !cand_dev) return fail(c, QS_ERR_ARG, "qs_score_pass2: NULL");
    if (!c->table && !c->view_table) return fail(c, QS_ERR_STATE, "qs_score_pass2: no table");
    QS_HIP(c, hipSetDevice(c->device));
    const RefHost *Rp = nullptr;
    int rc = get_ref(c, ref, true, &Rp);
    if (rc != QS_OK) return rc;
    const RefHost &R = *Rp;
    const size_t np = (size_t)R.n_inner * R.n_inner;
    QS_HIP(c, hipMemsetAsync(cand_dev, 0xFF, np * kCand * 8, c->stream));
    QS_HIP(c, hipMemsetAsync(c->dev_flags + 1, 0, 4, c->stream));
    { int rc_t = ensure_score_tables(c); if (rc_t != QS_OK) return rc_t; }
    ScoreDevice sd;
    fill_score_device(c, R, c->ref_lca_dev, sd);
    sd.pair_min = (long long *)min_dev; sd.pair_cand = (unsigned long long *)cand_dev;
    c->last_score_log = 0;
    if (c->log_valid && c->log_table == sd.table && c->log_rank_lo == sd.rank_lo && c->log_n_tuples == sd.n_tuples && c->log_ref == Rp &&
        c->log_tol == c->tune_score_tol) {
        // the preceding qs_score_pass1 over this very table logged its candidates: filter the log against min_dev
        c->log_valid = false;
        unsigned long long n_rec = 0;
        QS_HIP(c, hipMemcpyAsync(&n_rec, c->score_log + 4 * c->score_log_cap, 8, hipMemcpyDeviceToHost, c->stream));
        QS_HIP(c, hipStreamSynchronize(c->stream));
        // A counter that reached the capacity is an overflow: waves reserve chunks of 64 records, the capacity is a multiple
        // of 64, so the counter can stop exactly AT the capacity while waves that found the log full at the start of a row
        // have skipped their hits without moving it. Only a log with room left is known to be complete.
        if (n_rec < c->score_log_cap) {
            sd.list = c->score_log; sd.list_cap = c->score_log_cap;
            QS_HIP(c, launch_score_log(c->stream, sd, c->tune_score_tol, n_rec));
            c->last_score_log = n_rec;
            return QS_OK;
        }
        // (the log overflowed: read the table)
    }
    { int rc_b = ensure_bundle_plan(c, sd, 2); if (rc_b != QS_OK) return rc_b; }
    QS_HIP(c, launch_score_pass2(c->stream, sd, c->tune_score_tol, c->tune_score_kernel, c->n_cu, c->bundle[1].part_lo, c->bundle[1].part_n, c->bundle[1].n_parts));
    return QS_OK;   // asynchronous; node pairs whose slots did not suffice are marked in cand_dev (qs_score_overflow)
}

// Node pairs that pass 2 could not finish in their 8 packed slots (more than 8 distinct near-minimal triples, or a
// gcd-reduced count >= 2^21) carry kCandOverflow in their last slot. For those pairs the table is scanned once more
// and EVERY near-minimal quartet's (key, q1, q2, q3) is listed; the list is sorted and de-duplicated on the host and
// handed to qs_score_finish, which takes the exact minimum over it. The reference evaluates log_score for every
// quartet (QuartetScoreComputer.hpp:417-469), so any superset of the minimisers gives its result.
extern "C" int qs_score_overflow(qs_ctx *c, const qs_ref_tree *ref, const int64_t *min_dev, const int64_t *cand_dev,
                                 int64_t **list_out, uint64_t *n_out) {
    if (!c || !min_dev || !cand_dev || !list_out || !n_out) return fail(c, QS_ERR_ARG, "qs_score_overflow: NULL");
    *list_out = nullptr; *n_out = 0;
    QS_HIP(c, hipSetDevice(c->device));
    uint32_t fl = 0;
    QS_HIP(c, hipMemcpyAsync(&fl, c->dev_flags + 1, 4, hipMemcpyDeviceToHost, c->stream));
    QS_HIP(c, hipStreamSynchronize(c->stream));
    if ((fl & 3u) == 0) return QS_OK;
    const RefHost *Rp = nullptr;
    int rc = get_ref(c, ref, true, &Rp);
    if (rc != QS_OK) return rc;
    ScoreDevice sd;
    fill_score_device(c, *Rp, c->ref_lca_dev, sd);
    sd.pair_min = (long long *)const_cast<int64_t *>(min_dev); sd.pair_cand = (unsigned long long *)const_cast<int64_t *>(cand_dev);
    DevPtr cnt, list;
    QS_HIP(c, hipMalloc(&cnt.p, 8));
    uint64_t cap = 1ull << 20;
    std::vector<unsigned long long> host;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (list.p) { (void)hipFree(list.p); list.p = nullptr; }
        if (hipMalloc(&list.p, cap * 32) != hipSuccess) return fail(c, QS_ERR_OOM, "qs_score_overflow: list of " + std::to_string(cap) + " quartets does not fit");
        QS_HIP(c, hipMemsetAsync(cnt.p, 0, 8, c->stream));
        sd.list = (unsigned long long *)list.p; sd.list_count = (unsigned long long *)cnt.p; sd.list_cap = cap;
        QS_HIP(c, launch_score_overflow_list(c->stream, sd, c->tune_score_tol));
        unsigned long long got = 0;
        QS_HIP(c, hipMemcpyAsync(&got, cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
        QS_HIP(c, hipStreamSynchronize(c->stream));
        if (got <= cap) {
            host.resize(got * 4);
            if (got) QS_HIP(c, hipMemcpy(host.data(), list.p, got * 32, hipMemcpyDeviceToHost));
            break;
        }
        if (attempt == 1 || got > (1ull << 28)) return fail(c, QS_ERR_OVERFLOW, "qs_score_overflow: " + std::to_string(got) + " near-minimal quartets");
        cap = got;
    }
    // sort + unique the 4-word records
    const size_t n = host.size() / 4;
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = i;
    auto less = [&](size_t x, size_t y) { return std::lexicographical_compare(&host[4 * x], &host[4 * x] + 4, &host[4 * y], &host[4 * y] + 4); };
    std::sort(idx.begin(), idx.end(), less);
    std::vector<unsigned long long> uniq;
    for (size_t i = 0; i < n; ++i)
        if (i == 0 || less(idx[i - 1], idx[i])) uniq.insert(uniq.end(), &host[4 * idx[i]], &host[4 * idx[i]] + 4);
    if (!uniq.empty()) {
        int64_t *out = (int64_t *)malloc(uniq.size() * 8);
        if (!out) return fail(c, QS_ERR_OOM, "qs_score_overflow: host list");
        memcpy(out, uniq.data(), uniq.size() * 8);
        *list_out = out; *n_out = uniq.size() / 4;
    }
    return QS_OK;
}

extern "C" void qs_free_host(void *p) { free(p); }

// QS_SCORE_SAVEMEM_LOOKUPS + a degree-2 reference root. For a node pair (root, v) the reference calls
// countQuartetOccurrences(a, b, c, d) with a in S1 (the root's other side), b in S2 = ALL leaves on v's side, c in S3, d in S4
// (v's child subtrees; QuartetScoreComputer.hpp:393-396,412-424): b repeats c or d. Its compact table sorts the ids
// (quartet_lookup_table.hpp:170-212) and the const get_tuple throws std::runtime_error("id = ..., but quartet_lookup_.size() =
// ...") when the index C(t1,4)+C(t2,3)+C(t3,2)+t4 of the sorted ids t1 >= t2 >= t3 >= t4 lies behind the table (:79-85), e.g.
// whenever the two largest are both n-1: for every rooted reference tree with 4 or more taxa some call does (proved case by
// case in DESIGN.md 1, checked on the unmodified header in tests/test_oracle_reftable.py). The exception is not caught
// anywhere in the reference: its run ends there. This finds the FIRST throwing call in the reference's sequential order
// (pairs by node index; a, b, c, d nested in that order, each ascending) without walking the O(n^4) calls: the index is
// monotone in every id, so the calls of a prefix throw iff the one with the largest remaining ids does.
static uint64_t dup_index(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {   // lookup_index_ on ids that may repeat
    uint32_t t[4] = {a, b, c, d};
    std::sort(t, t + 4);
    return binom4(t[3]) + binom3(t[2]) + binom2(t[1]) + t[0];
}
static bool first_compact_throw(const RefHost &R, uint64_t nq, uint64_t *id_out) {
    for (const RootPairHost &P : R.root_pairs) {
        if (!P.s1_n || !P.s3_n || !P.s4_n) continue;
        const uint32_t a_hi = P.s1_lo + P.s1_n - 1, c_hi = P.s3_lo + P.s3_n - 1, d_hi = P.s4_lo + P.s4_n - 1;
        // the largest index any call of this pair reaches: b repeats the largest c or the largest d
        if (std::max(dup_index(a_hi, c_hi, c_hi, d_hi), dup_index(a_hi, d_hi, c_hi, d_hi)) < nq) continue;
        for (uint32_t a = P.s1_lo; a <= a_hi; ++a) {
            if (std::max(dup_index(a, c_hi, c_hi, d_hi), dup_index(a, d_hi, c_hi, d_hi)) < nq) continue;
            for (uint32_t b = P.s2_lo; b < P.s2_lo + P.s2_n; ++b) {
                const bool in3 = b >= P.s3_lo && b <= c_hi, in4 = b >= P.s4_lo && b <= d_hi;
                if (in3) {            // calls (a, b, c = b, d), d ascending: a throwing d exists iff the largest throws
                    if (dup_index(a, b, b, d_hi) < nq) continue;
                    for (uint32_t d = P.s4_lo; d <= d_hi; ++d)
                        if (dup_index(a, b, b, d) >= nq) { *id_out = dup_index(a, b, b, d); return true; }
                } else if (in4) {     // calls (a, b, c, d = b), c ascending
                    if (dup_index(a, b, c_hi, b) < nq) continue;
                    for (uint32_t c = P.s3_lo; c <= c_hi; ++c)
                        if (dup_index(a, b, c, b) >= nq) { *id_out = dup_index(a, b, c, b); return true; }
                }
            }
        }
    }
    return false;
}
static int check_savemem_lookups(qs_ctx *c, const RefHost &R, uint32_t flags) {
    if (!(flags & QS_SCORE_SAVEMEM_LOOKUPS) || (flags & QS_SCORE_ROOT_AS_EDGE) || !R.bifurcating || !R.root_deg2) return QS_OK;
    const uint64_t nq = binom4(R.n);
    uint64_t id = 0;
    if (!first_compact_throw(R, nq, &id)) return QS_OK;
    return fail(c, QS_ERR_REFERENCE_THROWS, "id = " + std::to_string(id) + ", but quartet_lookup_.size() = " + std::to_string(nq));
}

// Host-only: what qs_score / qs_score_finish would refuse for this reference tree and these flags, found from the tree alone
// (before any counting): today the QS_SCORE_SAVEMEM_LOOKUPS + rooted-reference case. ctx may be NULL.
extern "C" int qs_score_check(qs_ctx *c, const qs_ref_tree *ref, uint32_t flags) {
    if (!ref) return fail(c, QS_ERR_ARG, "qs_score_check: NULL");
    RefHost local;
    const RefHost *Rp = &local;
    const int rc = c ? get_ref(c, ref, false, &Rp) : build_ref(nullptr, ref, local);
    if (rc != QS_OK) return rc;
    return check_savemem_lookups(c, *Rp, flags);
}

// Pure host: log_score of the O(#node pairs) candidates and sums with the host libm
// (QuartetScoreComputer.hpp:135-159), then the min-propagation along path(u,v) (:448-454, :472-489).
extern "C" int qs_score_finish(qs_ctx *c, const qs_ref_tree *ref, uint32_t flags, const int64_t *sums_host,
                               const int64_t *cand_host, uint32_t n_cand_parts, const int64_t *extra_host, uint64_t n_extra,
                               double *lqic, double *qpic, double *eqpic, int *is_bifurcating) {
    if (!sums_host || !cand_host || !lqic || n_cand_parts == 0) return fail(c, QS_ERR_ARG, "qs_score_finish: NULL");
    RefHost local;          // ctx == NULL: pure host use (no device, no cache)
    const RefHost *Rp = &local;
    int rc = c ? get_ref(c, ref, false, &Rp) : build_ref(nullptr, ref, local);
    if (rc != QS_OK) return rc;
    const RefHost &R = *Rp;
    const bool root_as_edge = (flags & QS_SCORE_ROOT_AS_EDGE) != 0;
    if (is_bifurcating) *is_bifurcating = R.bifurcating ? 1 : 0;
    { int rc_s = check_savemem_lookups(c, R, flags); if (rc_s != QS_OK) return rc_s; }
    if (R.bifurcating && (!qpic || !eqpic)) return fail(c, QS_ERR_ARG, "qs_score: qpic/eqpic required for a bifurcating reference");
    const size_t np = (size_t)R.n_inner * R.n_inner;
    const unsigned long long *sums = (const unsigned long long *)sums_host;
    const unsigned long long *cand = (const unsigned long long *)cand_host;
    // extra candidates (qs_score_overflow lists, any order): key -> range in a sorted copy
    std::vector<std::array<unsigned long long, 4>> extra;
    // A record flagged kListSwap / a slot flagged kCandSwap is a quartet the reference evaluates twice for a degree-2 root, the
    // second time with q2 and q3 exchanged (qs_score.hip root_swapped; QuartetScoreComputer.hpp:393-396,417-454): both orders
    // enter the minimum. QS_SCORE_ROOT_AS_EDGE does not reproduce the reference's root pairs, hence not this either.
    const bool both_orders = !root_as_edge;
    if (extra_host && n_extra) {
        extra.reserve(n_extra);
        for (uint64_t i = 0; i < n_extra; ++i) {
            const unsigned long long *e = (const unsigned long long *)extra_host + 4 * i;
            extra.push_back({e[0] & 0xFFFFFFFFull, e[1], e[2], e[3]});
            if ((e[0] & kListSwap) && both_orders) extra.push_back({e[0] & 0xFFFFFFFFull, e[1], e[3], e[2]});
        }
        std::sort(extra.begin(), extra.end());
    }
    const double inf = std::numeric_limits<double>::infinity();
    const uint32_t N = R.n_nodes;
    for (uint32_t v = 0; v < N; ++v) {
        lqic[v] = inf;
        if (R.bifurcating) { qpic[v] = inf; eqpic[v] = inf; }
    }
    // The O(#node pairs) finalisation runs on a few host threads for large references. Each thread folds a contiguous
    // range of iu (pairs are visited in (iu, iv) order) into private vectors; merging the ranges in order reproduces
    // the sequential result exactly, including which of two equal minima / which of two writers of a QP value wins.
    struct Local { std::vector<double> lq, qp, eqp; std::vector<uint8_t> qp_set; };
    auto fold = [&](uint32_t iu0, uint32_t iu1, Local &out) {
        out.lq.assign(N, inf);
        if (R.bifurcating) { out.qp.assign(N, inf); out.eqp.assign(N, inf); out.qp_set.assign(N, 0); }
        for (uint32_t iu = iu0; iu < iu1; ++iu)
            for (uint32_t iv = iu + 1; iv < R.n_inner; ++iv) {
                const size_t key = (size_t)iu * R.n_inner + iv;
                double lqmin = inf;
                bool any = false;
                for (uint32_t part = 0; part < n_cand_parts; ++part) {
                    const unsigned long long *cs = cand + ((size_t)part * np + key) * kCand;
                    for (int s = 0; s < kCand && cs[s] != kCandEmpty; ++s) {
                        if (cs[s] == kCandOverflow) continue;   // marker: this pair's full list is in `extra`
                        const uint64_t q1 = (cs[s] >> 42) & 0x1FFFFFu, q2 = (cs[s] >> 21) & 0x1FFFFFu, q3 = cs[s] & 0x1FFFFFu;
                        const double v = host_log_score(q1, q2, q3);
                        lqmin = std::min(lqmin, v);
                        if ((cs[s] & kCandSwap) && both_orders) lqmin = std::min(lqmin, host_log_score(q1, q3, q2));
                        any = true;
                    }
                }
                if (!extra.empty()) {
                    const std::array<unsigned long long, 4> lo_key = {(unsigned long long)key, 0, 0, 0};
                    for (auto it = std::lower_bound(extra.begin(), extra.end(), lo_key); it != extra.end() && (*it)[0] == key; ++it) {
                        lqmin = std::min(lqmin, host_log_score((*it)[1], (*it)[2], (*it)[3]));
                        any = true;
                    }
                }
                if (!any) continue; // pair owns no resolved quartet
                uint64_t p1 = sums[key * 3], p2 = sums[key * 3 + 1], p3 = sums[key * 3 + 2];
                if (!(flags & QS_SCORE_QP_EXACT64)) { p1 &= 0xFFFFFFFFull; p2 &= 0xFFFFFFFFull; p3 &= 0xFFFFFFFFull; } // QSC:382
                const double qp = host_log_score(p1, p2, p3);
                // walk the path u..v (edges are indexed by their child node)
                uint32_t x = R.inner_node[iu], y = R.inner_node[iv];
                uint32_t path_edges = 0, last_edge_a = 0, last_edge_b = 0;
                while (x != y) {
                    uint32_t e;
                    if (R.depth[x] >= R.depth[y]) { e = x; x = (uint32_t)R.parent[x]; }
                    else { e = y; y = (uint32_t)R.parent[y]; }
                    out.lq[e] = std::min(out.lq[e], lqmin);
                    if (R.bifurcating) out.eqp[e] = std::min(out.eqp[e], qp);
                    if (path_edges == 0) last_edge_a = e; else last_edge_b = e;
                    ++path_edges;
                }
                if (R.bifurcating) {
                    const bool through_deg2_root = root_as_edge && path_edges == 2 && x == R.root && R.nchild[R.root] == 2;
                    if (path_edges == 1) { out.qp[last_edge_a] = qp; out.qp_set[last_edge_a] = 1; }
                    else if (through_deg2_root) { out.qp[last_edge_a] = qp; out.qp[last_edge_b] = qp; out.qp_set[last_edge_a] = out.qp_set[last_edge_b] = 1; }
                }
            }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned workers = R.n_inner >= 128 ? std::min(8u, hw) : 1u;
    std::vector<Local> parts(workers + 1);
    if (workers == 1) fold(0, R.n_inner, parts[0]);
    else {
        // ranges of iu with about the same number of pairs (row iu has n_inner - 1 - iu of them)
        std::vector<uint32_t> cut(workers + 1, R.n_inner);
        const uint64_t total = (uint64_t)R.n_inner * (R.n_inner - 1) / 2;
        uint64_t acc = 0;
        uint32_t w = 1;
        cut[0] = 0;
        for (uint32_t iu = 0; iu < R.n_inner && w < workers; ++iu) {
            acc += R.n_inner - 1 - iu;
            while (w < workers && acc >= total * w / workers) cut[w++] = iu + 1;
        }
        std::vector<std::thread> pool;
        for (unsigned k = 0; k < workers; ++k) pool.emplace_back([&, k] { fold(cut[k], cut[k + 1], parts[k]); });
        for (auto &th : pool) th.join();
    }
    if (R.bifurcating && R.root_deg2 && !root_as_edge) {
        // Reference-compatible handling of a degree-2 root (quirk Q5, QuartetScoreComputer.hpp:393-396,472-489): every
        // pair (root, v) has its own sums (root_pair_sums_kernel); QP-IC of the edge if v is a child of the root, EQP-IC
        // minimum along the path. (Pairs are visited with u = root first in the reference: these come before all others.)
        Local &L = parts[workers];
        L.lq.assign(N, inf); L.qp.assign(N, inf); L.eqp.assign(N, inf); L.qp_set.assign(N, 0);
        for (const RootPairHost &P : R.root_pairs) {
            uint64_t p1 = sums[(size_t)P.key * 3], p2 = sums[(size_t)P.key * 3 + 1], p3 = sums[(size_t)P.key * 3 + 2];
            if (!(flags & QS_SCORE_QP_EXACT64)) { p1 &= 0xFFFFFFFFull; p2 &= 0xFFFFFFFFull; p3 &= 0xFFFFFFFFull; }
            const double qp = host_log_score(p1, p2, p3);
            const uint32_t iu = P.key / R.n_inner, iv = P.key % R.n_inner;
            uint32_t v = R.inner_node[iu] == R.root ? R.inner_node[iv] : R.inner_node[iu];
            if ((uint32_t)R.parent[v] == R.root) { L.qp[v] = qp; L.qp_set[v] = 1; }
            for (uint32_t x = v; x != R.root; x = (uint32_t)R.parent[x]) L.eqp[x] = std::min(L.eqp[x], qp);
        }
        std::rotate(parts.begin(), parts.begin() + workers, parts.end());   // merged first
    } else parts.pop_back();
    for (const Local &L : parts) // in range order
        for (uint32_t v = 0; v < N; ++v) {
            lqic[v] = std::min(lqic[v], L.lq[v]);
            if (R.bifurcating) {
                eqpic[v] = std::min(eqpic[v], L.eqp[v]);
                if (L.qp_set[v]) qpic[v] = L.qp[v];
            }
        }
    return QS_OK;
}

extern "C" int qs_score(qs_ctx *c, const qs_ref_tree *ref, uint32_t flags, double *lqic, double *qpic, double *eqpic,
                        int *is_bifurcating) {
    if (!c || !lqic) return fail(c, QS_ERR_ARG, "qs_score: NULL");
    if (!c->table) return fail(c, QS_ERR_STATE, "qs_score: no table");
    if (c->d_lo != 0 || c->d_hi != c->n)
        return fail(c, QS_ERR_UNSUPPORTED, "qs_score: this context owns a table shard; use qs_score_pass1 / qs_score_pass2 / "
                                           "qs_score_finish with reductions over the shards in between");
    const size_t np = (size_t)qs_score_pair_slots(ref);
    if (np == 0) return fail(c, QS_ERR_ARG, "qs_score: bad reference tree");
    QS_HIP(c, hipSetDevice(c->device));
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); };
    const clk::time_point t_all = clk::now();
    for (hipEvent_t &e : c->score_ev) if (!e) QS_HIP(c, hipEventCreate(&e));
    for (float &x : c->score_ms) x = 0;
    // the accumulators (sums, minima, candidate slots: 24 MB at 512 taxa) and their pinned host copy live in the context: three
    // hipMalloc / hipFree pairs and a pageable 22 MB copy per call were 1 ms of a 15 ms call
    struct Ptr { void *p; } sums{nullptr}, mn{nullptr}, cand{nullptr};
    { int rc_a = ensure_score_accumulators(c, np); if (rc_a != QS_OK) return rc_a; }
    sums.p = c->score_acc; mn.p = (char *)c->score_acc + np * 3 * 8; cand.p = (char *)c->score_acc + np * 4 * 8;
    {   // the cached pieces both passes need (reference tree + LCA matrix, log table, bundle plans): built here so that
        // the first call's set-up cost shows as its own phase and the events below bracket kernels only
        const RefHost *Rp = nullptr;
        int rc0 = get_ref(c, ref, true, &Rp);
        if (rc0 != QS_OK) return rc0;
        rc0 = check_savemem_lookups(c, *Rp, flags);      // (the reference throws during its scoring loop: before any kernel here)
        if (rc0 != QS_OK) return rc0;
        rc0 = ensure_score_tables(c);
        if (rc0 != QS_OK) return rc0;
    }
    c->score_ms[1] = ms_since(t_all);
    QS_HIP(c, hipEventRecord(c->score_ev[0], c->stream));
    int rc = qs_score_pass1(c, ref, (int64_t *)sums.p, (int64_t *)mn.p);     // (may log its candidates: see there)
    if (rc != QS_OK) return rc;
    QS_HIP(c, hipEventRecord(c->score_ev[1], c->stream));
    rc = qs_score_pass2(c, ref, (const int64_t *)mn.p, (int64_t *)cand.p);   // a filter over that log, or the second read
    if (rc != QS_OK) return rc;
    QS_HIP(c, hipEventRecord(c->score_ev[2], c->stream));
    const clk::time_point t_ov = clk::now();
    int64_t *extra = nullptr;
    uint64_t n_extra = 0;
    rc = qs_score_overflow(c, ref, (const int64_t *)mn.p, (const int64_t *)cand.p, &extra, &n_extra);   // synchronises the stream
    if (rc != QS_OK) return rc;
    struct FreeHost { int64_t *p; ~FreeHost() { free(p); } } free_extra{extra};
    // The accumulators come home through the COPY stream (the count stream is idle: qs_score_overflow has just synchronised
    // it). The first large device-to-host copy enqueued on a stream that has run kernels costs ~8 ms on the host, once per
    // stream (tools/d2h_cold.hip); a stream that only ever copies does not pay it -- that was most of a first call's
    // "wait + copies" phase.
    int64_t *hs = (int64_t *)c->score_acc_host, *hc = hs + np * 3;
    if (!c->copy_stream) QS_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    QS_HIP(c, hipMemcpyAsync(hs, sums.p, np * 3 * 8, hipMemcpyDeviceToHost, c->copy_stream));
    QS_HIP(c, hipMemcpyAsync(hc, cand.p, np * kCand * 8, hipMemcpyDeviceToHost, c->copy_stream));
    QS_HIP(c, hipStreamSynchronize(c->copy_stream));
    (void)hipEventElapsedTime(&c->score_ms[2], c->score_ev[0], c->score_ev[1]);
    (void)hipEventElapsedTime(&c->score_ms[3], c->score_ev[1], c->score_ev[2]);
    c->score_ms[4] = ms_since(t_ov);            // waiting for the passes + overflow pass (if any) + the accumulators' way back
    const clk::time_point t_fin = clk::now();
    rc = qs_score_finish(c, ref, flags, hs, hc, 1, extra, n_extra, lqic, qpic, eqpic, is_bifurcating);
    c->score_ms[5] = ms_since(t_fin);
    c->score_ms[0] = ms_since(t_all);
    return rc;
}

// Phases of the last qs_score call in ms: [0] whole call (host clock), [1] set-up (accumulator allocation, reference tree
// + LCA matrix, log table; near zero once cached), [2] pass 1 and [3] pass 2 (HIP events on the context's stream: kernels
// + the bundle plan of a first call), [4] host wait for the passes incl. overflow pass and device-to-host copies,
// [5] qs_score_finish (host libm + min-propagation).
// Records the candidate log of the most recent qs_score held (single-read mode); 0 = it took two passes over the table.
extern "C" uint64_t qs_last_score_log(const qs_ctx *c) { return c ? c->last_score_log : 0; }
extern "C" uint64_t qs_last_score_estimate(const qs_ctx *c) { return c ? c->last_score_estimate : 0; }

extern "C" int qs_last_score_ms(qs_ctx *c, float out_ms[6]) {
    if (!c || !out_ms) return QS_ERR_ARG;
    for (int i = 0; i < 6; ++i) out_ms[i] = c->score_ms[i];
    return QS_OK;
}

static int raw_qic_impl(qs_ctx *c, const qs_ref_tree *ref, uint64_t r0, uint64_t nq, uint8_t *topo, uint64_t *q, bool lex) {
    if (!c || !c->table) return fail(c, QS_ERR_STATE, "qs_raw_qic: no table");
    if (r0 + nq > c->n_tuples) return fail(c, QS_ERR_ARG, "qs_raw_qic: rank range outside this context's table");
    if (lex && (c->d_lo != 0 || c->d_hi != c->n)) return fail(c, QS_ERR_UNSUPPORTED, "qs_raw_qic_lex: needs the whole table (not a shard)");
    if (nq == 0) return QS_OK;
    QS_HIP(c, hipSetDevice(c->device));
    const RefHost *Rp = nullptr;
    int rc = get_ref(c, ref, true, &Rp);
    if (rc != QS_OK) return rc;
    const RefHost &R = *Rp;
    uint8_t *dt = nullptr; unsigned long long *dq = nullptr;
    QS_HIP(c, hipMalloc(&dt, nq));
    hipError_t e = hipMalloc(&dq, nq * 24);
    if (e != hipSuccess) { (void)hipFree(dt); return fail(c, QS_ERR_OOM, "qs_raw_qic: hipMalloc"); }
    { int rc_t = ensure_score_tables(c); if (rc_t != QS_OK) return rc_t; }
    ScoreDevice sd;
    fill_score_device(c, R, c->ref_lca_dev, sd);
    sd.frame = 1; // printRawQICScores uses the multifurcating loop's argument order
    e = lex ? launch_raw_qic_lex(c->stream, sd, r0, nq, dt, dq) : launch_raw_qic(c->stream, sd, r0, nq, dt, dq);
    if (e == hipSuccess) e = hipMemcpyAsync(topo, dt, nq, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q, dq, nq * 24, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dt); (void)hipFree(dq);
    if (e != hipSuccess) return fail(c, QS_ERR_HIP, std::string("qs_raw_qic: ") + hipGetErrorString(e));
    return QS_OK;
}

extern "C" int qs_raw_qic(qs_ctx *c, const qs_ref_tree *ref, uint64_t r0, uint64_t nq, uint8_t *topo, uint64_t *q) {
    return raw_qic_impl(c, ref, r0, nq, topo, q, false);
}
extern "C" int qs_raw_qic_lex(qs_ctx *c, const qs_ref_tree *ref, uint64_t i0, uint64_t nq, uint8_t *topo, uint64_t *q) {
    return raw_qic_impl(c, ref, i0, nq, topo, q, true);
}
