// qs_internal.hpp -- launcher interface between the C-ABI layer (qs_abi.hip) and the kernels.
#pragma once
#include <vector>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qs {

// Panel element type / kernel mode of one batch
enum PanelBits { PANEL_U8 = 8, PANEL_U16 = 16 };
enum CountMode {
    MODE_BINARY_FULL = 0,  // every tree fully resolved and holding all n taxa: n2 = m - n0 - n1
    MODE_GENERAL_FULL = 1, // multifurcations possible, all taxa present
    MODE_PARTIAL = 2,      // taxa may be missing (flag bit in the panel), multifurcations possible
    MODE_BINARY_PARTIAL = 3 // every tree fully resolved, taxa may be missing (gene trees): n2 = (trees holding all four) - n0 - n1;
                            // bit-sliced kernel only (the byte-SWAR kernel takes such trees as MODE_PARTIAL)
};

// depth limits of the SWAR comparison (see qs_count.hip)
constexpr uint32_t kMaxDepthU8Full = 63;
constexpr uint32_t kMaxDepthU8Partial = 31;
constexpr uint32_t kMaxDepthU16Full = 16383;
constexpr uint32_t kMaxDepthU16Partial = 8191;

// device memory of one uploaded batch (all its arrays in one allocation); last_use: where the compute stream stood when
// the batch was freed (the slab is handed to a later upload, whose copy waits for it)
struct BatchSlab { void *p = nullptr; size_t cap = 0; hipEvent_t last_use = nullptr; };

// one workgroup of clamp_fix_kernel (qs_count.hip): the triples [t_lo, t_hi) (colex order) of the run of `run >> 16` leaves that
// starts at tour position `run & 0xFFFF` of tree `tree`
struct FixUnit { uint32_t tree, run, t_lo, t_hi; };

struct DeviceBatch {
    uint32_t n_trees = 0;
    uint32_t total_leaves = 0;
    uint32_t max_depth = 0;
    bool all_full = false;
    bool all_binary = false;
    // Trees are counted class by class. A class = (kernel mode of the tree -- binary / multifurcating x all taxa / missing
    // taxa --, bits of its deepest LCA), so that every tree runs the cheapest kernel instance that is exact for it: the
    // full binary trees of a batch keep the binary_full instance whatever else the batch holds. Slot s of the class-ordered
    // batch is tree tree_order[s]; class k holds the slots [class_end[k-1], class_end[k]), runs mode class_mode[k] and
    // needs class_bits[k] depth bits (> 10: byte-SWAR kernel). A sub-batch for the panel builders = slots
    // [slot0, slot0 + n_trees).
    uint32_t *tree_order = nullptr; // device, n_trees entries, or NULL = identity
    uint32_t slot0 = 0;
    uint32_t n_classes = 0;
    static constexpr uint32_t kMaxClasses = 32;   // 4 modes x 8 depth classes
    uint32_t class_bits[kMaxClasses] = {0}, class_end[kMaxClasses] = {0}, class_max_depth[kMaxClasses] = {0}, class_mode[kMaxClasses] = {0};
    uint32_t *leaf_off = nullptr;  // device
    uint16_t *leaf_ids = nullptr;  // device
    uint16_t *adj_depth = nullptr; // device
    // scatter-only
    uint32_t n_nodes = 0, n_links = 0;
    uint32_t *node_off = nullptr, *rng_off = nullptr, *node_tree = nullptr;
    uint16_t *ranges = nullptr;
    // depth clamp (qs_abi.hip plan_depth_clamp): the correction units of the trees counted in a class below their own depth bits,
    // ordered by the tree's slot in the class-ordered batch (qs_device_batch::fix_slot = that slot per unit, host)
    FixUnit *fix_units = nullptr;   // device
    uint32_t n_fix = 0;
    uint64_t fix_quartets = 0;      // (tree, quartet) corrections of the whole batch
    uint32_t clamped_trees = 0;
    BatchSlab slab;                 // owns the arrays above
    hipEvent_t ready = nullptr;     // fires when the batch's arrays have arrived (copy stream); the count waits for it
};

struct CountGeometry {
    uint32_t n;          // taxa
    uint32_t d_lo, d_hi; // shard of the largest id
    uint64_t rank_lo;    // C(d_lo,4)
    uint32_t n_dblk;     // d-blocks
    uint32_t total_tiles;
    const uint32_t *dprefix; // device, n_dblk+1
    const uint32_t *cprefix; // device, n+1
    const uint32_t *perm = nullptr; // device, total_tiles: launch slot -> tile id (nullptr = identity); see qs_abi.hip tile_order
    // binary_full tiling only: the launch order split for count_bitslice4_kernel. perm_coop: n_coop slots (a multiple of 4;
    // every aligned group of 4 = tiles of one (a-blocks, b-block, d-block), bit 31 = shadow tile that must not store);
    // perm_rest: the n_rest tiles count_bitslice3_kernel still runs. NULL / 0 = everything through count_bitslice3_kernel.
    const uint32_t *perm_coop = nullptr, *perm_rest = nullptr;
    uint32_t n_coop = 0, n_rest = 0;
};

// qs_count.hip
hipError_t launch_build_panel(hipStream_t s, const DeviceBatch &b, uint32_t n, int panel_bits, bool partial, void *panel,
                              uint32_t n_chunks);
hipError_t launch_count_gather(hipStream_t s, const CountGeometry &g, const void *panel, int panel_bits, int mode,
                               uint32_t n_chunks, uint32_t m_trees, void *table, int count_bits, uint32_t *overflow_flag,
                               bool overwrite);
hipError_t launch_build_bitpanel(hipStream_t s, const DeviceBatch &b, uint32_t n, bool partial, void *panel, uint32_t n_groups,
                                 uint32_t compact_nw, // words per panel element: planes (>= 4) [+ presence word]
                                 bool force_general);
hipError_t launch_count_bitslice3(hipStream_t s, const CountGeometry &g, const void *panel, int depth_bits, int mode,
                                  uint32_t n_groups, uint32_t m_trees, void *table, int count_bits, uint32_t *overflow_flag,
                                  bool overwrite, uint32_t *wire); // wire != NULL (binary_full only): one word n0 | n1 << 16 per tuple instead of the table
// the classes of a mixed batch that share their depth bits in ONE launch (qs_count_fused.hip); seg_* are indexed by CountMode, a mode
// without trees in this launch has seg_groups = 0
constexpr int kFusedMaxBits = 7;
hipError_t launch_count_bitslice3_fused(hipStream_t s, const CountGeometry &g, const void *const seg_panel[4], const uint32_t seg_groups[4],
                                        const uint32_t seg_trees[4], int depth_bits, void *table, int count_bits, uint32_t *overflow_flag,
                                        bool overwrite);
hipError_t launch_clamp_fix(hipStream_t s, const DeviceBatch &b, const FixUnit *units, uint32_t n_units, uint32_t d_lo, uint32_t d_hi,
                            uint64_t rank_lo, void *table, int count_bits, int mode, uint32_t *wire); // corrections of the depth clamp (after the class's count kernel)
uint32_t bitslice3_tiles_for_c(uint32_t c); // wave tiles per (d-block, c) of count_bitslice3_kernel
hipError_t launch_count_scatter(hipStream_t s, const DeviceBatch &b, uint32_t n, uint32_t d_lo, uint32_t d_hi,
                                uint64_t rank_lo, void *table, int count_bits);
hipError_t launch_pack16(hipStream_t s, const void *table_u32, void *dst, uint64_t n_cells, uint32_t *overflow_flag);
hipError_t launch_pack16x2(hipStream_t s, const void *table_u32, void *dst, uint64_t n_tuples, uint32_t trees,
                           uint32_t *overflow_flag, uint32_t *shape_flag);
hipError_t launch_unpack16x2(hipStream_t s, const void *src, void *dst_u16, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag);
hipError_t launch_pack32x2(hipStream_t s, const void *table_u32, void *dst, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag);
hipError_t launch_unpack32x2(hipStream_t s, const void *src, void *dst_u32, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag);
hipError_t launch_sum_words(hipStream_t s, void *dst, const void *const *src, uint32_t n_src, uint64_t n_words, int n_cu); // dst += sum of (peer) sources
hipError_t launch_lookup(hipStream_t s, uint32_t n, uint32_t d_lo, uint32_t d_hi, uint64_t rank_lo, const void *table,
                         int count_bits, uint64_t nq, const uint16_t *abcd_dev, uint64_t *out_dev);
size_t gather_lds_bytes(uint32_t d_hi);
uint32_t gather_tiles_for_c(uint32_t c); // workgroups of the gather kernel per (d-block, c)

// qs_score.hip
struct ScoreDevice {
    const uint32_t *ref_lca; // n*n: (depth << 16) | inner-node compact id, for leaf pairs (lookup ids)
    uint32_t n, n_inner;
    uint32_t d_lo, d_hi;
    uint64_t rank_lo, n_tuples;
    const void *table;
    int count_bits;
    unsigned long long *pair_sums; // n_inner*n_inner*3
    long long *pair_min;           // n_inner*n_inner (f64_to_sortable of the device QIC)
    unsigned long long *pair_cand; // n_inner*n_inner*kCand
    uint32_t cand_limit;           // candidate slots pass 2 may fill per node pair (kCand; smaller in tests)
    unsigned long long *list;      // pass 3 (qs_score_overflow): 4 words (key, q1, q2, q3) per near-minimal quartet of a marked pair;
                                   // pass 1 with list != NULL: the candidate log of the single-read scoring (same records)
    unsigned long long *list_count, list_cap;
    unsigned long long *last_trip; // logging pass 1: per node pair the packed (q1 << 42 | q2 << 21 | q3) of the record logged last for it (all
                                   // ones = none), or NULL: a quartet whose triple equals it is not logged again (ties at the bound)
    uint32_t *flags;               // [0] bit 0: a pair's candidate slots overflowed, bit 1: a reduced triple did not fit the packed slot
    const uint16_t *ref_next; // n*n: for a < b the first a' > a with lca(a',b) != lca(a,b), b if there is none
    const double *logk;            // log(k) (0 at k = 0) for k < tbl_n: integer arguments of the device QIC
    uint32_t tbl_n;
    uint32_t lds_n;                // leading entries of logk the scan kernel keeps in LDS
    const uint32_t *bundle_plo;    // bundle kernel (plan_bundles): per b the first pair index and the number of pairs (c,d) whose rows are scored
    const uint32_t *bundle_pcnt;
    const uint32_t *bundle_rounds; // n_rounds x (b / 16, pair group)
    uint32_t n_rounds;
    uint32_t coop_load;            // bundle kernel: 1 = a row's 96-byte chunk is loaded by eight lanes and handed over through LDS (QS_TUNE_SCORE_LOAD)
    uint32_t sample;               // pass 1, bundle kernel: 0 = every chunk; else the minima-only pre-pass of the single-read scoring:
                                   //    bits 0..15 = S (a power of two): one chunk (bit 16 clear) or one round (bit 16 set) in S
    uint32_t root_split;           // bifurcating reference with a degree-2 root: ids [0, root_split) lie under the root's first child (else 0);
                                   //    marks the quartets the reference evaluates in both (q2, q3) orders (qs_score.hip root_swapped)
    int frame;                     // 0: node-pair frame of processNodePair (QSC:417-431); 1: the (u,z|v,w) argument
                                   //    order of the multifurcating / raw-QIC loops (QSC:551-558, 661-668)
};
constexpr int kCand = 8;
constexpr unsigned long long kCandEmpty = ~0ull;
constexpr unsigned long long kCandOverflow = ~0ull - 1; // in the LAST slot of a node pair: its slots did not suffice (qs_score_overflow)
constexpr unsigned long long kCandSwap = 1ull << 63;    // candidate slot flag: the reference also evaluates log_score(q1, q3, q2) for it (degree-2 root)
constexpr unsigned long long kListSwap = 1ull << 32;    // the same flag in the key word of a (key, q1, q2, q3) record (candidate log, overflow lists)
uint32_t score_scan_max_lds_log(bool coop_load = false);
// kernel: 0 = bundle kernel (a wave walks 64 rows with the same b in lockstep; needs sd.bundle_* = plan_bundles of the
// rank range, and the partial rows at its ends, which go through the scan kernel), 1 = scan kernel (lane = 8 consecutive ranks)
struct BundlePlan { std::vector<uint32_t> plo, pcnt, rounds; uint64_t part_lo[2], part_n[2]; int n_parts; };
void plan_bundles(uint32_t n, uint64_t r0, uint64_t r1, uint32_t waves, BundlePlan &out);
uint32_t score_bundle_waves(int pass, bool coop_load = false);   // waves per workgroup (= consecutive b per round) of the bundle kernel in pass 1 / 2; cooperative loads: their own count
hipError_t launch_score_pass1(hipStream_t s, const ScoreDevice &sd, int kernel, int n_cu, const uint64_t *part_lo, const uint64_t *part_n, int n_parts, double tol = 0.0);
hipError_t launch_score_log(hipStream_t s, const ScoreDevice &sd, double tol, unsigned long long n_rec); // single-read scoring: filter pass 1's candidate log (sd.list)
hipError_t launch_score_pass2(hipStream_t s, const ScoreDevice &sd, double tol, int kernel, int n_cu, const uint64_t *part_lo, const uint64_t *part_n, int n_parts);
hipError_t launch_score_overflow_list(hipStream_t s, const ScoreDevice &sd, double tol);
// sums of the node pairs (degree-2 root, v): pairs_dev = RootPairHost records (qs_abi.hip), total = number of (v,a,b,c,d) items
struct RootPairHost { uint32_t s1_lo, s1_n, s2_lo, s2_n, s3_lo, s3_n, s4_lo, s4_n, key, pad; unsigned long long first; };
hipError_t launch_root_pair_sums(hipStream_t s, const ScoreDevice &sd, const void *pairs_dev, uint32_t n_pairs, uint64_t total);
hipError_t launch_raw_qic(hipStream_t s, const ScoreDevice &sd, uint64_t r0, uint64_t nq, uint8_t *topo_dev,
                          unsigned long long *q_dev);
hipError_t launch_raw_qic_lex(hipStream_t s, const ScoreDevice &sd, uint64_t i0, uint64_t nq, uint8_t *topo_dev,
                              unsigned long long *q_dev);

} // namespace qs
