/*
 * quartetscores_hip.h -- C-ABI of the MI355X (gfx950) quartet-support engine.
 *
 * This is the drop-in boundary for the one data-parallel hot path of
 * lutteropp/QuartetScores: quartet-topology counting into the
 * C(n,4) x 3 table and the LQ-/QP-/EQP-IC reduction. The reference has no FFI; the
 * seam is the two class templates its main() uses. Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference/src):
 *
 *   qs_create / qs_destroy        QuartetCounterLookup<CINT> ctor/dtor: table allocation
 *                                 (QuartetCounterLookup.hpp:245-273, quartet_lookup_table.hpp:59-63,135-139)
 *   qs_count_trees                QuartetCounterLookup::countQuartets / updateQuartets /
 *                                 updateQuartetsThreeLinks / updateQuartetsThreeClades
 *                                 (QuartetCounterLookup.hpp:65-238), on pre-flattened trees
 *   qs_lookup                     QuartetCounterLookup::countQuartetOccurrences
 *                                 (QuartetCounterLookup.hpp:299-318)
 *   qs_table_*                    QuartetLookupTable<T> storage (quartet_lookup_table.hpp:19-228)
 *   qs_score                      QuartetScoreComputer: processNodePair /
 *                                 computeQuartetScoresBifurcating / ...Multifurcating
 *                                 (QuartetScoreComputer.hpp:379-593) + getLQIC/QPIC/EQPICScores (:106-125)
 *   qs_raw_qic                    QuartetScoreComputer::printRawQICScores (:623-690), numeric part
 *
 * Conventions: plain pointers and sizes only; every function returns a status code
 * (QS_OK == 0) and never throws; qs_last_error() gives the message. The caller owns
 * all host buffers and lends them for the duration of a call; the library owns device
 * memory except a table attached with qs_table_attach(). One context drives one GPU
 * from one host thread. Multi-GPU = one context per GPU; the library never calls a
 * collective itself. Tree-sharded: every context counts its share of the trees into a
 * private table and the CALLER combines the tables with one RCCL reduce-scatter or
 * all-reduce -- csrc/host/multi_gpu.hpp (one process, RCCL's C API: `QuartetScores
 * --gpus N`) or quartetscores_amd/distributed.py (one process per GPU, torch.distributed)
 * -- then scores its part in steps (qs_score_set_view, qs_score_pass1/2, qs_score_finish).
 * Table-sharded: contexts created with [d_lo, d_hi) own disjoint rank ranges, every context
 * counts all trees, no table collective; only the per-node-pair accumulators of the score
 * passes are combined (`QuartetScores --gpus N --table-shards K`). See INTEGRATION.md.
 * There is NO CPU fallback: every entry point that computes needs a gfx950 device.
 *
 * Data model. Taxa have lookup ids 0..n-1 = the reference
 * tree's leaf order in a depth-first tour (QuartetCounterLookup.hpp:252-258). The
 * count table holds, for every 4-set s0<s1<s2<s3 at
 *     rank = C(s3,4) + C(s2,3) + C(s1,2) + s0          (quartet_lookup_table.hpp:161-165)
 * the tuple [ #s0s1|s2s3, #s0s2|s1s3, #s0s3|s1s2 ]     (quartet_lookup_table.hpp:87-111)
 * as `count_bits`-wide unsigned integers, array-of-tuples exactly like the reference's
 * std::vector<std::array<T,3>>. Counts are the SEMANTIC counts (number of evaluation
 * trees displaying the topology) = the reference's fast-table values = half of what its
 * --savemem table stores.
 */
#ifndef QUARTETSCORES_HIP_H
#define QUARTETSCORES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qs_ctx qs_ctx;

/* status codes */
enum {
    QS_OK = 0,
    QS_ERR_ARG = -1,        /* bad argument / malformed flattened tree */
    QS_ERR_HIP = -2,        /* HIP runtime error (message has the HIP string) */
    QS_ERR_OOM = -3,        /* "Insufficient memory!" (QuartetScoreComputer.hpp:735-737) */
    QS_ERR_STATE = -4,      /* call order (e.g. score before count) */
    QS_ERR_OVERFLOW = -5,   /* a counter would exceed count_bits, or candidate buffer overflow */
    QS_ERR_NO_DEVICE = -6,  /* no gfx950 device: the library does not fall back to the CPU */
    QS_ERR_UNSUPPORTED = -7,
    QS_ERR_REFERENCE_THROWS = -8 /* the reference itself ends with an uncaught std::runtime_error on this input; qs_last_error()
                                  * is that exception's what() (QS_SCORE_SAVEMEM_LOOKUPS) */
};

/* qs_create flags */
#define QS_FLAG_NONE 0u

/* counting algorithm selector for qs_count_trees */
#define QS_ALGO_AUTO 0u     /* = QS_ALGO_GATHER */
#define QS_ALGO_GATHER 1u   /* quartet-major: each lane owns table cells, trees streamed as pair-depth panels */
#define QS_ALGO_SCATTER 2u  /* tree-major: one wavefront per (tree, inner node, orientation), atomicAdd */

/* OR-ed into `algo`: the table's previous contents are discarded -- same result as qs_table_clear
 * followed by the count, without the extra clear and read passes over the table (gather only) */
#define QS_COUNT_OVERWRITE 0x100u
/* OR-ed into `algo`: bracket the kernels of this call with HIP events for qs_last_count_ms (costs ~5 % of a
 * 0.25 ms batch, so it is off unless asked for) */
#define QS_COUNT_TIMED 0x200u
/* OR-ed into `algo`: count into the wire buffer given to qs_wire_attach instead of the table -- one 32-bit word
 * n0 | n1 << 16 per tuple, the two-cell format of qs_table_pack16x2, without the table write and the pack pass.
 * Only for batches of binary trees that hold all taxa (anything else: QS_ERR_STATE), gather algorithm, fewer than
 * 65536 trees accumulated. QS_COUNT_OVERWRITE applies to the wire buffer. */
#define QS_COUNT_WIRE16X2 0x400u

/* qs_score flags */
#define QS_SCORE_QP_WRAP32 0u   /* reference-compatible: QP sums kept mod 2^32 (QuartetScoreComputer.hpp:382) */
#define QS_SCORE_QP_EXACT64 1u  /* 64-bit sums */
/* Reference tree with a degree-2 root (rooted Newick). Default: what the reference computes -- it takes the subtrees
 * beside a node pair with next() / next().next() on the link cycle, and on the root's two-link cycle that reaches back
 * to the subtree of v itself (QuartetScoreComputer.hpp:393-396): the pairs (root, v) get sums over
 * (other side) x (v's whole side) x (v's two child subtrees), which set the QP-IC of the two root edges and enter the
 * EQP-IC minima below them. With QS_SCORE_ROOT_AS_EDGE the root is treated as a
 * subdivision of one edge instead: both root edges carry the scores of the unrooted internode. */
#define QS_SCORE_ROOT_AS_EDGE 2u
/* The reference's memory-efficient table (`-s`, QuartetCounterLookup.hpp:303-311) behind the lookups of a degree-2 root's node
 * pairs. Those calls repeat an id (b == c or b == d); quartet_lookup_table.hpp:170-212 sorts the ids and :79-85 throws
 * std::runtime_error when the resulting index lies behind the table -- which happens for EVERY rooted reference tree with
 * four or more taxa (e.g. two ids equal to n-1 give C(n,4) + ...; pinned on the unmodified header, tests/test_oracle_reftable.py),
 * so the reference's run ends with that uncaught exception. With this flag qs_score / qs_score_finish report exactly that:
 * QS_ERR_REFERENCE_THROWS, qs_last_error() = the what() of the first throwing call in the reference's sequential order
 * ("id = <index>, but quartet_lookup_.size() = <C(n,4)>"). Without it (default) the pairs (root, v) are scored as the
 * reference's n^4 table scores them, where a repeated id reads a cell that is never incremented. No effect on unrooted
 * reference trees or with QS_SCORE_ROOT_AS_EDGE. */
#define QS_SCORE_SAVEMEM_LOOKUPS 4u

/*
 * A batch of evaluation trees, flattened by the host (see quartetscores_amd/csrc/host
 * or quartetscores_amd/flatten.py). For tree t, with L_t = leaf_off[t+1]-leaf_off[t]:
 *   leaf_ids [leaf_off[t] + i], i < L_t : lookup id of the i-th leaf in a depth-first tour
 *                                         == eulerTourLeaves (QuartetCounterLookup.hpp:211-221)
 *   adj_depth[leaf_off[t] + i], i < L_t-1: depth (edges from the tour's root) of the lowest
 *                                         common ancestor of leaves i and i+1; entry L_t-1 is 0.
 *                                         (leaf_ids, adj_depth) determine the tree.
 * Inner nodes (needed by QS_ALGO_SCATTER only; may be NULL otherwise):
 *   node_off[t] .. node_off[t+1]        : inner nodes of tree t
 *   rng_off[v] .. rng_off[v+1]          : the links of inner node v
 *   ranges[2*k], ranges[2*k+1]          : circular half-open [start,end) leaf positions behind
 *                                         link k == subtreeLeafIndices (QuartetCounterLookup.hpp:117-121)
 * A leaf label unknown to the reference is a host-side error (QuartetCounterLookup.hpp:218);
 * taxa missing from a tree simply do not appear in leaf_ids.
 */
typedef struct {
    uint32_t n_trees;
    const uint32_t *leaf_off;  /* n_trees + 1 */
    const uint16_t *leaf_ids;
    const uint16_t *adj_depth;
    const uint32_t *node_off;  /* n_trees + 1, or NULL */
    const uint32_t *rng_off;   /* node_off[n_trees] + 1, or NULL */
    const uint16_t *ranges;    /* 2 * rng_off[last], or NULL */
} qs_tree_batch;

/*
 * The reference tree, flattened. Nodes 0..n_nodes-1, parent[root] = -1. leaf_node[i] is
 * the node of the taxon with lookup id i; lookup ids must be in depth-first order (every
 * node's leaves form one contiguous id interval), which qs_score verifies. A degree-2
 * root is scored like the reference does unless QS_SCORE_ROOT_AS_EDGE
 * is given.
 */
typedef struct {
    uint32_t n_nodes;
    uint32_t n_taxa;
    const int32_t *parent;     /* n_nodes */
    const uint32_t *leaf_node; /* n_taxa */
} qs_ref_tree;

/* ---- lifecycle ----------------------------------------------------------------------- */

/*
 * n_taxa in [4, 4096]; count_bits 16 or 32 (the reference picks u8/u16/u32/u64 from m,
 * QuartetScores.cpp:115-147; counts compare as integers). device = HIP
 * device ordinal. stream = hipStream_t to launch on (NULL = the default stream).
 * The table is not allocated until qs_table_alloc / qs_table_attach.
 * d_lo/d_hi shard the table by the LARGEST taxon id of the 4-set: this context owns the
 * ranks [C(d_lo,4), C(d_hi,4)); pass 0, n_taxa for the whole table.
 */
int qs_create(qs_ctx **out, uint32_t n_taxa, uint32_t count_bits, uint32_t flags, int device, void *stream,
              uint32_t d_lo, uint32_t d_hi);
void qs_destroy(qs_ctx *ctx);
const char *qs_last_error(const qs_ctx *ctx); /* ctx may be NULL: message of the last failed qs_create */

/* Tuning knobs of one context (A/B measurements and tests; the defaults are what the product uses). They replace
 * the reference's compile-time switches of this path (USE_STXXL, quartet_lookup_table.hpp:3; table mode by RAM,
 * QuartetScoreComputer.hpp:739). */
#define QS_TUNE_PANEL_SLICE_BYTES 1u /* upper bound of the pair-depth panel of one sub-batch; 0 = automatic */
#define QS_TUNE_GATHER_IMPL 2u       /* QS_IMPL_AUTO | QS_IMPL_SWAR (byte-SWAR kernel) | QS_IMPL_BITSLICE */
#define QS_TUNE_PANEL_KERNEL 3u      /* 0 = automatic, 1 = always the general bit-plane panel builder */
#define QS_TUNE_TILE_ORDER 4u        /* launch order of the count kernel's tiles: 0 = (d,c)-major (the order of the table);
                                      * chunk | cblock << 16 = (a,b)-major: b-block, chunks of `chunk` a-blocks, blocks of
                                      * `cblock` values of c, d-blocks (default 2 | 32 << 16) */
#define QS_TUNE_SCORE_CAND_SLOTS 5u   /* candidate slots score pass 2 fills per node pair, 1..8 (default 8; tests force overflows) */
#define QS_TUNE_SCORE_KERNEL 7u       /* score passes 1 and 2: 0 = bundle kernel (default; a wave walks 64 table rows with the same second id in lockstep), 1 = scan kernel (a lane walks 8 consecutive ranks; A/B and tests) */
#define QS_TUNE_SCORE_TOL_EXP 6u      /* pass 2 keeps count triples whose device QIC is within 10^-value of the pair's minimum (default 12) */
#define QS_TUNE_TABLE_TREES 8u        /* number of trees behind a table this context did not count itself (reduced over GPUs, uploaded,
                                       * attached or viewed): sizes the log table of the device QIC so that every count takes the
                                       * table path (speed only; scores never depend on it). 0 = what the context counted (default) */
#define QS_TUNE_SCORE_PASSES 10u      /* qs_score: 0 = automatic (default), 1 = two passes over the table, 2 = single read. Single read: a
                                       * minima-only pre-pass over a sample of the table, then ONE pass that adds the sums, lowers the
                                       * minima and logs every quartet within the tolerance of the bound it knows (a superset of pass 2's
                                       * candidates); a filter over the log replaces pass 2; pass 2 still runs when the log overflows.
                                       * Automatic: single read for tables from 1 GB unless a second sample predicts that the log would
                                       * not hold (tie-heavy tables) -- 14.1 instead of 21.8 ms at 512 taxa */
#define QS_TUNE_SCORE_LOG_CAP 11u     /* records the candidate log of the single-read scoring may hold (0 = 8 M = 256 MB); tests force overflows */
#define QS_TUNE_SCORE_SAMPLE 12u      /* single-read scoring: the pre-pass takes one round in S (value = S | 65536; default S = 64) or one 96-byte
                                       * chunk of every row in S (value = S); S a power of two; 0 = no pre-pass (and no automatic mode) */
#define QS_TUNE_CLASS_PCT 15u         /* batches are counted class by class (kernel mode of a tree x its depth bits); a class holding less than this
                                       * share of the trees, or fewer than QS_TUNE_CLASS_MIN_TREES, joins a more general mode / the next deeper
                                       * class: every class costs a table pass (default 10) */
#define QS_TUNE_DEPTH_CLAMP 17u       /* depth clamp: a tree may be counted in a class BELOW its own depth bits (its LCA depths cut at the class's
                                       * largest value; the quartets the cut ties -- three leaves below one node of that depth -- are added by a
                                       * correction kernel with atomics) when that costs at most `value` millionths of C(n,4) (tree, quartet)
                                       * corrections per depth bit saved (default 20; 16 x that for the trees of a class too small for a pass of
                                       * its own); 0 = every tree in the class of its own depth bits. Read by qs_batch_upload. */
#define QS_TUNE_FUSE_CLASSES 18u      /* 1 (default): the classes of a batch that share their depth bits (up to 7) are counted in ONE launch of the count
                                       * kernel -- one pass over the table whatever the mix of tree shapes, like the reference's shape-independent loop
                                       * (QuartetCounterLookup.hpp:65-106,166-188); no mode joins a dearer one any more. 0: one launch per class (round 5).
                                       * Read by qs_batch_upload (class plan) and qs_count_batch (launches). */
#define QS_TUNE_CLASS_MIN_TREES 16u   /* ... and the absolute floor of a class (default 1024 trees; tests lower it to split small batches) */
#define QS_TUNE_SCORE_LOAD 14u        /* bundle score kernel, shape of the table loads: 0 (default) = every lane loads its own row in 16-byte pieces;
                                       * 2 = ... and requests the next chunk before it processes the current one; 1 = eight lanes load the
                                       * 96-byte chunk of a row together and hand it over through LDS; 3 = 1 with the next chunk requested
                                       * ahead (1, 3: workgroups of 8 waves). All exact; 1-3 measured slower on MI355X (profiles/r05_experiments.md 6): A/B switches */
#define QS_TUNE_SCORE_DEDUPE 13u      /* single-read scoring: 1 (default) = a quartet that repeats the triple logged last for its node pair is
                                       * not logged again (tables of similar trees put thousands of equal triples at a pair's bound); 0 = off */
#define QS_TUNE_COOP 9u               /* binary full batches: 1 = run the tiles with two a-blocks through count_bitslice4_kernel, whose
                                       * workgroups (four consecutive third ids of one (a,b,d) tile) share their panel loads through
                                       * LDS; 0 / 2 = off (default: the barrier it needs costs more than the loads it saves) */
#define QS_IMPL_AUTO 0u
#define QS_IMPL_SWAR 1u
#define QS_IMPL_BITSLICE 2u
int qs_set_tuning(qs_ctx *ctx, uint32_t key, uint64_t value);
/* Optional: do ahead of time what the first qs_count_batch does before its first launch -- the launch order of the
 * count kernel's tiles for binary batches and the pair-depth panel for a batch of n_trees_hint trees (0 = no panel) --
 * e.g. on the host's GPU-initialisation thread while the evaluation trees are still being parsed. The reference does
 * its set-up in the table's constructor (QuartetCounterLookup.hpp:245-273). Never required. */
int qs_prepare(qs_ctx *ctx, uint64_t n_trees_hint);
const char *qs_version(void);

/* ---- count table (QuartetLookupTable) ------------------------------------------------- */

uint64_t qs_table_tuples(const qs_ctx *ctx); /* number of 4-sets owned = C(d_hi,4)-C(d_lo,4) */
uint64_t qs_table_bytes(const qs_ctx *ctx);  /* tuples * 3 * count_bits/8 (compare quartet_lookup_table.hpp:69-71) */
int qs_table_alloc(qs_ctx *ctx);             /* hipMalloc + zero; QS_ERR_OOM if it does not fit */
int qs_table_attach(qs_ctx *ctx, void *device_ptr, uint64_t bytes); /* caller-owned device memory, e.g. a torch tensor: 4-byte
                                              * aligned, at least qs_table_bytes() rounded up to a multiple of 4.
                                              * (NULL, 0) detaches a caller-owned table after waiting for the context's
                                              * stream: the caller may free it, e.g. once a reduce-scattered shard is
                                              * the scoring view (qs_score_set_view) */
void *qs_table_device_ptr(const qs_ctx *ctx);
int qs_table_clear(qs_ctx *ctx);
int qs_table_download(qs_ctx *ctx, void *host_dst, uint64_t bytes);
int qs_table_upload(qs_ctx *ctx, const void *host_src, uint64_t bytes);

/* dst[i] += src[0][i] + ... + src[n_src-1][i] for i < n_words (32-bit words; u16 cells and the two-cell wire words add as
 * packed words while the totals stay in range). The sources may be memory of PEER devices that the caller has made
 * accessible (hipDeviceEnablePeerAccess): the reduce(-scatter) of tree-sharded tables inside ONE process without a
 * communicator -- GPU g sums chunk g of every peer's table with plain loads over xGMI (the C++ host's `--reduce p2p`;
 * the reference has no cross-process reduction). 16-byte aligned pointers, at most 15 sources.
 * Asynchronous on the context's stream; the caller orders it after the peers' counting (events / synchronisation). */
int qs_sum_words(qs_ctx *ctx, void *dst_device, const void *const *src_device, uint32_t n_src, uint64_t n_words);

/* Writes the u32 table as a u16 table (same [rank][3] layout, 2 bytes per cell, padded to a whole 32-bit word)
 * into caller-owned device memory: the wire format for the multi-GPU all-reduce while all totals stay below 2^16
 * (packed words add without carry between the halves), and itself a valid count_bits = 16 table that a second
 * context can attach and score. A cell >= 65536 raises QS_ERR_OVERFLOW at the next qs_sync. Asynchronous. */
int qs_table_pack16(qs_ctx *ctx, void *dst_device, uint64_t dst_bytes);

/* Two-cell wire format for batches in which every tree resolves every quartet (binary evaluation trees holding all
 * taxa: n0 + n1 + n2 = number of trees): a tuple travels as ONE 32-bit word n0 | n1 << 16 (a third less than
 * qs_table_pack16). qs_table_pack16x2 writes table_tuples words; after the collective qs_unpack16x2 turns n_tuples
 * reduced words back into a count_bits = 16 table ([tuple][3] u16, n2 = total_trees - n0 - n1) in caller-owned
 * memory. A tuple that does not sum to the number of trees raises QS_ERR_STATE at the next qs_sync (use the
 * three-cell format then), a count >= 65536 QS_ERR_OVERFLOW. Asynchronous. */
int qs_table_pack16x2(qs_ctx *ctx, void *dst_device, uint64_t dst_bytes);
/* destination of QS_COUNT_WIRE16X2 (table_tuples 32-bit words in caller-owned device memory); NULL detaches */
int qs_wire_attach(qs_ctx *ctx, void *dst_device, uint64_t dst_bytes);
int qs_unpack16x2(qs_ctx *ctx, const void *src_device, uint64_t n_tuples, uint32_t total_trees, void *dst_device);
/* The same two-cell format with 32-bit cells, for totals of 65536 trees and more (BASELINE configs[3]: 100 000 trees over
 * 8 GPUs): qs_table_pack32x2 writes (n0, n1) per tuple = 8 bytes instead of 12 on the wire, qs_unpack32x2 restores
 * [rank][3] u32 tuples with n2 = total_trees - n0 - n1. A tuple that does not sum to the number of trees (the batch was
 * not binary with all taxa) raises the flag qs_sync reports as QS_ERR_STATE. */
int qs_table_pack32x2(qs_ctx *ctx, void *dst_device, uint64_t dst_bytes);
int qs_unpack32x2(qs_ctx *ctx, const void *src_device, uint64_t n_tuples, uint64_t total_trees, void *dst_device);

/* ---- counting (QuartetCounterLookup::countQuartets) ------------------------------------ */

/* Opaque device-resident copy of a batch. */
typedef struct qs_device_batch qs_device_batch;

/* Validates the batch and copies it into HBM: on return the host buffers are free again (the arrays sit in pinned staging
 * memory of the context; two buffers, so that the next batch can be staged meanwhile), the copy to the device runs on a
 * copy stream of the library and qs_count_batch orders itself behind it. qs_batch_free never waits for kernels: the
 * device memory is kept for a later upload, whose copy is ordered behind the kernels that still read it. */
int qs_batch_upload(qs_ctx *ctx, const qs_tree_batch *batch, qs_device_batch **out);
void qs_batch_free(qs_ctx *ctx, qs_device_batch *b);
/* What the validation found: QS_BATCH_ALL_TAXA = every tree holds all n taxa, QS_BATCH_BINARY = every tree is fully
 * resolved. Both set = the batch may be counted with QS_COUNT_WIRE16X2 / travel in the two-cell wire format. */
#define QS_BATCH_ALL_TAXA 1u
#define QS_BATCH_BINARY 2u
uint32_t qs_batch_flags(const qs_device_batch *b);

/* Adds the quartet topologies of every tree of the batch to the table. Asynchronous on the
 * context's stream; inputs already resident in HBM. */
int qs_count_batch(qs_ctx *ctx, const qs_device_batch *b, uint32_t algo);

/* Convenience = qs_batch_upload + qs_count_batch + qs_sync + qs_batch_free. */
int qs_count_trees(qs_ctx *ctx, const qs_tree_batch *batch, uint32_t algo);

int qs_sync(qs_ctx *ctx); /* hipStreamSynchronize + deferred error check */

/* Total evaluation trees counted so far (the m of the reference). */
uint64_t qs_trees_counted(const qs_ctx *ctx);

/* countQuartetOccurrences for nq quartets: abcd[4*i..] are lookup ids (any order, distinct),
 * out3[3*i..] = (#ab|cd, #ac|bd, #ad|bc). Quartets outside this context's shard give 0,0,0. */
int qs_lookup(qs_ctx *ctx, uint64_t nq, const uint16_t *abcd, uint64_t *out3);

/* ---- scoring (QuartetScoreComputer) ----------------------------------------------------- */

/*
 * LQ-IC, QP-IC, EQP-IC per edge. Output arrays have n_nodes entries, indexed by the CHILD
 * node of each edge (entry of the root unused); untouched edges = +inf like the reference
 * (QuartetScoreComputer.hpp:762-774). For a multifurcating reference only lqic is written
 * (qpic/eqpic may be NULL) and *is_bifurcating = 0 (QuartetScoreComputer.hpp:760-765).
 * The O(C(n,4)) enumeration, sums and minima run on the GPU; the final O(#node pairs)
 * log_score evaluations use the host's libm so scores are bit-identical to the
 * reference's CPU arithmetic (QuartetScoreComputer.hpp:135-159).
 */
int qs_score(qs_ctx *ctx, const qs_ref_tree *ref, uint32_t flags, double *lqic, double *qpic, double *eqpic,
             int *is_bifurcating);
/* Optional: what a first qs_score / qs_score_pass1 does before its kernels (reference tree and LCA matrix on the device, log
 * table for n_trees_total trees, round tables, accumulators with their pinned host copy, the candidate log), ahead of time.
 * May be called while count kernels of this context are still in flight (from the context's host thread): it uses the
 * copy stream and does not wait for the count stream -- with one exception: if the context already caches a DIFFERENT
 * reference tree, replacing it waits for the count stream first (kernels may still read the old one). The copy stream is
 * drained on every exit path, also on errors. The CLI calls it behind its last qs_count_batch. */
int qs_score_prepare(qs_ctx *ctx, const qs_ref_tree *ref, uint64_t n_trees_total);

/*
 * The same computation in steps, for table-sharded contexts (qs_create with a [d_lo,d_hi) shard) and
 * multi-GPU runs. All buffers are CALLER-owned (e.g. torch tensors) so that the caller can reduce them
 * over the shards between the steps; element type int64 throughout.
 *   P = qs_score_pair_slots(ref)  (= (#inner nodes)^2; 0 on a malformed tree)
 *   qs_score_pass1 : sums_dev[3*P] = per node pair, 64-bit sums of (q1,q2,q3) over THIS context's quartets;
 *                    min_dev[P]   = the smallest device-evaluated QIC, order-preserving int64 encoding
 *                    -> reduce over shards: SUM on sums_dev, MIN on min_dev (plain int64 reductions)
 *                    For tables from 1 GB (QS_TUNE_SCORE_PASSES) this pass also LOGS, in the context, every quartet within
 *                    the tolerance of the bound it knows for its node pair -- a superset of what pass 2 can ask for as long
 *                    as min_dev there is <= this context's own minima (it is: the MIN over the shards) -- so that
 *                    the pass 2 that follows can filter the log instead of reading the table again. Asynchronous on the
 *                    context's stream -- except in that automatic mode, where the call waits once on the host for a
 *                    counter (the sample that predicts the log's size: ~0.3 ms of kernels) before it enqueues the pass.
 *   qs_score_pass2 : cand_dev[QS_SCORE_CAND_SLOTS*P] = distinct gcd-reduced count triples of this context
 *                    whose QIC is within 1e-12 of min_dev (-1 = empty slot; bit 63 of a slot: the reference evaluates
 *                    this triple a second time with q2 and q3 exchanged -- degree-2 root, DESIGN.md 1 Q5 -- and
 *                    qs_score_finish takes both) -> all-gather over shards.
 *                    Filters the log of the qs_score_pass1 that preceded it on the same context, table / view and
 *                    reference (the table is then read ONCE); reads the table again if there is no such log (any
 *                    qs_count_* / qs_table_* / qs_set_tuning / qs_score_set_view call in between discards it, and the
 *                    caller must not write into an attached table between the two passes) or if the log overflowed.
 *                    A node pair with more than 8 such triples (or one whose reduced counts need more than 21
 *                    bits) is MARKED in cand_dev instead of failing the run:
 *   qs_score_overflow: lists every near-minimal quartet (key, q1, q2, q3; 4 int64 per entry, sorted, distinct; bit 32 of the
 *                    key word = the same "both orders" flag) of
 *                    this context's marked pairs into a malloc'ed host array (*list_out, free with qs_free_host;
 *                    NULL / 0 when nothing was marked -- the usual case) -> concatenate over the shards
 *   qs_score_finish: pure host arithmetic on the reduced sums, the n_cand_parts gathered candidate arrays (each
 *                    QS_SCORE_CAND_SLOTS*P long, concatenated) and the n_extra listed quartets (may be NULL / 0)
 *                    -> the three score vectors. ctx may be NULL (no device is touched).
 */
#define QS_SCORE_CAND_SLOTS 8
uint64_t qs_score_pair_slots(const qs_ref_tree *ref);
/* Host-only (no device call; ctx may be NULL, the message then comes from qs_last_error(NULL)): the status qs_score /
 * qs_score_finish would return for this reference tree and these flags for reasons that depend on the TREE alone -- today
 * QS_ERR_REFERENCE_THROWS for QS_SCORE_SAVEMEM_LOOKUPS with a rooted reference tree (same message). Lets a host know before it
 * counts; the reference itself counts first and dies in its scoring loop (QuartetScoreComputer.hpp:393-431). */
int qs_score_check(qs_ctx *ctx, const qs_ref_tree *ref, uint32_t flags);

/* Scoring view: qs_score_pass1 / qs_score_pass2 read the tuples of ranks [rank_lo, rank_lo + n_tuples) from
 * caller-owned device memory ([tuple][3] cells of count_bits bits) instead of the context's own table -- the
 * shard a rank holds after a reduce-scatter of the count table. table_dev = NULL returns to the own table.
 * qs_score / qs_lookup / qs_raw_qic keep using the context's own table. */
int qs_score_set_view(qs_ctx *ctx, const void *table_dev, uint32_t count_bits, uint64_t rank_lo, uint64_t n_tuples);
int qs_score_pass1(qs_ctx *ctx, const qs_ref_tree *ref, int64_t *sums_dev, int64_t *min_dev);
int qs_score_pass2(qs_ctx *ctx, const qs_ref_tree *ref, const int64_t *min_dev, int64_t *cand_dev);
int qs_score_overflow(qs_ctx *ctx, const qs_ref_tree *ref, const int64_t *min_dev, const int64_t *cand_dev,
                      int64_t **list_out, uint64_t *n_out);
void qs_free_host(void *p);
int qs_score_finish(qs_ctx *ctx, const qs_ref_tree *ref, uint32_t flags, const int64_t *sums_host,
                    const int64_t *cand_host, uint32_t n_cand_parts, const int64_t *extra_host, uint64_t n_extra,
                    double *lqic, double *qpic, double *eqpic, int *is_bifurcating);

/*
 * Raw per-quartet QIC (numeric part of printRawQICScores): for ranks [r0, r0+nq) of this
 * context's shard, topo[i] = 0 (s0s1|s2s3), 2 (s0s3|s1s2) or 255 (unresolved in the
 * reference tree: skipped by the reference, QuartetScoreComputer.hpp:669-672) and
 * q[3*i..] = (q1,q2,q3) in the reference's argument order for log_score.
 */
int qs_raw_qic(qs_ctx *ctx, const qs_ref_tree *ref, uint64_t r0, uint64_t nq, uint8_t *topo, uint64_t *q);
/* The same for the quartets number [i0, i0+nq) in LEXICOGRAPHIC order of their sorted lookup ids (a < b < c < d with a
 * outermost, d innermost) -- the line order of the reference's -q file, which walks its Euler-tour leaf order with four
 * nested loops (QuartetScoreComputer.hpp:626-630). Whole-table contexts only. */
int qs_raw_qic_lex(qs_ctx *ctx, const qs_ref_tree *ref, uint64_t i0, uint64_t nq, uint8_t *topo, uint64_t *q);

/* ---- measurement hooks ------------------------------------------------------------------ */

/* Device time in milliseconds of the most recent qs_count_batch that carried QS_COUNT_TIMED, split by kernel
 * (HIP events on the context's stream after every launch): [0] pair-panel builds, [1] count kernels (summed over
 * the panel slices of the batch), [2] whole call. QS_ERR_STATE if the most recent call was not timed.
 * qs_last_count_launches: how many count-kernel launches [1] covers (0 if the call was not timed). */
int qs_last_count_ms(qs_ctx *ctx, float out_ms[3]);
int qs_last_count_launches(const qs_ctx *ctx);
/* ... and the share of [1] spent in the depth-clamp correction kernels (QS_TUNE_DEPTH_CLAMP; 0 without clamped trees). */
float qs_last_count_fix_ms(qs_ctx *ctx);
/* ... and every kernel of that call in launch order: ms[k] = its duration, kind[k] = 0 panel build, 1 count kernel, 2 depth-clamp
 * corrections; returns the number of kernels written (at most cap). */
int qs_last_count_events(qs_ctx *ctx, float *ms, uint8_t *kind, int cap);
/* Depth clamp of an uploaded batch: out[0] = trees counted in a class below their own depth bits, out[1] = the (tree, quartet)
 * corrections they cost, out[2] = workgroups of the correction kernel. */
int qs_batch_clamp_info(const qs_device_batch *batch, uint64_t out[3]);
/* Host-only (runs without a GPU): the per-tree class plan qs_batch_upload applies for n_taxa taxa and the budget `ppm`
 * (QS_TUNE_DEPTH_CLAMP): own_bits[t] = depth bits of tree t's deepest LCA (4 .. 10, 11 = deeper), class_bits[t] = depth bits of the
 * cheapest class the budget allows it, corrections[t] = (tree, quartet) corrections of that choice. Any output may be NULL.
 * The reference's loop is shape-independent (QuartetCounterLookup.hpp:65-106): this replaces nothing there. */
int qs_depth_clamp_plan(uint32_t n_taxa, const qs_tree_batch *batch, uint32_t ppm, uint8_t *own_bits, uint8_t *class_bits,
                        uint64_t *corrections);
/* Host-only: the classes qs_batch_upload forms for these trees with the floors QS_TUNE_CLASS_MIN_TREES / QS_TUNE_CLASS_PCT and the clamp
 * budget given: per tree the kernel mode (0 binary_full, 1 general_full, 2 partial, 3 binary_partial) and depth bits of the class it is
 * counted in (a small class joins a more general mode, goes down to a larger class where the corrections allow it, or joins the next
 * deeper class), and the tree's slot in the class-ordered batch. Any output may be NULL. The trees are not validated here. */
#define QS_CLASS_PLAN_FUSED 0x100u   /* or-ed into class_pct: the plan of QS_TUNE_FUSE_CLASSES = 1 (classes of equal depth bits share a launch) */
int qs_class_plan(uint32_t n_taxa, const qs_tree_batch *batch, uint32_t class_min_trees, uint32_t class_pct, uint32_t clamp_ppm,
                  uint8_t *mode_of_tree, uint8_t *bits_of_tree, uint32_t *slot_of_tree);
/* Phases of the most recent qs_score call in milliseconds: [0] the whole call (host clock), [1] set-up (accumulator
 * allocation, reference tree + LCA matrix, log table: near zero once cached in the context), [2] pass 1 and [3] pass 2
 * (HIP events on the context's stream; [3] = the filter over pass 1's candidate log in single-read mode), [4] host wait for the passes incl. the overflow pass and the accumulators' way
 * back, [5] qs_score_finish (host libm + min-propagation; QuartetScoreComputer.hpp:448-454,484-489). */
int qs_last_score_ms(qs_ctx *ctx, float out_ms[6]);
/* Records in the candidate log of the most recent qs_score when it read the table once; 0 = it read the table twice (two
 * passes asked for, a table below 1 GB, a predicted or an actual overflow of the log). */
uint64_t qs_last_score_log(const qs_ctx *ctx);
/* Automatic mode: the log size the most recent qs_score predicted from its sample (hits of the sample x S); 0 = no estimate ran. */
uint64_t qs_last_score_estimate(const qs_ctx *ctx);
/* Diagnostic: nanoseconds per wave instruction and SIMD of the count kernel's bare instruction slot (24 v_bitop3 + 4 v_bcnt,
 * operands in registers, 4 waves per SIMD on every CU) on THIS device -- devices of one pool hold different clocks under a
 * VALU-dense load, and the figure makes measurements from different boxes comparable. `iterations` loop trips of 4 slots
 * each (100000 = ~60 ms); synchronous. */
int qs_issue_probe(qs_ctx *ctx, uint32_t iterations, float *ns_per_instruction);
/* Name of the kernel variant the last qs_count_batch dispatched (for logs/profiles). */
const char *qs_last_count_variant(const qs_ctx *ctx);
/* How score passes 1 and 2 decompose the tuples [rank_lo, rank_lo + n_tuples) of an n_taxa table (host arithmetic only,
 * no device, no context; the reference walks node pairs instead, QuartetScoreComputer.hpp:495-508). A table ROW = the
 * b consecutive tuples (a = 0..b-1) of one (b,c,d). For every second id b the rows that lie completely inside the
 * range are those of the pairs (c,d), b < c < d, number [first_pair[b], first_pair[b] + n_pairs[b]) in the order
 * C(d-b-1,2) + (c-b-1); the bundle kernel walks 64 of them per wave. parts[0..3] = (first rank, tuples) of the at most
 * two partial rows at the ends of the range (0,0 if absent), which the scan kernel walks. first_pair / n_pairs: n_taxa
 * entries each. */
int qs_score_plan(uint32_t n_taxa, uint64_t rank_lo, uint64_t n_tuples, uint32_t *first_pair, uint32_t *n_pairs, uint64_t parts[4]);
/* Host-only: where to cut the table into n_shards contiguous shards by the largest taxon id (table-sharded counting on N GPUs or
 * through one GPU: `QuartetScores --table-shards K`, bench.py --mode table). bounds[0..n_shards]: shard k owns the quartets whose largest
 * id lies in [bounds[k], bounds[k+1]) -- contiguous rank ranges, because the rank's leading term is C(s3,4)
 * (/root/reference/src/quartet_lookup_table.hpp:161-165; the reference itself never cuts its table). by = QS_SHARDS_BY_TUPLES balances
 * the tuples a shard holds (memory), QS_SHARDS_BY_COST what the count kernel spends on it (tiles of its d-blocks; the cut with the
 * smallest largest shard). Shards may be empty for tiny n. */
#define QS_SHARDS_BY_TUPLES 0u
#define QS_SHARDS_BY_COST 1u
int qs_shard_bounds(uint32_t n_taxa, uint32_t n_shards, uint32_t by, uint32_t *bounds);

#ifdef __cplusplus
}
#endif
#endif /* QUARTETSCORES_HIP_H */
