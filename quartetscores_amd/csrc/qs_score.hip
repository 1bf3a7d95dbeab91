// qs_score.hip -- LQ-/QP-/EQP-IC reductions over the count table on gfx950.
//
// Replaces the C(n,4)-sized part of QuartetScoreComputer (QuartetScoreComputer.hpp):
//   processNodePair / computeQuartetScoresBifurcating   :379-508
//   computeQuartetScoresMultifurcating                  :513-593
//   the topology test of printRawQICScores              :636-672
//
// The reference walks node pairs (u,v) and enumerates S1xS2xS3xS4; every 4-set is visited
// exactly once overall. Here the walk is quartet-major, in the order of the table: the bundle
// kernel (passes 1 and 2) gives a wave 64 table rows with the same second id and walks them in
// lockstep, the scan kernel (pass 3, partial rows, A/B) gives a lane 8 consecutive ranks; the
// table is read once per pass. Lookup ids are the reference tree's own
// depth-first leaf order (QuartetCounterLookup.hpp:252-258), therefore for sorted ids
// a<b<c<d only the two non-crossing pairings ab|cd and ad|bc can be the reference topology,
// decided from the LCA depths of the three adjacent pairs (equivalent to the reference's
// "strictly largest LCA-to-LCA distance" test, :535-562, because
// dist(lca_xy, lca_zw) = depth(lca_xy) + depth(lca_zw) - 2*depth(lca of all four)).
// The same three LCAs give the two junction nodes (u,v) of the quartet = the node pair that
// owns it (:436-447). In the reference's frame (a in S1, b in S2, c in S3, d in S4 with
// S1,S2,S3,S4 consecutive in the cyclic leaf order) p2 always receives the crossing pairing.
//
// Pass 1: per node pair, 64-bit sums of (q1,q2,q3) and the minimum device-evaluated QIC.
// Pass 2: every distinct count triple whose device QIC is within `tol` of the pair's minimum
//         is recorded (scaled by its gcd, which leaves log_score bit-identical), so that the
//         host can evaluate log_score with the same libm as the reference's CPU path and take
//         the exact minimum. The host work is O(#node pairs), the device work O(C(n,4)).
#include "qs_common.hpp"
#include "qs_internal.hpp"

#include <algorithm>

namespace qs {

// Ranks are walked in blocks of consecutive values: the block's first rank is un-ranked once (f64 sqrt / cbrt,
// ~300 instructions), every other rank of the block from it: rank = C(d,4) + C(c,3) + (C(b,2) + a), so adding `off`
// to the pair rank and carrying into c (and d) is enough; unrank2 is a float sqrt and two corrections.
struct Ids4 { uint32_t a, b, c, d; };
__device__ __forceinline__ Ids4 decode_near(const Ids4 &base, uint32_t off) {
    Ids4 r;
    uint32_t c = base.c, d = base.d;
    uint64_t pr = binom2(base.b) + base.a + off;
    for (;;) {
        const uint64_t lim = binom2(c);
        if (pr < lim) break;
        pr -= lim;
        if (++c == d) { ++d; c = 2; }
    }
    unrank2((uint32_t)pr, r.a, r.b);
    r.c = c; r.d = d;
    return r;
}

struct QuartetRef {
    bool resolved;
    uint32_t key;        // lo_inner * n_inner + hi_inner
    uint32_t q1, q2, q3; // counts in the reference's log_score argument order
    uint8_t topo;        // 0: s0s1|s2s3, 2: s0s3|s1s2, 255: unresolved
};

template <typename CT>
__device__ __forceinline__ QuartetRef classify(const ScoreDevice &sd, uint64_t local_rank, const Ids4 &ids) {
    QuartetRef r;
    const uint32_t a = ids.a, b = ids.b, c = ids.c, d = ids.d;
    const uint32_t e01 = sd.ref_lca[(size_t)b * sd.n + a];
    const uint32_t e12 = sd.ref_lca[(size_t)c * sd.n + b];
    const uint32_t e23 = sd.ref_lca[(size_t)d * sd.n + c];
    const uint32_t d01 = e01 >> 16, d12 = e12 >> 16, d23 = e23 >> 16;
    const CT *tup = reinterpret_cast<const CT *>(sd.table) + local_rank * 3;
    const uint32_t n0 = tup[0], n1 = tup[1], n2 = tup[2];
    const uint32_t mx = max(d01, d23);
    uint32_t j1, j2;
    if (d12 < mx) { // ab|cd
        r.resolved = true; r.topo = 0;
        r.q1 = n0; r.q2 = n1; r.q3 = n2;
        j1 = (d01 > d12) ? (e01 & 0xFFFFu) : (e12 & 0xFFFFu);
        j2 = (d23 > d12) ? (e23 & 0xFFFFu) : (e12 & 0xFFFFu);
    } else if (d12 > mx) { // ad|bc
        r.resolved = true; r.topo = 2;
        r.q1 = n2;
        if (sd.frame == 0) { r.q2 = n1; r.q3 = n0; } // S1S3|S2S4 is the crossing pairing
        else { r.q2 = n0; r.q3 = n1; }               // occ(u,z,v,w) = (uz|vw, uv|zw, uw|zv)
        j1 = e12 & 0xFFFFu;
        j2 = (d01 >= d23) ? (e01 & 0xFFFFu) : (e23 & 0xFFFFu);
    } else {
        r.resolved = false; r.topo = 255; r.q1 = r.q2 = r.q3 = 0; j1 = j2 = 0;
    }
    const uint32_t lo = min(j1, j2), hi = max(j1, j2);
    r.key = lo * sd.n_inner + hi;
    return r;
}

// ---- pass 1 / pass 2: one scan of the table -------------------------------------------------
// The round-1 kernels gave every lane one rank at a time: un-rank, three LCA lookups, a device QIC with four 8-byte
// gathers from the log tables (64 different cache lines per wave instruction), then a segmented scan over the wave and
// an LDS hash -- about 310 wave instructions per 64 quartets, 28 ms for the 34 GB table of 512 taxa (1.2 TB/s).
// This kernel walks the ranks lane-SEQUENTIALLY: a lane owns kSK consecutive ranks of a chunk and keeps its quartet
// (a,b,c,d) in registers, incrementing a. Ranks are consecutive in a, and along a row (b,c,d fixed) the reference
// topology and the owning node pair only change where lca(a,b) changes -- at the precomputed breakpoints ref_next[b][a]
// (lca(a,b) climbs down b's ancestor chain as a approaches b; about one change per 11 quartets at 512 taxa). Between
// two changes a quartet costs its tuple, the count permutation, the device QIC and four adds: no un-ranking, no LCA
// lookup, no cross-lane operation; a finished run goes to the workgroup's LDS hash (pass 1) in one insert. log(k) of
// the integer arguments comes from a copy of the table in LDS (<= 15744 entries = every count of up to 15743 trees; 126
// KB), which is why a workgroup is 1024 threads: one per CU, 16 waves. A workgroup takes rounds of 16 x 512 consecutive
// ranks and flushes the hash to memory after each round (the keys of 8192 consecutive ranks fit its 1024 slots).
constexpr int kSK = 8;                              // consecutive ranks per lane and chunk
constexpr int kSThreads = 1024;                     // 16 waves
constexpr int kSChunk = kWave * kSK;                // ranks per wave and round (512)
constexpr int kSRound = (kSThreads / kWave) * kSChunk; // ranks per workgroup and round (8192)
constexpr int kSSlots = 1024;                       // hash slots (power of two) = one per thread at the flush
constexpr uint32_t kKeyEmpty = 0xFFFFFFFFu;
constexpr uint32_t kScanMaxLdsLog = 15744;          // (160 KB - 36 KB hash - slack) / 8

// log(k) of the four integer arguments of one quartet's QIC. Usual case: all below lds_n, four LDS reads behind ONE
// range check; otherwise (counts beyond the LDS copy) the global table, beyond that libm's log -- out of line.
struct Logs4 { double l1, l2, l3, ls; };
__device__ __noinline__ Logs4 scan_logs_slow(const double *__restrict__ logk, uint32_t tbl_n, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t s) {
    Logs4 r;
    r.l1 = q1 < tbl_n ? logk[q1] : log((double)q1);
    r.l2 = q2 < tbl_n ? logk[q2] : log((double)q2);
    r.l3 = q3 < tbl_n ? logk[q3] : log((double)q3);
    r.ls = s < tbl_n ? logk[s] : log((double)s);
    if (q1 == 0) r.l1 = 0.0;   // 0 log 0 = 0 (the table stores 0 at k = 0)
    if (q2 == 0) r.l2 = 0.0;
    if (q3 == 0) r.l3 = 0.0;
    return r;
}
// QuartetScoreComputer.hpp:135-159, device evaluation (orders candidates only; the host re-evaluates the near-minimal
// ones with libm). sum_i (q_i/s) log(q_i/s) = (sum_i q_i log q_i) / s - log s with integer arguments <= number of trees.
// 1/s: v_rcp_f64 + two Newton steps (the same code in both passes, so the values they compare agree).
__device__ __forceinline__ double scan_qic(const double *__restrict__ lds_logk, const ScoreDevice &sd, uint32_t q1, uint32_t q2, uint32_t q3) {
    if ((q1 | q2 | q3) == 0) return 0.0;
    const uint64_t s64 = (uint64_t)q1 + q2 + q3;
    const uint32_t s32 = (uint32_t)min(s64, (uint64_t)0xFFFFFFFFu); // (three u32 counts summing beyond 2^32: only ordering is at stake)
    const double inv_log3 = 0.91023922662683739361;
    Logs4 L;
    if (s32 < sd.lds_n) { L.l1 = lds_logk[q1]; L.l2 = lds_logk[q2]; L.l3 = lds_logk[q3]; L.ls = lds_logk[s32]; } // q_i <= s
    else L = scan_logs_slow(sd.logk, sd.tbl_n, q1, q2, q3, s32);
    double acc = (double)q1 * L.l1;
    acc += (double)q2 * L.l2;
    acc += (double)q3 * L.l3;
    const double sd_ = (double)s32;
    double r = __builtin_amdgcn_rcp(sd_);
    r = fma(fma(-sd_, r, 1.0), r, r);
    r = fma(fma(-sd_, r, 1.0), r, r);
    const double qic = 1.0 + (acc * r - L.ls) * inv_log3;
    return (q1 < q2 || q1 < q3) ? -qic : qic;
}

struct ScanSeg { uint32_t key; uint32_t code; uint32_t end; }; // node pair, count permutation (3 = unresolved), first a of the next run
// classification of the run that starts at (a,b,c,d): the same decision as classify() above
__device__ __forceinline__ ScanSeg scan_classify(const ScoreDevice &sd, uint32_t a, uint32_t b, uint32_t e12, uint32_t e23) {
    ScanSeg r;
    const uint32_t e01 = sd.ref_lca[(size_t)b * sd.n + a];
    r.end = sd.ref_next[(size_t)b * sd.n + a];
    const uint32_t d01 = e01 >> 16, d12 = e12 >> 16, d23 = e23 >> 16;
    const uint32_t mx = max(d01, d23);
    uint32_t j1 = 0, j2 = 0;
    if (d12 < mx) {        // ab|cd
        r.code = 0;
        j1 = (d01 > d12) ? (e01 & 0xFFFFu) : (e12 & 0xFFFFu);
        j2 = (d23 > d12) ? (e23 & 0xFFFFu) : (e12 & 0xFFFFu);
    } else if (d12 > mx) { // ad|bc
        r.code = sd.frame == 0 ? 1u : 2u;
        j1 = e12 & 0xFFFFu;
        j2 = (d01 >= d23) ? (e01 & 0xFFFFu) : (e23 & 0xFFFFu);
    } else r.code = 3;
    r.key = r.code == 3 ? kKeyEmpty : min(j1, j2) * sd.n_inner + max(j1, j2);
    return r;
}

template <int HS> struct HashLds {
    uint32_t key[HS];
    unsigned long long sum[HS * 3];
    long long mn[HS];
};
using ScanLds = HashLds<kSSlots>;

__device__ __forceinline__ void scan_global_add(const ScoreDevice &sd, uint32_t key, unsigned long long s1, unsigned long long s2,
                                                unsigned long long s3, long long mn) {
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 0], s1);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 1], s2);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 2], s3);
    atomicMin(&sd.pair_min[key], mn);
}
template <int HS> __device__ __forceinline__ void scan_flush(HashLds<HS> &h, const ScoreDevice &sd, uint32_t key, unsigned long long s1,
                                                             unsigned long long s2, unsigned long long s3, long long mn) {
    static_assert((HS & (HS - 1)) == 0 && HS >= 64, "hash slots: a power of two");
    uint32_t slot = (key * 2654435761u) >> (32 - __builtin_ctz((unsigned)HS)); // log2(HS) bits
#pragma unroll 1
    for (int probe = 0; probe < 8; ++probe) {
        const uint32_t old = atomicCAS(&h.key[slot], kKeyEmpty, key);
        if (old == kKeyEmpty || old == key) {
            atomicAdd(&h.sum[3 * slot + 0], s1);
            atomicAdd(&h.sum[3 * slot + 1], s2);
            atomicAdd(&h.sum[3 * slot + 2], s3);
            atomicMin(&h.mn[slot], mn);
            return;
        }
        slot = (slot + 1) & (HS - 1);
    }
    scan_global_add(sd, key, s1, s2, s3, mn); // crowded neighbourhood: straight to memory
}

__device__ __forceinline__ uint32_t gcd_u32(uint32_t x, uint32_t y) {
    if (x == 0) return y;
    if (y == 0) return x;
    const int sh = __ffs((int)(x | y)) - 1;
    x >>= (__ffs((int)x) - 1);
    while (y) {
        y >>= (__ffs((int)y) - 1);
        if (x > y) { uint32_t t = x; x = y; y = t; }
        y -= x;
    }
    return x << sh;
}
// Reference tree with a degree-2 root (sd.root_split = k > 0: the lookup ids [0, k) lie under the root's first child): for
// the node pairs (root, v) the reference's processNodePair also walks quartets it does not own (S2 = v's whole side,
// QuartetScoreComputer.hpp:393-396) and takes std::min of THEIR log_score into the same path edges (:448-454). For a leaf
// x alone on one side of the root and three leaves on the other, two of them (c', d') under the two children of v and the
// third (b') outside v's subtree, that second evaluation is countQuartetOccurrences(x, b', c', d'); the owning pair
// (lca(b', v), v) evaluates (b', x, c', d') when b' follows v's subtree in the leaf order -- the same counts with q2 and q3
// exchanged, a log_score that can differ in the last bits (the sum p1 log p1 + p2 log p2 + p3 log p3 is taken in that
// order, :141-156), and the reference keeps the smaller. In sorted ids a < b < c < d that is exactly:
//   a alone on the first side  (a < k <= b) with the three others shaped (b,c)|d  <=>  reference topology ad|bc, or
//   d alone on the second side (c < k <= d) with the three others shaped (a,b)|c  <=>  reference topology ab|cd.
// Such a quartet's candidate carries kCandSwap and qs_score_finish evaluates both orders.
__device__ __forceinline__ bool root_swapped(const ScoreDevice &sd, uint32_t code, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t k = sd.root_split;
    return k != 0 && ((code == 1 && a < k && b >= k) || (code == 0 && c < k && d >= k));
}
// pass 2: record the gcd-reduced triple of a near-minimal quartet (log_score(k q) is bit-identical to log_score(q))
// swp (bit 63 of the slot, kCandSwap): the reference evaluates this quartet a second time with q2 and q3 exchanged (root_swapped below)
__device__ __noinline__ void scan_candidate(const ScoreDevice &sd, uint32_t key, uint32_t q1, uint32_t q2, uint32_t q3, bool swp = false) {
    uint32_t g = gcd_u32(gcd_u32(q1, q2), q3);
    if (g == 0) g = 1;
    const uint32_t a = q1 / g, b = q2 / g, c = q3 / g;
    unsigned long long *slots = sd.pair_cand + (size_t)key * kCand;
    if ((a | b | c) >> 21) {   // does not fit the packed slot: this node pair is finished by qs_score_overflow
        atomicOr(&sd.flags[0], 2u);
        __hip_atomic_store(&slots[kCand - 1], kCandOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned long long packed = ((unsigned long long)a << 42) | ((unsigned long long)b << 21) | c | (swp ? kCandSwap : 0ull);
    if (packed >= kCandOverflow) {   // (a flagged triple of counts 2^21 - 1 would read as a marker: finished by qs_score_overflow)
        atomicOr(&sd.flags[0], 2u);
        __hip_atomic_store(&slots[kCand - 1], kCandOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // cheap pre-check avoids hammering CAS when thousands of quartets share one triple
    for (uint32_t s = 0; s < sd.cand_limit; ++s) {
        unsigned long long cur = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == packed || cur == kCandOverflow) return;
        if (cur == kCandEmpty) {
            const unsigned long long old = atomicCAS(&slots[s], kCandEmpty, packed);
            if (old == kCandEmpty || old == packed) return;
            if (old == kCandOverflow) return;
        }
    }
    // more than kCand distinct near-minimal triples: mark the pair, qs_score_overflow lists its quartets
    atomicOr(&sd.flags[0], 1u);
    __hip_atomic_store(&slots[kCand - 1], kCandOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename CT> __device__ __forceinline__ void scan_load_tuple(const CT *p, uint32_t &n0, uint32_t &n1, uint32_t &n2) {
    if (sizeof(CT) == 4) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
        const u32x3 v = *reinterpret_cast<const u32x3_a4 *>(p);
        n0 = v.x; n1 = v.y; n2 = v.z;
    } else { n0 = p[0]; n1 = p[1]; n2 = p[2]; }
}

template <typename CT, int PASS>
__global__ __launch_bounds__(kSThreads) void score_scan_kernel(ScoreDevice sd, uint32_t rounds_per_wg, double tol) {
    extern __shared__ __align__(16) unsigned char scan_smem[];
    ScanLds &hash = *reinterpret_cast<ScanLds *>(scan_smem);                   // used by pass 1 only
    double *lds_logk = reinterpret_cast<double *>(scan_smem + (PASS == 1 ? sizeof(ScanLds) : 0));
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    for (uint32_t i = tid; i < sd.lds_n; i += kSThreads) lds_logk[i] = sd.logk[i];
    if (PASS == 1) {
        hash.key[tid] = kKeyEmpty;
        hash.sum[3 * tid] = hash.sum[3 * tid + 1] = hash.sum[3 * tid + 2] = 0;
        hash.mn[tid] = kSortableMax;
    }
    __syncthreads();
    const CT *table = reinterpret_cast<const CT *>(sd.table);
    const uint64_t wg_base = (uint64_t)blockIdx.x * rounds_per_wg * kSRound;
    Ids4 st;
    bool have = false;
    for (uint32_t round = 0; round < rounds_per_wg; ++round) {
        const uint64_t round_base = wg_base + (uint64_t)round * kSRound;
        if (round_base >= sd.n_tuples) break;                                   // uniform over the workgroup
        const uint64_t r0 = round_base + (uint64_t)wave * kSChunk + (uint64_t)lane * kSK;
        const uint32_t cnt = r0 < sd.n_tuples ? (uint32_t)min((uint64_t)kSK, sd.n_tuples - r0) : 0u;
        if (cnt > 0) {
            if (!have) { unrank4(r0 + sd.rank_lo, st.a, st.b, st.c, st.d); have = true; }
            else st = decode_near(st, kSRound);
            uint32_t a = st.a, b = st.b, c = st.c, d = st.d;
            uint32_t e12 = sd.ref_lca[(size_t)c * sd.n + b], e23 = sd.ref_lca[(size_t)d * sd.n + c];
            ScanSeg seg = scan_classify(sd, a, b, e12, e23);
            bool swp = PASS >= 2 && root_swapped(sd, seg.code, a, b, c, d);
            unsigned long long s1 = 0, s2 = 0, s3 = 0;
            long long mn = kSortableMax;
            double thr = 0.0;
            bool marked = false;   // pass 3: this node pair's candidate slots overflowed in pass 2
            if (PASS >= 2 && seg.code != 3) thr = sortable_to_f64(sd.pair_min[seg.key]) + tol;
            if (PASS == 3 && seg.code != 3) marked = sd.pair_cand[(size_t)seg.key * kCand + kCand - 1] == kCandOverflow;
            // the lane's kSK consecutive tuples, one ahead (the loop is NOT unrolled: the run-end block exists once)
            const CT *tp = table + r0 * 3;
            uint32_t n0, n1, n2, p0 = 0, p1 = 0, p2 = 0;
            scan_load_tuple<CT>(tp, n0, n1, n2);
#pragma unroll 1
            for (uint32_t i = 0; i < cnt; ++i) {
                if (i + 1 < cnt) scan_load_tuple<CT>(tp + (i + 1) * 3, p0, p1, p2);
                if (a == seg.end) {       // the run ends here: next breakpoint of lca(a,b), or the end of the row
                    if (PASS == 1 && seg.code != 3) scan_flush(hash, sd, seg.key, s1, s2, s3, mn);
                    if (a == b) {         // next row: ranks carry a -> b -> c -> d
                        a = 0; ++b;
                        if (b == c) { b = 1; ++c; if (c == d) { c = 2; ++d; } }
                        e12 = sd.ref_lca[(size_t)c * sd.n + b]; e23 = sd.ref_lca[(size_t)d * sd.n + c];
                    }
                    seg = scan_classify(sd, a, b, e12, e23);
                    swp = PASS >= 2 && root_swapped(sd, seg.code, a, b, c, d);
                    s1 = s2 = s3 = 0; mn = kSortableMax;
                    if (PASS >= 2 && seg.code != 3) thr = sortable_to_f64(sd.pair_min[seg.key]) + tol;
                    if (PASS == 3) marked = seg.code != 3 && sd.pair_cand[(size_t)seg.key * kCand + kCand - 1] == kCandOverflow;
                }
                if (seg.code != 3) {
                    const uint32_t q1 = seg.code == 0 ? n0 : n2;
                    const uint32_t q2 = seg.code == 2 ? n0 : n1;
                    const uint32_t q3 = seg.code == 0 ? n2 : (seg.code == 1 ? n0 : n1);
                    const double qic = scan_qic(lds_logk, sd, q1, q2, q3);
                    if (PASS == 1) {
                        s1 += q1; s2 += q2; s3 += q3;
                        const long long sq = f64_to_sortable(qic);
                        mn = sq < mn ? sq : mn;
                    } else if (PASS == 2) { if (qic <= thr) scan_candidate(sd, seg.key, q1, q2, q3, swp); }
                    else if (marked && qic <= thr) {   // pass 3: (key, q1, q2, q3) of every near-minimal quartet of a marked pair
                        const unsigned long long at = atomicAdd(sd.list_count, 1ull);
                        if (at < sd.list_cap) {
                            unsigned long long *e = sd.list + at * 4;
                            e[0] = seg.key | (swp ? kListSwap : 0ull); e[1] = q1; e[2] = q2; e[3] = q3;
                        }
                    }
                }
                ++a;
                n0 = p0; n1 = p1; n2 = p2;
            }
            if (PASS == 1 && seg.code != 3) scan_flush(hash, sd, seg.key, s1, s2, s3, mn);
        }
        if (PASS == 1) {   // the hash holds the node pairs of this round's 8192 ranks: one slot per thread to memory
            __syncthreads();
            const uint32_t k = hash.key[tid];
            if (k != kKeyEmpty) {
                scan_global_add(sd, k, hash.sum[3 * tid], hash.sum[3 * tid + 1], hash.sum[3 * tid + 2], hash.mn[tid]);
                hash.key[tid] = kKeyEmpty;
                hash.sum[3 * tid] = hash.sum[3 * tid + 1] = hash.sum[3 * tid + 2] = 0;
                hash.mn[tid] = kSortableMax;
            }
            __syncthreads();
        }
    }
}

// ---- the bundle kernel: passes 1 and 2, one wave = 64 rows with the same b -------------------------
// score_scan_kernel is bound by instruction issue (r02 counters at 512 taxa: the SIMDs issue ~80 % of the time, 180
// instructions per 64 quartets in pass 1): its lanes sit in different rows, so in nearly every iteration SOME lane
// reaches a run end and the whole wave walks through the re-classification and the hash insert.
// Along a row (b,c,d fixed, a = 0..b-1) the node pair of the quartet changes only where lca(a,b) changes, and those
// places depend on b alone. So this kernel gives the 64 lanes of a wave 64 rows WITH THE SAME b -- the rows of 64
// consecutive pairs (c,d), b < c < d, in the table's own (d-major) order -- and walks them in lockstep:
//   * a, the run ends (ref_next[b][a]) and lca(a,b) are wave-uniform (scalar); only lca(b,c), lca(c,d) differ per lane;
//   * the re-classification is a uniform branch taken ~10 times per row instead of in every iteration;
//   * per quartet a lane does: raw sums and the device QIC from a k log k table in LDS -- sum_i T[n_i] * (1/s) - log s
//     with 1/s and log s cached per lane while the tuple sum s repeats; its row comes in chunks of 8 tuples = 96 bytes
//     requested as six 16-byte loads back to back;
//   * at a change of node pair the lanes that change insert their piece into the workgroup's LDS hash (neighbouring
//     lanes mostly share the pair: same-address LDS atomics), which goes to memory after every round (the waves of a
//     workgroup take consecutive b and the same group of pairs: ~100 distinct node pairs per round).
// Rows are contiguous (b tuples at rank C(d,4)+C(c,3)+C(b,2)), so a lane streams its own row; a wave reads 64 streams.
// The host plans the rounds for the rank range at hand (plan_bundles): rows that lie completely inside it; the at most
// two partial rows at its ends (views of a reduce-scattered table) go through score_scan_kernel.
__device__ __forceinline__ uint32_t pair_classify(const ScoreDevice &sd, uint32_t e01, uint32_t e12, uint32_t e23, uint32_t &code) {
    const uint32_t d01 = e01 >> 16, d12 = e12 >> 16, d23 = e23 >> 16;
    const uint32_t mx = max(d01, d23);
    const uint32_t n01 = e01 & 0xFFFFu, n12 = e12 & 0xFFFFu, n23 = e23 & 0xFFFFu;
    // ab|cd: (d01 > d12 ? n01 : n12, d23 > d12 ? n23 : n12);  ad|bc: (n12, d01 >= d23 ? n01 : n23)
    const bool abcd = d12 < mx, adbc = d12 > mx;
    const uint32_t j1 = abcd ? (d01 > d12 ? n01 : n12) : n12;
    const uint32_t j2 = abcd ? (d23 > d12 ? n23 : n12) : (d01 >= d23 ? n01 : n23);
    code = abcd ? 0u : (adbc ? (sd.frame == 0 ? 1u : 2u) : 3u);
    return (abcd || adbc) ? min(j1, j2) * sd.n_inner + max(j1, j2) : kKeyEmpty;
}
__device__ __forceinline__ void permute_counts(uint32_t code, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t &q1, uint32_t &q2, uint32_t &q3) {
    q1 = code == 0 ? n0 : n2;
    q2 = code == 2 ? n0 : n1;
    q3 = code == 0 ? n2 : (code == 1 ? n0 : n1);
}

// device QIC of a raw tuple, only the sign depends on which count is the reference topology's (first_is_n0)
struct QicCache { uint32_t s; double r, ls; };
__device__ __noinline__ double bundle_qic_slow(const double *__restrict__ logk, uint32_t tbl_n, uint32_t n0, uint32_t n1, uint32_t n2) {
    const uint64_t s64 = (uint64_t)n0 + n1 + n2;
    if (s64 == 0) return 0.0;
    const double sd_ = (double)s64;
    auto L = [&](uint64_t k) { return k == 0 ? 0.0 : (k < tbl_n ? logk[k] : log((double)k)); };
    const double acc = (double)n0 * L(n0) + (double)n1 * L(n1) + (double)n2 * L(n2);
    return 1.0 + (acc / sd_ - L(s64)) * 0.91023922662683739361;
}
__device__ __forceinline__ double bundle_qic(const double *__restrict__ t1, const ScoreDevice &sd, QicCache &qc, uint32_t n0, uint32_t n1,
                                             uint32_t n2, bool first_is_n0) {
    const uint32_t mx = max(max(n0, n1), n2);
    const uint32_t s = n0 + n1 + n2;                       // (no wrap below: mx < 2^30)
    double qic;
    if (mx < (1u << 30) && s < sd.lds_n) {
#if defined(QS_PROBE_SCORE_NOLDS)        /* timing probes only (wrong scores): what the three LDS look-ups cost (profiles/r06_experiments.md 5) */
        const double acc = (double)n0 * 1.5 + (double)n1 * 2.5 + (double)n2 * 3.5;
#elif defined(QS_PROBE_SCORE_F32LOG)
        const float f0 = (float)n0, f1 = (float)n1, f2 = (float)n2;
        const double acc = (double)(f0 * __builtin_amdgcn_logf(f0 + 1.0f) + f1 * __builtin_amdgcn_logf(f1 + 1.0f) + f2 * __builtin_amdgcn_logf(f2 + 1.0f)) * 0.6931471805599453;
#else
        const double acc = t1[n0] + t1[n1] + t1[n2];       // sum_i n_i log n_i
#endif
        if (s != qc.s) {                                   // 1/s and log s: kept while the tuple sum repeats
            qc.s = s;
            if (s == 0) { qc.r = 0.0; qc.ls = 1.0986122886681098; }   // all-zero tuple: QIC 0 (QuartetScoreComputer.hpp:136)
            else {
                const double sd_ = (double)s;
                double r = __builtin_amdgcn_rcp(sd_);
                r = fma(fma(-sd_, r, 1.0), r, r);
                r = fma(fma(-sd_, r, 1.0), r, r);
                qc.r = r; qc.ls = t1[s] * r;
            }
        }
        qic = fma(fma(acc, qc.r, -qc.ls), 0.91023922662683739361, 1.0);
    } else qic = bundle_qic_slow(sd.logk, sd.tbl_n, n0, n1, n2);
    const uint32_t q1 = first_is_n0 ? n0 : n2;
    return q1 != mx ? -qic : qic;
}

// waves of a workgroup (= consecutive b of a round). A wave reads 64 streams, and a CU keeps one workgroup (LDS): more
// waves hide more latency, but with too many streams per CU the 32 KB L1 loses a row's cache line between two of the
// lane's accesses and the requests to the L2 multiply: the L1 sustains only ~60 outstanding misses per CU, so the kernel's
// time is (requests to L2) x (their latency) (profiles/r02_experiments.md: 512 taxa, 12-byte loads, pass 1 / pass 2 in ms
// at 4, 6, 8, 10, 16 waves: 15.8/15.7, 11.8/12.1, 16.6/11.2, 17.8/15.2, 19.7/18.6; with 16-byte loads of whole chunks
// 10.3/9.4 at 8/10 waves, flat from 7 to 12).
#ifndef QS_BUNDLE_CH
#define QS_BUNDLE_CH 8
#endif
#ifndef QS_BUNDLE_CH16
#define QS_BUNDLE_CH16 16   /* u16 cells: 16 tuples = the same 96 bytes (1024-taxon shard: 8.3 -> 7.6 ms in pass 1) */
#endif
#ifndef QS_BUNDLE_W1
#define QS_BUNDLE_W1 12   /* round 3: the LOGGING pass 1 (the default from 1 GB) wants more waves than the plain one: 8 / 10 / 12 / 14 / 16 waves =
                           * 12.4 / 11.8 / 11.7 / 12.1 / 12.1 ms incl. the samples at 512 taxa, 14.0 / 12.8 / 12.3 / 12.4 / 12.4 on reference + NNI trees;
                           * the plain pass 1: 9.95 / 9.89 / 10.08 / 10.56 / 10.02 (profiles/r03_experiments.md 13) */
#endif
#ifndef QS_BUNDLE_W2
#define QS_BUNDLE_W2 10
#endif
typedef uint32_t qs_u32x4 __attribute__((ext_vector_type(4)));
typedef qs_u32x4 qs_u32x4_a2 __attribute__((aligned(2)));   // rows start at any tuple: 4-byte (u32 cells) or 2-byte (u16 cells) aligned
constexpr int kBundleWaves1 = QS_BUNDLE_W1, kBundleWaves2 = QS_BUNDLE_W2;

// COOP (QS_TUNE_SCORE_LOAD = 1): the 96-byte chunk of a row is loaded by EIGHT lanes (12 bytes each, one load instruction
// covers 8 rows x 96 contiguous bytes instead of 64 x 16 scattered ones) into a per-wave staging area in LDS, from which
// every lane reads its own row's chunk back. The L1 sustains ~64 misses per CU whatever their size: the per-lane 16-byte
// loads make one L2 request per load, this makes one per 64-byte segment. The hash shrinks to 512 slots to make room.
constexpr int kBundleHashCoop = 512;
constexpr int kStageRow = 96;                          // bytes per row and chunk in the staging area
// ... which lie kStagePitch bytes apart: 28 dwords, so that the eight lanes a 16-byte LDS read serves per clock start in eight different
// groups of four banks (a pitch of 96 bytes = 24 dwords puts lanes l and l + 4 on the same banks)
constexpr int kStagePitch = 112;
#ifndef QS_BUNDLE_WC
#define QS_BUNDLE_WC 8    /* waves per workgroup with cooperative loads, both passes: 8 x 64 x 112 bytes of staging leave the k log k table 10 900 entries
                           * (12 waves: 8 100 -- at 10 000 trees every tuple then took the range-checked slow path, which is what round 3 measured) */
#endif
constexpr int kBundleWavesCoop = QS_BUNDLE_WC;
template <int PASS, int WAVES, bool COOP> constexpr size_t bundle_lds_fixed() {   // COOP = load mode 1
    return (PASS == 1 ? (COOP ? sizeof(HashLds<kBundleHashCoop>) : sizeof(ScanLds)) : 0) + (COOP ? (size_t)WAVES * kWave * kStagePitch : 0);
}
typedef uint32_t qs_u32x3 __attribute__((ext_vector_type(3)));
typedef qs_u32x3 qs_u32x3_a2 __attribute__((aligned(2)));
typedef const qs_u32x3_a2 __attribute__((address_space(1))) *qs_u32x3_gptr;   // (global_load, not flat_load, from an address kept as an integer)
// LM = 2 (QS_TUNE_SCORE_LOAD = 2): the per-lane loads of the NEXT chunk are issued before the current chunk is processed (two
// chunks of every row in flight; tools/row_bw.hip: the bare access pattern reads 4.75 TB/s with one chunk in flight, 5.67 with two).
template <typename CT, int PASS, int WAVES, int LM>
__global__ __launch_bounds__(WAVES * kWave) void score_bundle_kernel(ScoreDevice sd, double tol) {
    constexpr bool COOP = LM == 1 || LM == 3, PF = LM == 2, CPF = LM == 3;   // 3 = cooperative loads, next chunk requested ahead
    static_assert(PASS == 1 || PASS == 2, "pass 3 stays on score_scan_kernel");
    constexpr int kBundleWaves = WAVES, kBThreads = WAVES * kWave;
    constexpr int HS = COOP ? kBundleHashCoop : kSSlots;
    extern __shared__ __align__(16) unsigned char scan_smem[];
    HashLds<HS> &hash = *reinterpret_cast<HashLds<HS> *>(scan_smem);           // used by pass 1 only
    unsigned char *stage_all = scan_smem + (PASS == 1 ? sizeof(HashLds<HS>) : 0);   // COOP: WAVES x 64 rows x 96 bytes
    double *t1 = reinterpret_cast<double *>(scan_smem + bundle_lds_fixed<PASS, WAVES, COOP>());   // k log k
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    for (uint32_t i = tid; i < sd.lds_n; i += kBThreads) t1[i] = (double)i * sd.logk[i];
    if (PASS == 1) {
        for (uint32_t t = tid; t < (uint32_t)HS; t += kBThreads) {
            hash.key[t] = kKeyEmpty;
            hash.sum[3 * t] = hash.sum[3 * t + 1] = hash.sum[3 * t + 2] = 0;
            hash.mn[t] = kSortableMax;
        }
    }
    __syncthreads();
    const CT *table = reinterpret_cast<const CT *>(sd.table);
    const uint32_t *__restrict__ L = sd.ref_lca;
    const uint32_t n = sd.n;
    const double kHuge = sortable_to_f64(kSortableMax);
    constexpr uint32_t kLogChunk = 64;                  // records a wave reserves at a time in the candidate log (pass 1, single-read scoring)
    unsigned long long wbase = 0;                       // the wave's current chunk: next free record, records left (wave-uniform)
    uint32_t wleft = 0;
    // The minima-only pre-pass of the single-read scoring (sd.sample): a sample of the table -- one 96-byte chunk of every row
    // in S, or one round in S -- lowers pair_min to the sample's minimum and adds NOTHING to the sums, so that the logging
    // pass that follows starts every run with a bound close to the final minimum (min is idempotent: taking the sample's
    // quartets twice changes nothing).
    // Estimate mode (bit 17): the same over ANOTHER sample (offset S/2), with the logging condition evaluated against the
    // bounds the first pre-pass left and the hits only counted (sd.list_count += hits): S x that count predicts the log of
    // the full pass, and the host falls back to two plain passes when it would not fit (tie-heavy tables: reference + NNI
    // trees put thousands of quartets with one and the same triple at a node pair's bound).
    const bool sampling = PASS == 1 && sd.sample != 0;
    const bool by_round = (sd.sample >> 16) & 1u;
    const bool estimate = sampling && ((sd.sample >> 17) & 1u);
    const uint32_t s_n = sd.sample & 0xFFFFu, s_off = estimate ? s_n / 2 : 0u;
    const uint32_t smask = by_round ? 0xFFFFFFFFu : s_n - 1u;                               // chunk sampling: mask of S
    const uint32_t rstep = sampling && by_round ? s_n : 1u;
    unsigned long long west = 0;                         // estimate mode: hits of this wave (wave-uniform)
    for (uint32_t round = blockIdx.x * rstep + (by_round ? s_off : 0u); round < sd.n_rounds; round += gridDim.x * rstep) {
        const uint32_t rk = sd.bundle_rounds[2 * round], rg = sd.bundle_rounds[2 * round + 1];
        const uint32_t b = __builtin_amdgcn_readfirstlane(rk * kBundleWaves + wave);
        const uint32_t pcnt = b < n ? sd.bundle_pcnt[b] : 0u;
        if (pcnt > rg * kWave) {                                       // uniform over the wave
            const bool live = rg * kWave + lane < pcnt;
            const uint32_t p = sd.bundle_plo[b] + min(rg * kWave + lane, pcnt - 1);   // (idle lanes repeat the last row, and never hand anything in)
            uint32_t c, d;
            unrank2(p, c, d);
            c += b + 1; d += b + 1;
            const CT *row = table + (rank4(0, b, c, d) - sd.rank_lo) * 3;
            // COOP: in load instruction i this lane fetches piece (lane & 7) of the chunk of row 8 i + lane / 8
            unsigned long long rb[COOP ? 8 : 1];
            unsigned char *stg = stage_all + (size_t)wave * (kWave * kStagePitch);
            if (COOP) {
                const unsigned long long rp = (unsigned long long)(uintptr_t)row;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int src = 8 * i + (int)(lane >> 3);
                    rb[i] = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(rp >> 32), src, kWave) << 32) | (uint32_t)__shfl((int)(uint32_t)rp, src, kWave);
                }
            }
            const uint32_t e12 = L[(size_t)c * n + b], e23 = L[(size_t)d * n + c];
            const uint32_t *__restrict__ lrow = L + (size_t)b * n;
            const uint16_t *__restrict__ nrow = sd.ref_next + (size_t)b * n;
            constexpr int CH = COOP ? (sizeof(CT) == 2 ? 16 : 8) : (sizeof(CT) == 2 ? QS_BUNDLE_CH16 : QS_BUNDLE_CH);   // tuples a lane requests at once: 96 bytes of its row, back to back, so that
            uint32_t q[CH][3];                      // the requests for one cache line meet in the L1 while it is still pending
            uint32_t key = kKeyEmpty, code = 3;
            bool first_is_n0 = true, swp = false;
            unsigned long long S0 = 0, S1 = 0, S2 = 0;
            double mn = kHuge, thr = -kHuge;
            uint32_t h0 = 0, h1 = 0, h2 = 0;        // pass 2 / logging pass 1: the lane's previous near-minimal tuple in this run
            bool hprev = false;
            // Pass 1 with a candidate log (sd.list, single-read scoring): a quartet is logged when its QIC is within tol of
            // min(the node pair's minimum as this lane last saw it in memory, the lane's own running minimum of the run).
            // Both bounds are >= the pair's FINAL minimum, so every quartet within tol of the final minimum is in the log;
            // score_log_kernel filters the log against the final minima afterwards -- the table is read once, not twice.
            const bool logging = PASS == 1 && (sd.list != nullptr || estimate);
            // (wave-uniform) stop logging once the log is full: the counter is looked at once per round
            bool log_on = estimate || (logging && __builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(sd.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sd.list_cap)) != 0);
            double bound = -kHuge;                  // nothing is logged before the first node pair (a run may start unresolved: key stays empty)
            QicCache qc = {0xFFFFFFFFu, 0.0, 0.0};
            uint32_t end = 0;
            constexpr int NVP = CH * 3 * (int)sizeof(CT) / 16;
            qs_u32x4 nx[PF ? NVP : 1];                  // PF: the next chunk, already requested
            qs_u32x3 nxp[CPF ? 8 : 1];                  // CPF: the same for the cooperative loads
            bool nx_valid = false;                      // (uniform)
            for (uint32_t a0 = 0; a0 < b; a0 += CH) {                   // uniform
                if (sampling) {                                         // uniform: the pre-pass takes one chunk in S and classifies afresh there
                    if (smask != 0xFFFFFFFFu && (((a0 / CH) + round + s_off) & smask) != 0) continue;
                    end = a0;
                }
                if (a0 + CH <= b) {                                     // uniform: the whole chunk lies in the row
                    constexpr int NV = CH * 3 * (int)sizeof(CT) / 16;   // 16-byte loads: 6 (u32 cells) / 3 (u16 cells)
                    static_assert(!COOP || NV * 16 == kStageRow, "cooperative loads: chunks of 96 bytes");
                    uint32_t w[NV * 4];
                    if (COOP) {
                        const uint32_t boff = a0 * 3u * (uint32_t)sizeof(CT) + 12u * (lane & 7u);
                        qs_u32x3 part[8];
                        if (CPF && nx_valid) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) part[i] = nxp[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) part[i] = *(qs_u32x3_gptr)(uintptr_t)(rb[i] + boff);
                        }
                        if (CPF) {
                            nx_valid = !(sampling && smask != 0xFFFFFFFFu) && a0 + 2 * CH <= b;
                            if (nx_valid) {
                                const uint32_t boffn = boff + (uint32_t)(CH * 3 * sizeof(CT));
#pragma unroll
                                for (int i = 0; i < 8; ++i) nxp[i] = *(qs_u32x3_gptr)(uintptr_t)(rb[i] + boffn);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            *reinterpret_cast<qs_u32x3 *>(stg + (8 * i + (lane >> 3)) * kStagePitch + 12u * (lane & 7u)) = part[i];
                        // the rows were written by other lanes of this wave: LDS operations of a wave complete in order, the
                        // fences only keep the compiler from moving the reads in front of the writes
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            const qs_u32x4 v = *reinterpret_cast<const qs_u32x4 *>(stg + lane * kStagePitch + 16 * j);
                            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    } else if (PF) {
                        qs_u32x4 cur[NV];
                        if (nx_valid) {
#pragma unroll
                            for (int j = 0; j < NV; ++j) cur[j] = nx[j];
                        } else {
                            const qs_u32x4_a2 *src = reinterpret_cast<const qs_u32x4_a2 *>(row + 3 * (size_t)a0);
#pragma unroll
                            for (int j = 0; j < NV; ++j) cur[j] = src[j];
                        }
                        // (a pre-pass that takes one chunk in S skips chunks: nothing to request ahead there)
                        nx_valid = !(sampling && smask != 0xFFFFFFFFu) && a0 + 2 * CH <= b;
                        if (nx_valid) {
                            const qs_u32x4_a2 *srcn = reinterpret_cast<const qs_u32x4_a2 *>(row + 3 * (size_t)(a0 + CH));
#pragma unroll
                            for (int j = 0; j < NV; ++j) nx[j] = srcn[j];
                        }
#pragma unroll
                        for (int j = 0; j < NV; ++j) { w[4 * j] = cur[j].x; w[4 * j + 1] = cur[j].y; w[4 * j + 2] = cur[j].z; w[4 * j + 3] = cur[j].w; }
                    } else {
                        const qs_u32x4_a2 *src = reinterpret_cast<const qs_u32x4_a2 *>(row + 3 * (size_t)a0);
#pragma unroll
                        for (int j = 0; j < NV; ++j) { const qs_u32x4 v = src[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int e = 3 * u + k;                    // element index in the chunk
                            q[u][k] = sizeof(CT) == 4 ? w[e] : ((w[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < CH; ++u)
                        if (a0 + u < b) scan_load_tuple<CT>(row + 3 * (size_t)(a0 + u), q[u][0], q[u][1], q[u][2]);
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const uint32_t a = a0 + u;
                    if (a < b) {                                        // uniform
                        if (a == end) {                                 // uniform: lca(a,b) changes here
                            const uint32_t e01 = __builtin_amdgcn_readfirstlane(lrow[a]);
                            end = __builtin_amdgcn_readfirstlane((uint32_t)nrow[a]);
                            uint32_t ncode;
                            uint32_t nkey = pair_classify(sd, e01, e12, e23, ncode);
                            if (!live) { nkey = kKeyEmpty; ncode = 3; }
                            const bool changed = nkey != key;
                            if (PASS == 1) {
                                const bool f = changed && key != kKeyEmpty;
                                if (__any(f)) {
                                    if (f) scan_flush(hash, sd, key, code == 0 ? S0 : S2, code == 2 ? S0 : S1,
                                                      code == 0 ? S2 : (code == 1 ? S0 : S1), f64_to_sortable(mn));
                                }
                                if (changed) {
                                    S0 = S1 = S2 = 0; mn = kHuge;
                                    // the pair's minimum as this CU sees it (a plain load: the L1 may hold an older value, but 34 GB of
                                    // table stream through it, and ANY older value is still an upper bound of the final minimum; an
                                    // agent-scope load past the L1 made pass 1 4x slower: 46.5 ms against 10.3 at 512 taxa)
                                    if (logging) { bound = nkey != kKeyEmpty ? sortable_to_f64(sd.pair_min[nkey]) : -kHuge; hprev = false; }
                                }
                            } else if (changed) {
                                thr = nkey != kKeyEmpty ? sortable_to_f64(sd.pair_min[nkey]) + tol : -kHuge;
                                hprev = false;
                            }
                            key = nkey; code = ncode; first_is_n0 = ncode == 0;
                            swp = (PASS == 2 || logging) && root_swapped(sd, ncode, a, b, c, d);   // (constant along a run: a crossing of the root split is a change of lca(a,b))
                        }
                        const uint32_t n0 = q[u][0], n1 = q[u][1], n2 = q[u][2];
                        const double qic = bundle_qic(t1, sd, qc, n0, n1, n2, first_is_n0);
                        if (PASS == 1) {
                            if (!sampling) { S0 += n0; S1 += n1; S2 += n2; }
                            if (logging) {
                                const bool near = qic <= fmin(bound, mn) + tol;
                                // (everything else only in the rare wave step that holds a near quartet at all: with bounds from the
                                // pre-pass the common step costs the comparison above and the ballot of `near`)
                                const unsigned long long nears = __ballot(near);
                                bool hit = false;
                                unsigned long long hits = 0ull;
                                unsigned long long packed = 0;
                                if (nears) {
                                    hit = near && !(hprev && n0 == h0 && n1 == h1 && n2 == h2);
                                    // Ties at the bound: in a table of similar trees thousands of quartets of a node pair carry one
                                    // and the same triple, and every lane's every run would log it once (90 M records at 512 taxa x
                                    // 10000 reference + NNI trees). The pair's last LOGGED triple is kept in memory; a quartet that
                                    // repeats it adds nothing to the candidates (they are sets of triples). A stale value read through
                                    // the L1 is an older logged triple of this pass or "none": skipping stays sound.
                                    if (sd.last_trip != nullptr && log_on) {
                                        if (hit) {
                                            uint32_t q1, q2, q3;
                                            permute_counts(code, n0, n1, n2, q1, q2, q3);
                                            // (three counts below 2^21 pack without loss; anything larger is simply never filtered: all ones)
                                            packed = ((q1 | q2 | q3) >> 21) ? ~0ull : ((unsigned long long)q1 << 42) | ((unsigned long long)q2 << 21) | q3 | (swp ? kCandSwap : 0ull);
                                            if (packed != ~0ull && sd.last_trip[key] == packed) hit = false;
                                        }
                                    }
                                    hits = log_on ? __ballot(hit) : 0ull;
                                    if (near) { h0 = n0; h1 = n1; h2 = n2; }
                                }
                                if (estimate) {
                                    west += (unsigned long long)__builtin_popcountll(hits);
                                    if (hit && sd.last_trip != nullptr) sd.last_trip[key] = packed;   // (cleared again before the full pass)
                                } else if (hits) {
                                    // The wave writes into a chunk of kLogChunk records it has reserved with ONE atomic on the log's
                                    // counter (a single counter word takes ~88 updates per microsecond: one update per logging wave
                                    // instruction made pass 1 4.5x slower). Records of a chunk that stay unwritten keep the key the
                                    // host's memset put there (all ones) and are skipped by score_log_kernel.
                                    const uint32_t cnt = (uint32_t)__builtin_popcountll(hits);
                                    if (cnt > wleft) {                  // uniform
                                        unsigned long long nb = 0;
                                        if (lane == 0) nb = atomicAdd(sd.list_count, (unsigned long long)kLogChunk);
                                        nb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nb >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
                                        wbase = nb; wleft = kLogChunk;
                                        if (nb + kLogChunk > sd.list_cap) { log_on = false; wleft = 0; }   // full: the caller falls back to a second pass
                                    }
                                    if (log_on) {
                                        if (hit) {
                                            uint32_t q1, q2, q3;
                                            permute_counts(code, n0, n1, n2, q1, q2, q3);
                                            unsigned long long *rec = sd.list + 4 * (wbase + (unsigned long long)__builtin_popcountll(hits & ((1ull << lane) - 1ull)));
                                            rec[0] = key | (swp ? kListSwap : 0ull); rec[1] = q1; rec[2] = q2; rec[3] = q3;
                                            if (sd.last_trip != nullptr) sd.last_trip[key] = packed;
                                        }
                                        wbase += cnt; wleft -= cnt;
                                    }
                                }
                                hprev = near;
                            }
                            mn = fmin(mn, qic);
                        } else {
                            // a near-minimal quartet; a repeat of the lane's previous one in this run -- ties such as
                            // (m,0,0) -- is the same packed triple, nothing to record
                            const bool near = qic <= thr;
                            const bool hit = near && !(hprev && n0 == h0 && n1 == h1 && n2 == h2);
                            if (__any(hit)) {
                                if (hit) {
                                    uint32_t q1, q2, q3;
                                    permute_counts(code, n0, n1, n2, q1, q2, q3);
                                    scan_candidate(sd, key, q1, q2, q3, swp);
                                }
                            }
                            if (near) { h0 = n0; h1 = n1; h2 = n2; }
                            hprev = near;
                        }
                    }
                }
            }
            if (PASS == 1) {
                const bool f = key != kKeyEmpty;
                if (f) scan_flush(hash, sd, key, code == 0 ? S0 : S2, code == 2 ? S0 : S1, code == 0 ? S2 : (code == 1 ? S0 : S1),
                                  f64_to_sortable(mn));
            }
        }
        if (PASS == 1) {   // the hash holds the node pairs of this round: one slot per thread to memory
            __syncthreads();
            for (uint32_t t = tid; t < (uint32_t)HS; t += kBThreads) {
                const uint32_t k = hash.key[t];
                if (k != kKeyEmpty) {
                    scan_global_add(sd, k, hash.sum[3 * t], hash.sum[3 * t + 1], hash.sum[3 * t + 2], hash.mn[t]);
                    hash.key[t] = kKeyEmpty;
                    hash.sum[3 * t] = hash.sum[3 * t + 1] = hash.sum[3 * t + 2] = 0;
                    hash.mn[t] = kSortableMax;
                }
            }
            __syncthreads();
        }
    }
    if (estimate && lane == 0 && west) atomicAdd(sd.list_count, west);
}

// The rounds of the bundle kernel for the rank range [r0, r1) of an n-taxon table (host). For every b the rows (b,c,d)
// that lie completely inside the range are, in the order of their pairs (c,d) (index C(d-b-1,2) + c-b-1), one
// contiguous stretch [plo[b], plo[b] + pcnt[b]); a round = `waves` consecutive b x one group of 64 pairs. Rounds are listed
// from the largest b down (longest first). parts: the at most two partial rows at the ends of the range.
void plan_bundles(uint32_t n, uint64_t r0, uint64_t r1, uint32_t waves, BundlePlan &out) {
    const uint32_t kBundleWaves = waves;
    out.plo.assign(n, 0); out.pcnt.assign(n, 0); out.rounds.clear(); out.n_parts = 0;
    const uint64_t total = binom4(n);
    if (r1 > total) r1 = total;
    if (r0 >= r1 || n < 4) return;
    uint32_t a0, b0, c0, d0, a1 = 0, b1 = 0, c1 = 0, d1 = 0;
    unrank4(r0, a0, b0, c0, d0);
    const bool open_end = r1 == total;
    if (!open_end) unrank4(r1, a1, b1, c1, d1);
    auto idx = [](uint32_t b, uint32_t c, uint32_t d) -> uint64_t {   // pairs (c',d') of b before (c,d)
        const uint64_t dd = d > b + 1 ? d - b - 1 : 0, cc = c > b + 1 ? c - b - 1 : 0;
        return dd * (dd > 0 ? dd - 1 : 0) / 2 + (dd > 0 ? std::min(cc, dd) : 0);
    };
    for (uint32_t b = 1; b + 2 < n; ++b) {
        const uint64_t tp = binom2((uint64_t)n - 1 - b);
        uint64_t lo = idx(b, c0, d0);
        if (b < c0) { const bool inside = b > b0 || (b == b0 && a0 == 0); if (!inside) lo += 1; }
        uint64_t hi = tp;
        if (!open_end) { hi = idx(b, c1, d1); if (b < c1 && b < b1) hi += 1; }
        if (hi > tp) hi = tp;
        if (hi > lo) { out.plo[b] = (uint32_t)lo; out.pcnt[b] = (uint32_t)(hi - lo); }
    }
    const uint32_t blocks = (n + kBundleWaves - 1) / kBundleWaves;
    for (uint32_t k = blocks; k-- > 0;) {
        uint32_t groups = 0;
        for (uint32_t b = k * kBundleWaves; b < std::min(n, (k + 1) * kBundleWaves); ++b) groups = std::max(groups, (out.pcnt[b] + kWave - 1) / kWave);
        for (uint32_t g = 0; g < groups; ++g) { out.rounds.push_back(k); out.rounds.push_back(g); }
    }
    // partial rows: [r0, end of r0's row) if r0 is not a row start; [start of r1's row, r1) if r1 is not one
    uint64_t pa_lo = 0, pa_hi = 0, pb_lo = 0, pb_hi = 0;
    if (a0 > 0) { pa_lo = r0; pa_hi = std::min(r1, r0 + (b0 - a0)); }
    if (!open_end && a1 > 0) { pb_lo = std::max(r0, r1 - a1); pb_hi = r1; }
    if (pa_hi > pa_lo && pb_hi > pb_lo && pb_lo < pa_hi) { pa_hi = std::max(pa_hi, pb_hi); pb_lo = pb_hi = 0; }   // the same row
    if (pa_hi > pa_lo) { out.part_lo[out.n_parts] = pa_lo; out.part_n[out.n_parts++] = pa_hi - pa_lo; }
    if (pb_hi > pb_lo) { out.part_lo[out.n_parts] = pb_lo; out.part_n[out.n_parts++] = pb_hi - pb_lo; }
}

constexpr int kP1Iters = 16;      // passes of 256 ranks per workgroup (raw QIC)

template <typename CT>
__global__ __launch_bounds__(256) void raw_qic_kernel(ScoreDevice sd, uint64_t r0, uint64_t nq, uint8_t *__restrict__ topo,
                                                      unsigned long long *__restrict__ qout) {
    const uint64_t base = (uint64_t)blockIdx.x * (256ull * kP1Iters);
    if (base >= nq) return;
    Ids4 base_ids;
    unrank4(r0 + base + sd.rank_lo, base_ids.a, base_ids.b, base_ids.c, base_ids.d);
    for (int it = 0; it < kP1Iters; ++it) {
        const uint64_t i = base + (uint64_t)it * 256 + threadIdx.x;
        if (i >= nq) return;
        const QuartetRef q = classify<CT>(sd, r0 + i, decode_near(base_ids, (uint32_t)it * 256 + threadIdx.x));
        topo[i] = q.topo;
        qout[3 * i] = q.q1; qout[3 * i + 1] = q.q2; qout[3 * i + 2] = q.q3;
    }
}

template <typename CT, int PASS> static hipError_t launch_scan(hipStream_t s, const ScoreDevice &sd, double tol) {
    if (sd.n_tuples == 0) return hipSuccess;
    const uint64_t rounds = (sd.n_tuples + kSRound - 1) / kSRound;
    // as many rounds per workgroup as still leave >= 2048 workgroups (a workgroup's first rank is un-ranked once)
    const uint32_t rpw = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, rounds / 2048));
    const size_t lds = (PASS == 1 ? sizeof(ScanLds) : 0) + (size_t)sd.lds_n * 8;
    dim3 block(kSThreads), grid((unsigned)((rounds + rpw - 1) / rpw));
    auto k = score_scan_kernel<CT, PASS>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, s, sd, rpw, tol);
    return hipGetLastError();
}

// bundle kernel over the planned rounds (sd.bundle_*), scan kernel over the partial rows at the ends of the range
template <typename CT, int PASS> static hipError_t launch_bundle(hipStream_t s, const ScoreDevice &sd, double tol, int n_cu,
                                                                 const uint64_t *part_lo, const uint64_t *part_n, int n_parts) {
    constexpr int WAVES = PASS == 1 ? kBundleWaves1 : kBundleWaves2, WC = kBundleWavesCoop;
    if (sd.n_rounds > 0) {
        const bool coop = sd.coop_load == 1 || sd.coop_load == 3;
        const size_t lds = (coop ? bundle_lds_fixed<PASS, WC, true>() : bundle_lds_fixed<PASS, WAVES, false>()) + (size_t)sd.lds_n * 8;
        if (lds > 160u * 1024u) return hipErrorInvalidValue;   // (lds_n is capped by score_scan_max_lds_log(coop))
        auto k = sd.coop_load == 1 ? score_bundle_kernel<CT, PASS, WC, 1> : sd.coop_load == 3 ? score_bundle_kernel<CT, PASS, WC, 3>
                 : (sd.coop_load == 2 ? score_bundle_kernel<CT, PASS, WAVES, 2> : score_bundle_kernel<CT, PASS, WAVES, 0>);
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        dim3 block((coop ? WC : WAVES) * kWave), grid(std::min<uint32_t>(sd.n_rounds, (uint32_t)std::max(1, n_cu)));
        hipLaunchKernelGGL(k, grid, block, lds, s, sd, tol);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    for (int i = 0; i < n_parts; ++i) {
        ScoreDevice part = sd;
        part.table = reinterpret_cast<const unsigned char *>(sd.table) + (part_lo[i] - sd.rank_lo) * 3 * sizeof(CT);
        part.rank_lo = part_lo[i]; part.n_tuples = part_n[i];
        hipError_t e = launch_scan<CT, PASS>(s, part, tol);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
uint32_t score_bundle_waves(int pass, bool coop_load) { return coop_load ? kBundleWavesCoop : (pass == 1 ? kBundleWaves1 : kBundleWaves2); }

// tol: only read when sd.list is set (pass 1 with the candidate log of the single-read scoring; bundle kernel, whole rows only)
hipError_t launch_score_pass1(hipStream_t s, const ScoreDevice &sd, int kernel, int n_cu, const uint64_t *part_lo, const uint64_t *part_n, int n_parts, double tol) {
    if (kernel == 1) return sd.count_bits == 32 ? launch_scan<uint32_t, 1>(s, sd, 0.0) : launch_scan<uint16_t, 1>(s, sd, 0.0);
    return sd.count_bits == 32 ? launch_bundle<uint32_t, 1>(s, sd, tol, n_cu, part_lo, part_n, n_parts)
                               : launch_bundle<uint16_t, 1>(s, sd, tol, n_cu, part_lo, part_n, n_parts);
}

hipError_t launch_score_pass2(hipStream_t s, const ScoreDevice &sd, double tol, int kernel, int n_cu, const uint64_t *part_lo, const uint64_t *part_n, int n_parts) {
    if (kernel == 1) return sd.count_bits == 32 ? launch_scan<uint32_t, 2>(s, sd, tol) : launch_scan<uint16_t, 2>(s, sd, tol);
    return sd.count_bits == 32 ? launch_bundle<uint32_t, 2>(s, sd, tol, n_cu, part_lo, part_n, n_parts)
                               : launch_bundle<uint16_t, 2>(s, sd, tol, n_cu, part_lo, part_n, n_parts);
}

// Single-read scoring, second step: every logged (node pair, count triple) whose device QIC is within tol of the pair's
// FINAL minimum goes into the pair's candidate slots exactly as pass 2 would have put it there (scan_candidate). The QIC
// is recomputed from the global log table (it may differ from pass 1's LDS-table value in the last bits: far inside tol).
__global__ __launch_bounds__(256) void score_log_kernel(ScoreDevice sd, double tol, unsigned long long n_rec) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const unsigned long long *rec = sd.list + 4 * i;
    const uint32_t key = (uint32_t)rec[0], q1 = (uint32_t)rec[1], q2 = (uint32_t)rec[2], q3 = (uint32_t)rec[3];
    if (key >= sd.n_inner * sd.n_inner) return;         // (never logged; a guard in front of the indexed reads)
    const bool swp = (rec[0] & kListSwap) != 0;         // root_swapped: the reference evaluates it in both orders
    const double mag = bundle_qic_slow(sd.logk, sd.tbl_n, q1, q2, q3);
    const double qic = q1 != max(max(q1, q2), q3) ? -mag : mag;
    if (qic <= sortable_to_f64(sd.pair_min[key]) + tol) scan_candidate(sd, key, q1, q2, q3, swp);
}
hipError_t launch_score_log(hipStream_t s, const ScoreDevice &sd, double tol, unsigned long long n_rec) {
    if (n_rec == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_rec + 255) / 256));
    hipLaunchKernelGGL(score_log_kernel, grid, block, 0, s, sd, tol, n_rec);
    return hipGetLastError();
}

hipError_t launch_score_overflow_list(hipStream_t s, const ScoreDevice &sd, double tol) {
    return sd.count_bits == 32 ? launch_scan<uint32_t, 3>(s, sd, tol) : launch_scan<uint16_t, 3>(s, sd, tol);
}

// ---- node pairs (root, v) of a reference tree with a degree-2 root --------
// processNodePair takes the two subtrees beside the path with next() / next().next() on the link cycle
// (QuartetScoreComputer.hpp:393-396). The root of a rooted Newick tree has only TWO links, so for a pair (root, v)
// next().next() is the link towards v itself: S1 = the leaves on the other side of the root, S2 = ALL leaves on v's
// side (v's own subtrees included), S3, S4 = v's two child subtrees. The reference then sums
// countQuartetOccurrences(a,b,c,d) over S1 x S2 x S3 x S4 -- an argument that occurs twice reads cells of the n^4 table
// that are never incremented, i.e. (0,0,0) -- and writes log_score of the sums to the QP-IC of the edge (root, v) if v is a
// child of the root and to the EQP-IC minimum of every edge on the path. (Its per-quartet LQ-IC update goes to the path
// between lca(b,v) and v, which the pair that really owns the quartet updates with the same value: LQ-IC is unaffected.)
// This kernel produces those sums: item = (v, a, b, c, d); a workgroup reduces its items of one v and adds them to
// pair_sums[key(root, v)]. Quartets outside this context's table shard / view contribute 0 (each rank adds its part).
struct RootPair { uint32_t s1_lo, s1_n, s2_lo, s2_n, s3_lo, s3_n, s4_lo, s4_n, key, pad; unsigned long long first; }; // first = index of its first item
template <typename CT>
__global__ __launch_bounds__(256) void root_pair_sums_kernel(ScoreDevice sd, const RootPair *__restrict__ pairs, uint32_t n_pairs,
                                                             unsigned long long total, uint32_t items_per_thread) {
    __shared__ unsigned long long red[3][4];
    const unsigned long long wg_first = (unsigned long long)blockIdx.x * 256ull * items_per_thread;
    if (wg_first >= total) return;
    // the pair that holds this workgroup's first item (uniform binary search); a workgroup that runs past the end of a
    // pair flushes and moves on
    uint32_t lo = 0, hi = n_pairs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pairs[mid].first <= wg_first) lo = mid; else hi = mid; }
    uint32_t pi = lo;
    const CT *table = reinterpret_cast<const CT *>(sd.table);
    unsigned long long item = wg_first + threadIdx.x;
    const unsigned long long wg_end = min(total, wg_first + 256ull * items_per_thread);
    while (wg_first < wg_end) {   // (loop over the pairs this workgroup touches; usually one)
        const RootPair P = pairs[pi];
        const unsigned long long pair_end = (pi + 1 < n_pairs) ? pairs[pi + 1].first : total;
        const unsigned long long stop = min(wg_end, pair_end);
        unsigned long long s1 = 0, s2 = 0, s3 = 0;
        // The sums do not depend on the order of the items, so the FASTEST digit of the item index is the set that holds the
        // quartet's smallest lookup id whenever there is one (ids are the reference's leaf order, every set is an id range):
        // S1 in front of v's side -> a (always the smallest: consecutive threads read consecutive tuples); S1 behind it ->
        // c (the smallest whenever b > c). With d fastest (round 2) every read was a scattered one: 90 ms at 512 taxa.
        // item = ((g3 * n2 + g2) * n1 + g1) * n0 + g0, decoded ONCE per thread and pair (three 64-bit divisions), then
        // advanced by the thread stride 256 in mixed radix (adds and compares).
        const bool front = P.s1_lo < P.s2_lo;
        const uint32_t n0 = front ? P.s1_n : P.s3_n, n1 = front ? P.s4_n : P.s2_n, n2 = front ? P.s3_n : P.s4_n;
        const uint32_t l0 = front ? P.s1_lo : P.s3_lo, l1 = front ? P.s4_lo : P.s2_lo, l2 = front ? P.s3_lo : P.s4_lo, l3 = front ? P.s2_lo : P.s1_lo;
        uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        if (item < stop) {
            unsigned long long r = item - P.first;
            g0 = (uint32_t)(r % n0); r /= n0;
            g1 = (uint32_t)(r % n1); r /= n1;
            g2 = (uint32_t)(r % n2); r /= n2;
            g3 = (uint32_t)r;
        }
        uint32_t st0, st1, st2, st3;                       // the digits of 256 in the same radix
        { uint32_t r = 256; st0 = r % n0; r /= n0; st1 = r % n1; r /= n1; st2 = r % n2; r /= n2; st3 = r; }
        for (; item < stop; item += 256) {
            const uint32_t y0 = l0 + g0, y1 = l1 + g1, y2 = l2 + g2, y3 = l3 + g3;
            const uint32_t a = front ? y0 : y3, b = front ? y3 : y1, c = front ? y2 : y0, d = front ? y1 : y2;
            {   // advance to this thread's next item
                g0 += st0; uint32_t cy = g0 >= n0; if (cy) g0 -= n0;
                g1 += st1 + cy; cy = g1 >= n1; if (cy) g1 -= n1;
                g2 += st2 + cy; cy = g2 >= n2; if (cy) g2 -= n2;
                g3 += st3 + cy;
            }
            if (b == c || b == d) continue;   // a repeated argument: (0,0,0)
            uint32_t lo1 = min(a, b), hi1 = max(a, b), lo2 = min(c, d), hi2 = max(c, d);
            uint32_t m0 = min(lo1, lo2), m3 = max(hi1, hi2);
            uint32_t x1 = max(lo1, lo2), x2 = min(hi1, hi2);
            uint32_t m1 = min(x1, x2), m2 = max(x1, x2);
            const uint64_t rank = rank4(m0, m1, m2, m3);
            if (rank < sd.rank_lo || rank - sd.rank_lo >= sd.n_tuples) continue;
            const uint64_t cell = (rank - sd.rank_lo) * 3;
            s1 += table[cell + slot_of_pairing(a, b, c, d)];
            s2 += table[cell + slot_of_pairing(a, c, b, d)];
            s3 += table[cell + slot_of_pairing(a, d, b, c)];
        }
        // workgroup reduction: waves by DPP-free shuffles, then LDS
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); s3 += __shfl_down(s3, off, 64);
        }
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; red[2][wave] = s3; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const unsigned long long v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
            if (v) atomicAdd(&sd.pair_sums[(size_t)P.key * 3 + threadIdx.x], v);
        }
        if (stop >= wg_end) break;
        ++pi;
    }
}

hipError_t launch_root_pair_sums(hipStream_t s, const ScoreDevice &sd, const void *pairs_dev, uint32_t n_pairs, uint64_t total) {
    if (total == 0 || n_pairs == 0) return hipSuccess;
    const uint32_t ipt = 64;
    dim3 block(256), grid((unsigned)((total + 256ull * ipt - 1) / (256ull * ipt)));
    if (sd.count_bits == 32) hipLaunchKernelGGL(root_pair_sums_kernel<uint32_t>, grid, block, 0, s, sd, (const RootPair *)pairs_dev, n_pairs, (unsigned long long)total, ipt);
    else hipLaunchKernelGGL(root_pair_sums_kernel<uint16_t>, grid, block, 0, s, sd, (const RootPair *)pairs_dev, n_pairs, (unsigned long long)total, ipt);
    return hipGetLastError();
}

// entries of the k log k table the kernels keep in LDS; with cooperative loads the staging areas take 48-60 KB of it
uint32_t score_scan_max_lds_log(bool coop_load) {
    constexpr size_t fixed = bundle_lds_fixed<1, kBundleWavesCoop, true>();   // (pass 2 has no hash: less)
    return coop_load ? (uint32_t)((160u * 1024u - fixed - 256u) / 8u) : kScanMaxLdsLog;
}

// The same per quartet, but item i = the i-th 4-subset in LEXICOGRAPHIC order of its sorted lookup ids (a outermost, d
// innermost): the order in which printRawQICScores walks the reference's Euler-tour leaves
// (QuartetScoreComputer.hpp:626-630). Lexicographic index i <-> rank r' = C(n,4)-1-i of the mirrored set
// {n-1-d, n-1-c, n-1-b, n-1-a} in the table's own (colexicographic) order, so the block's ids come from one un-ranking
// plus decode_near; the tuple is read at rank(a,b,c,d) (scattered reads: the dump is bound by its text output anyway).
template <typename CT>
__global__ __launch_bounds__(256) void raw_qic_lex_kernel(ScoreDevice sd, uint64_t i0, uint64_t nq, uint64_t total, uint8_t *__restrict__ topo,
                                                          unsigned long long *__restrict__ qout) {
    const uint64_t base = (uint64_t)blockIdx.x * (256ull * kP1Iters);
    if (base >= nq) return;
    const uint64_t len = min((uint64_t)256 * kP1Iters, nq - base);
    Ids4 low;   // mirrored ids of the block's LAST item (smallest mirrored rank)
    const uint64_t r_min = total - 1 - (i0 + base + len - 1);
    unrank4(r_min, low.a, low.b, low.c, low.d);
    for (int it = 0; it < kP1Iters; ++it) {
        const uint64_t t = (uint64_t)it * 256 + threadIdx.x;
        if (t >= len) return;
        const Ids4 m = decode_near(low, (uint32_t)(len - 1 - t));
        Ids4 ids;
        ids.a = sd.n - 1 - m.d; ids.b = sd.n - 1 - m.c; ids.c = sd.n - 1 - m.b; ids.d = sd.n - 1 - m.a;
        const QuartetRef q = classify<CT>(sd, rank4(ids.a, ids.b, ids.c, ids.d) - sd.rank_lo, ids);
        const uint64_t i = base + t;
        topo[i] = q.topo;
        qout[3 * i] = q.q1; qout[3 * i + 1] = q.q2; qout[3 * i + 2] = q.q3;
    }
}

hipError_t launch_raw_qic_lex(hipStream_t s, const ScoreDevice &sd, uint64_t i0, uint64_t nq, uint8_t *topo_dev,
                              unsigned long long *q_dev) {
    if (nq == 0) return hipSuccess;
    const uint64_t per_block = 256ull * kP1Iters, total = binom4(sd.n);
    dim3 block(256), grid((unsigned)((nq + per_block - 1) / per_block));
    if (sd.count_bits == 32) hipLaunchKernelGGL(raw_qic_lex_kernel<uint32_t>, grid, block, 0, s, sd, i0, nq, total, topo_dev, q_dev);
    else hipLaunchKernelGGL(raw_qic_lex_kernel<uint16_t>, grid, block, 0, s, sd, i0, nq, total, topo_dev, q_dev);
    return hipGetLastError();
}

hipError_t launch_raw_qic(hipStream_t s, const ScoreDevice &sd, uint64_t r0, uint64_t nq, uint8_t *topo_dev,
                          unsigned long long *q_dev) {
    if (nq == 0) return hipSuccess;
    const uint64_t per_block = 256ull * kP1Iters;
    dim3 block(256), grid((unsigned)((nq + per_block - 1) / per_block));
    if (sd.count_bits == 32) hipLaunchKernelGGL(raw_qic_kernel<uint32_t>, grid, block, 0, s, sd, r0, nq, topo_dev, q_dev);
    else hipLaunchKernelGGL(raw_qic_kernel<uint16_t>, grid, block, 0, s, sd, r0, nq, topo_dev, q_dev);
    return hipGetLastError();
}

} // namespace qs
