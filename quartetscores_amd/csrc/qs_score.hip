// qs_score.hip -- LQ-/QP-/EQP-IC reductions over the count table on gfx950.
//
// Replaces the C(n,4)-sized part of QuartetScoreComputer (QuartetScoreComputer.hpp):
//   processNodePair / computeQuartetScoresBifurcating   :379-508
//   computeQuartetScoresMultifurcating                  :513-593
//   the topology test of printRawQICScores              :636-672
//
// The reference walks node pairs (u,v) and enumerates S1xS2xS3xS4; every 4-set is visited
// exactly once overall. Here the walk is quartet-major: lane = table rank, so the table is
// read once with fully coalesced 12-byte tuples. Lookup ids are the reference tree's own
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

struct ScanLds {
    uint32_t key[kSSlots];
    unsigned long long sum[kSSlots * 3];
    long long mn[kSSlots];
};

__device__ __forceinline__ void scan_global_add(const ScoreDevice &sd, uint32_t key, unsigned long long s1, unsigned long long s2,
                                                unsigned long long s3, long long mn) {
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 0], s1);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 1], s2);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 2], s3);
    atomicMin(&sd.pair_min[key], mn);
}
__device__ __forceinline__ void scan_flush(ScanLds &h, const ScoreDevice &sd, uint32_t key, unsigned long long s1, unsigned long long s2,
                                           unsigned long long s3, long long mn) {
    uint32_t slot = (key * 2654435761u) >> 22; // 10 bits
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
        slot = (slot + 1) & (kSSlots - 1);
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
// pass 2: record the gcd-reduced triple of a near-minimal quartet (log_score(k q) is bit-identical to log_score(q))
__device__ __noinline__ void scan_candidate(const ScoreDevice &sd, uint32_t key, uint32_t q1, uint32_t q2, uint32_t q3) {
    uint32_t g = gcd_u32(gcd_u32(q1, q2), q3);
    if (g == 0) g = 1;
    const uint32_t a = q1 / g, b = q2 / g, c = q3 / g;
    unsigned long long *slots = sd.pair_cand + (size_t)key * kCand;
    if ((a | b | c) >> 21) {   // does not fit the packed slot: this node pair is finished by qs_score_overflow
        atomicOr(&sd.flags[0], 2u);
        __hip_atomic_store(&slots[kCand - 1], kCandOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned long long packed = ((unsigned long long)a << 42) | ((unsigned long long)b << 21) | c;
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
                    } else if (PASS == 2) { if (qic <= thr) scan_candidate(sd, seg.key, q1, q2, q3); }
                    else if (marked && qic <= thr) {   // pass 3: (key, q1, q2, q3) of every near-minimal quartet of a marked pair
                        const unsigned long long at = atomicAdd(sd.list_count, 1ull);
                        if (at < sd.list_cap) {
                            unsigned long long *e = sd.list + at * 4;
                            e[0] = seg.key; e[1] = q1; e[2] = q2; e[3] = q3;
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

hipError_t launch_score_pass1(hipStream_t s, const ScoreDevice &sd) {
    return sd.count_bits == 32 ? launch_scan<uint32_t, 1>(s, sd, 0.0) : launch_scan<uint16_t, 1>(s, sd, 0.0);
}

hipError_t launch_score_pass2(hipStream_t s, const ScoreDevice &sd, double tol) {
    return sd.count_bits == 32 ? launch_scan<uint32_t, 2>(s, sd, tol) : launch_scan<uint16_t, 2>(s, sd, tol);
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
        for (; item < stop; item += 256) {
            unsigned long long r = item - P.first;
            const uint32_t d = P.s4_lo + (uint32_t)(r % P.s4_n); r /= P.s4_n;
            const uint32_t c = P.s3_lo + (uint32_t)(r % P.s3_n); r /= P.s3_n;
            const uint32_t b = P.s2_lo + (uint32_t)(r % P.s2_n); r /= P.s2_n;
            const uint32_t a = P.s1_lo + (uint32_t)r;
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

uint32_t score_scan_max_lds_log() { return kScanMaxLdsLog; }

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
