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

// QuartetScoreComputer.hpp:135-159 (device evaluation; only used to ORDER candidates -- the host re-evaluates the
// near-minimal ones with libm). With s = q1+q2+q3:  sum_i (q_i/s) log(q_i/s) = (sum_i q_i log q_i) / s - log s,
// and every argument is a small integer (<= number of trees), so log k and 1/k come from tables built on the host:
// 4 lookups + a handful of f64 ops instead of 3 divisions + 3 logs. Values beyond the tables take the slow path.
__device__ __forceinline__ double dev_logk(const ScoreDevice &sd, uint32_t k) {
    return k < sd.tbl_n ? sd.logk[k] : log((double)k);
}
__device__ __forceinline__ double dev_log_score(const ScoreDevice &sd, uint32_t q1, uint32_t q2, uint32_t q3) {
    if ((q1 | q2 | q3) == 0) return 0.0;
    const uint64_t s64 = (uint64_t)q1 + q2 + q3;
    const double inv_log3 = 0.91023922662683739361;
    double acc = 0.0; // sum q_i log q_i (0 log 0 = 0: logk[0] is stored as 0)
    acc += (double)q1 * dev_logk(sd, q1);
    acc += (double)q2 * dev_logk(sd, q2);
    acc += (double)q3 * dev_logk(sd, q3);
    double inv_s, log_s;
    if (s64 < sd.tbl_n) { inv_s = sd.invk[s64]; log_s = sd.logk[s64]; }
    else { inv_s = 1.0 / (double)s64; log_s = log((double)s64); }
    const double qic = 1.0 + (acc * inv_s - log_s) * inv_log3;
    return (q1 < q2 || q1 < q3) ? -qic : qic;
}

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

// ---- pass 1 -----------------------------------------------------------------------------
// A workgroup walks kP1Iters * 256 consecutive ranks. Consecutive ranks vary the smallest id a,
// and along a the owning node pair is piecewise constant, so (1) each wave first reduces its
// runs of equal keys with a segmented scan (shuffles), (2) the run tails add into a small
// open-addressing hash table in LDS (ds atomics), (3) the table is flushed once with global
// atomics. This cuts the global atomics from 4 per quartet to 4 per distinct pair per workgroup.
constexpr int kP1Iters = 16;      // passes of 256 ranks per workgroup (pass 2, raw QIC; lower bound for pass 1)
constexpr int kP1ItersMax = 64;   // pass 1 on large tables: 4x fewer hash flushes (global atomics) per quartet
constexpr int kP1Slots = 1024; // power of two
constexpr uint32_t kKeyEmpty = 0xFFFFFFFFu;

struct P1Lds {
    uint32_t key[kP1Slots];
    unsigned long long sum[kP1Slots * 3];
    long long mn[kP1Slots];
};

__device__ __forceinline__ void p1_global_add(const ScoreDevice &sd, uint32_t key, unsigned long long s1,
                                              unsigned long long s2, unsigned long long s3, long long mn) {
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 0], s1);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 1], s2);
    atomicAdd(&sd.pair_sums[(size_t)key * 3 + 2], s3);
    atomicMin(&sd.pair_min[key], mn);
}

// lane l <- lane l - k of the same 16-lane row (DPP row_shr:k, CTRL = 0x110 + k); lanes without a source get `old`
template <int CTRL> __device__ __forceinline__ uint32_t dpp_shr(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_shr64(unsigned long long v) {
    const uint32_t lo = dpp_shr<CTRL>(0u, (uint32_t)v), hi = dpp_shr<CTRL>(0u, (uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ void seg_step(uint32_t run, unsigned long long &s1, unsigned long long &s2, unsigned long long &s3,
                                         long long &mn) {
    const uint32_t orun = dpp_shr<CTRL>(0xFFFFFFFFu, run);
    const unsigned long long o1 = dpp_shr64<CTRL>(s1), o2 = dpp_shr64<CTRL>(s2), o3 = dpp_shr64<CTRL>(s3);
    const long long om = (long long)dpp_shr64<CTRL>((unsigned long long)mn);
    if (orun == run) { s1 += o1; s2 += o2; s3 += o3; mn = om < mn ? om : mn; }
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
template <int SRC>
__device__ __forceinline__ void seg_carry(uint32_t lane, uint32_t run, unsigned long long &s1, unsigned long long &s2,
                                          unsigned long long &s3, long long &mn) {
    const uint32_t crun = (uint32_t)__builtin_amdgcn_readlane((int)run, SRC);
    const unsigned long long c1 = readlane64(s1, SRC), c2 = readlane64(s2, SRC), c3 = readlane64(s3, SRC);
    const long long cm = (long long)readlane64((unsigned long long)mn, SRC);
    if (lane > (uint32_t)SRC && lane <= (uint32_t)SRC + 16 && run == crun) { s1 += c1; s2 += c2; s3 += c3; mn = cm < mn ? cm : mn; }
}

template <typename CT>
__global__ __launch_bounds__(256) void score_pass1_kernel(ScoreDevice sd, int iters) {
    __shared__ P1Lds lds;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < kP1Slots; i += 256) {
        lds.key[i] = kKeyEmpty;
        lds.sum[3 * i] = lds.sum[3 * i + 1] = lds.sum[3 * i + 2] = 0;
        lds.mn[i] = kSortableMax;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (256ull * iters);
    Ids4 base_ids;
    unrank4(base + sd.rank_lo, base_ids.a, base_ids.b, base_ids.c, base_ids.d);
    for (int it = 0; it < iters; ++it) {
        const uint64_t r = base + (uint64_t)it * 256 + tid;
        QuartetRef q;
        q.resolved = false; q.key = kKeyEmpty; q.q1 = q.q2 = q.q3 = 0;
        if (r < sd.n_tuples) q = classify<CT>(sd, r, decode_near(base_ids, (uint32_t)it * 256 + tid));
        const uint32_t key = q.resolved ? q.key : kKeyEmpty;
        unsigned long long s1 = q.q1, s2 = q.q2, s3 = q.q3;
        long long mn = q.resolved ? f64_to_sortable(dev_log_score(sd, q.q1, q.q2, q.q3)) : kSortableMax;
        // runs of equal keys inside the wave
        const uint32_t prev = __shfl_up(key, 1, 64);
        const bool head = (lane == 0) || (prev != key);
        const unsigned long long heads = __ballot(head);
        const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull));
        // segmented inclusive scan over the runs: inside each row of 16 lanes with DPP row shifts (VALU, no LDS
        // traffic; the ds_bpermute version of this scan was what bounded the kernel), then the last lane of each
        // row is carried into the lanes of the next row that continue its run (3 readlane steps)
        seg_step<0x111>(run, s1, s2, s3, mn);
        seg_step<0x112>(run, s1, s2, s3, mn);
        seg_step<0x114>(run, s1, s2, s3, mn);
        seg_step<0x118>(run, s1, s2, s3, mn);
        seg_carry<15>(lane, run, s1, s2, s3, mn);
        seg_carry<31>(lane, run, s1, s2, s3, mn);
        seg_carry<47>(lane, run, s1, s2, s3, mn);
        const bool tail = (lane == 63) || ((heads >> (lane + 1)) & 1ull);
        if (tail && key != kKeyEmpty) {
            uint32_t slot = (key * 2654435761u) >> 22; // 10 bits
            bool done = false;
            for (int probe = 0; probe < 32 && !done; ++probe) {
                const uint32_t old = atomicCAS(&lds.key[slot], kKeyEmpty, key);
                if (old == kKeyEmpty || old == key) {
                    atomicAdd(&lds.sum[3 * slot + 0], s1);
                    atomicAdd(&lds.sum[3 * slot + 1], s2);
                    atomicAdd(&lds.sum[3 * slot + 2], s3);
                    atomicMin(&lds.mn[slot], mn);
                    done = true;
                }
                slot = (slot + 1) & (kP1Slots - 1);
            }
            if (!done) p1_global_add(sd, key, s1, s2, s3, mn); // table crowded: go straight to memory
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < kP1Slots; i += 256)
        if (lds.key[i] != kKeyEmpty) p1_global_add(sd, lds.key[i], lds.sum[3 * i], lds.sum[3 * i + 1], lds.sum[3 * i + 2], lds.mn[i]);
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

template <typename CT>
__global__ __launch_bounds__(256) void score_pass2_kernel(ScoreDevice sd, double tol) {
    const uint64_t base = (uint64_t)blockIdx.x * (256ull * kP1Iters);
    Ids4 base_ids;
    unrank4(base + sd.rank_lo, base_ids.a, base_ids.b, base_ids.c, base_ids.d);
    for (int it = 0; it < kP1Iters; ++it) {
        const uint64_t r = base + (uint64_t)it * 256 + threadIdx.x;
        if (r >= sd.n_tuples) return;
        const QuartetRef q = classify<CT>(sd, r, decode_near(base_ids, (uint32_t)it * 256 + threadIdx.x));
        if (!q.resolved) continue;
        const double sc = dev_log_score(sd, q.q1, q.q2, q.q3);
        const double mn = sortable_to_f64(sd.pair_min[q.key]);
        if (!(sc <= mn + tol)) continue;
        uint32_t g = gcd_u32(gcd_u32(q.q1, q.q2), q.q3);
        if (g == 0) g = 1;
        const uint32_t a = q.q1 / g, b = q.q2 / g, c = q.q3 / g;
        if ((a | b | c) >> 21) { atomicOr(&sd.flags[0], 2u); continue; }
        const unsigned long long packed = ((unsigned long long)a << 42) | ((unsigned long long)b << 21) | c;
        unsigned long long *slots = sd.pair_cand + (size_t)q.key * kCand;
        // cheap pre-check avoids hammering CAS when thousands of quartets share one triple
        bool placed = false;
        for (int s = 0; s < kCand && !placed; ++s) {
            unsigned long long cur = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == packed) placed = true;
            else if (cur == kCandEmpty) {
                unsigned long long old = atomicCAS(&slots[s], kCandEmpty, packed);
                if (old == kCandEmpty || old == packed) placed = true;
            }
        }
        if (!placed) atomicOr(&sd.flags[0], 1u);
    }
}

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

hipError_t launch_score_pass1(hipStream_t s, const ScoreDevice &sd) {
    if (sd.n_tuples == 0) return hipSuccess;
    // as many passes per workgroup as still leave >= 8192 workgroups (128 taxa: 16, from ~230 taxa on: 64)
    const int iters = (int)std::min<uint64_t>(kP1ItersMax, std::max<uint64_t>(kP1Iters, sd.n_tuples / (256ull * 8192)));
    const uint64_t per_block = 256ull * iters;
    dim3 block(256), grid((unsigned)((sd.n_tuples + per_block - 1) / per_block));
    if (sd.count_bits == 32) hipLaunchKernelGGL(score_pass1_kernel<uint32_t>, grid, block, 0, s, sd, iters);
    else hipLaunchKernelGGL(score_pass1_kernel<uint16_t>, grid, block, 0, s, sd, iters);
    return hipGetLastError();
}

hipError_t launch_score_pass2(hipStream_t s, const ScoreDevice &sd, double tol) {
    if (sd.n_tuples == 0) return hipSuccess;
    const uint64_t per_block = 256ull * kP1Iters;
    dim3 block(256), grid((unsigned)((sd.n_tuples + per_block - 1) / per_block));
    if (sd.count_bits == 32) hipLaunchKernelGGL(score_pass2_kernel<uint32_t>, grid, block, 0, s, sd, tol);
    else hipLaunchKernelGGL(score_pass2_kernel<uint16_t>, grid, block, 0, s, sd, tol);
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
