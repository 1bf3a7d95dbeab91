// qs_count.hip -- quartet-topology counting on gfx950 (MI355X).
//
// Replaces QuartetCounterLookup::countQuartets / updateQuartets / updateQuartetsThreeLinks /
// updateQuartetsThreeClades (QuartetCounterLookup.hpp:65-238) and the per-increment index
// arithmetic of QuartetLookupTable (quartet_lookup_table.hpp:87-111,141-212).
//
// Two formulations of the same count:
//
//  GATHER (default).  The reference walks every tree and does one random read-modify-write
//  per (tree, displayed quartet). Here the loop nest is turned inside out: every lane OWNS
//  table cells (a run of consecutive ranks) and the trees are streamed past it. A tree is
//  reduced once to its "pair-depth panel": for every taxon pair {x,y} the depth of their
//  lowest common ancestor (build_panel kernel, O(n^2) per tree). By the four-point
//  condition a tree displays ab|cd  iff  M[ab]+M[cd] > M[ac]+M[bd] (= M[ad]+M[bc]); all
//  three sums equal means the tree does not resolve the quartet. Counters live in registers
//  for the whole launch and each table cell is written exactly once, with plain coalesced
//  stores: no atomics, and HBM traffic is one pass over the table instead of m passes.
//  Two implementations of the comparison:
//    * bit-sliced (count_bitslice3_kernel, default): the panel holds the depths of 32 trees as
//      bit planes; a (B+1)-bit magnitude comparison of 32 trees costs 2(B+1) v_bitop3_b32;
//    * byte-SWAR (count_gather_kernel; depths beyond 7 bits, and the independent cross-check
//      of bench.py): 16 (u8) or 8 (u16) trees per 16-byte panel element, 4 or 2 trees per
//      32-bit integer instruction (the sums are kept below 128 / 32768 so that "(S1 | H) - S2"
//      never borrows across fields), v_bcnt accumulates the hits.
//
//  SCATTER.  The tree-major formulation of BASELINE.json's north_star: one wavefront per
//  (tree, inner node), the tree's tour staged in LDS, lanes enumerating the third clade while
//  the wave walks pairs of the first and members of the second, atomicAdd into the table.
//  Each displayed quartet is counted once (only when the pair side holds the 4-set's minimum
//  id; the reference counts it at both ends of its middle path, SURVEY.md 3.2 iii). Kept as
//  the literal restatement and as an independent cross-check of the gather path.
//
// Integer/byte work only: no MFMA.
#include "qs_common.hpp"
#include "qs_internal.hpp"
#include "qs_bitslice3.hpp"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace qs {

// ======================================================================================
// pair-depth panel
// ======================================================================================

constexpr int kPanelThreads = 256;
constexpr int kPanelPPT = 8; // pairs per thread
constexpr int kPanelPB = kPanelThreads * kPanelPPT;

// Panel layout: uint4 P[n_chunks][npairs]; element = TPC consecutive trees of one taxon pair
// (TPC = 16 for u8 depths, 8 for u16). pair index of x<y is C(y,2)+x.
template <typename DT, bool PARTIAL>
__global__ __launch_bounds__(kPanelThreads) void build_panel_kernel(const uint32_t *__restrict__ leaf_off,
                                                                    const uint32_t *__restrict__ order, uint32_t slot0,
                                                                    const uint16_t *__restrict__ leaf_ids,
                                                                    const uint16_t *__restrict__ adj_depth,
                                                                    uint32_t n_trees, uint32_t n, uint32_t npairs,
                                                                    uint32_t levels, uint4 *__restrict__ P) {
    constexpr int TPC = 16 / (int)sizeof(DT);
    constexpr uint32_t FLAG = sizeof(DT) == 1 ? 0x20u : 0x2000u;
    extern __shared__ __align__(16) unsigned char smem[];
    DT *out = reinterpret_cast<DT *>(smem);                                  // [kPanelPB][TPC]
    uint16_t *pos = reinterpret_cast<uint16_t *>(smem + (size_t)kPanelPB * 16); // [n]
    uint16_t *st = pos + n;                                                  // [levels][n]

    const uint32_t tid = threadIdx.x;
    const uint32_t tc = blockIdx.y;
    const uint32_t p0 = blockIdx.x * kPanelPB;

    uint32_t px[kPanelPPT], py[kPanelPPT];
#pragma unroll
    for (int q = 0; q < kPanelPPT; ++q) {
        uint32_t p = p0 + q * kPanelThreads + tid;
        if (p < npairs) unrank2(p, px[q], py[q]);
        else { px[q] = 0; py[q] = 1; }
    }

    for (int j = 0; j < TPC; ++j) {
        const uint32_t slot = tc * TPC + j;
        if (slot >= n_trees) { // uniform: padding trees resolve nothing (all sums equal)
#pragma unroll
            for (int q = 0; q < kPanelPPT; ++q) out[(q * kPanelThreads + tid) * TPC + j] = 0;
            continue;
        }
        const uint32_t t = order ? order[slot0 + slot] : slot0 + slot; // tree behind this slot of the sub-batch
        const uint32_t base = leaf_off[t];
        const uint32_t L = leaf_off[t + 1] - base;
        __syncthreads(); // previous tree's queries done
        for (uint32_t x = tid; x < n; x += kPanelThreads) pos[x] = 0xFFFFu;
        __syncthreads();
        for (uint32_t i = tid; i < L; i += kPanelThreads) {
            pos[leaf_ids[base + i]] = (uint16_t)i;
            st[i] = adj_depth[base + i];
        }
        __syncthreads();
        // sparse table over D[0 .. L-2]
        for (uint32_t k = 1; k < levels; ++k) {
            const uint32_t half = 1u << (k - 1), span = 1u << k;
            if (span + 1 <= L) {
                for (uint32_t i = tid; i + span <= L - 1; i += kPanelThreads)
                    st[k * n + i] = min(st[(k - 1) * n + i], st[(k - 1) * n + i + half]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < kPanelPPT; ++q) {
            uint32_t a = pos[px[q]], b = pos[py[q]];
            uint32_t val;
            if (PARTIAL && (a == 0xFFFFu || b == 0xFFFFu)) {
                val = FLAG;
            } else {
                uint32_t lo = min(a, b), hi = max(a, b);
                uint32_t len = hi - lo; // >= 1: D[lo .. hi-1]
                uint32_t k = 31u - (uint32_t)__clz((int)len);
                val = min(st[k * n + lo], st[k * n + hi - (1u << k)]);
            }
            out[(q * kPanelThreads + tid) * TPC + j] = (DT)val;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPanelPPT; ++q) {
        uint32_t p = p0 + q * kPanelThreads + tid;
        if (p < npairs) P[(size_t)tc * npairs + p] = reinterpret_cast<const uint4 *>(out)[q * kPanelThreads + tid];
    }
}

static uint32_t panel_levels(uint32_t n) {
    uint32_t lv = 1;
    while ((1u << lv) < n) ++lv;
    return lv + 1;
}

hipError_t launch_build_panel(hipStream_t s, const DeviceBatch &b, uint32_t n, int panel_bits, bool partial, void *panel,
                              uint32_t n_chunks) {
    const uint32_t npairs = (uint32_t)binom2(n);
    const uint32_t levels = panel_levels(n);
    const size_t lds = (size_t)kPanelPB * 16 + (size_t)n * 2 + (size_t)levels * n * 2;
    dim3 grid((npairs + kPanelPB - 1) / kPanelPB, n_chunks), block(kPanelThreads);
#define QS_PANEL(DT, PART)                                                                                      \
    do {                                                                                                        \
        auto k = build_panel_kernel<DT, PART>;                                                                  \
        if (lds > 48 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                      \
        }                                                                                                       \
        hipLaunchKernelGGL(k, grid, block, lds, s, b.leaf_off, b.tree_order, b.slot0, b.leaf_ids, b.adj_depth, b.n_trees, n, npairs, levels, \
                           (uint4 *)panel);                                                                     \
    } while (0)
    if (panel_bits == 8) { if (partial) QS_PANEL(uint8_t, true); else QS_PANEL(uint8_t, false); }
    else { if (partial) QS_PANEL(uint16_t, true); else QS_PANEL(uint16_t, false); }
#undef QS_PANEL
    return hipGetLastError();
}

// ======================================================================================
// gather count kernel
// ======================================================================================

template <int BITS> struct Swar;
template <> struct Swar<8> {
    static constexpr uint32_t H = 0x80808080u, ONES = 0x01010101u;
};
template <> struct Swar<16> {
    static constexpr uint32_t H = 0x80008000u, ONES = 0x00010001u;
};

// SWAR comparison of one 32-bit word = 4 (u8) or 2 (u16) trees of one quartet.
// With S1 = M[ab]+M[cd], S2 = M[ac]+M[bd], S3 = M[ad]+M[bc] (all fields < 128 / 32768):
//   x = (S1 + H) - S2   field top bit <=> S1 >= S2, field never 0
//   t = x - ONES        field top bit <=> S1 >  S2            -> topology ab|cd
//   ~x                  field top bit <=> S2 >  S1            -> topology ac|bd
//   w = (S1 + H) - S3   field top bit <=> S1 >= S3; S1 == S2 and S3 > S1 -> topology ad|bc
// The parts that do not depend on d are hoisted by the caller: k12 = ab + H - ac, k13 = ab + H - bc,
// so x = k12 + (cd - bd) and w = k13 + (cd - ad): two integer ops each per (d, word).

template <int BITS, int MODE>
__device__ __forceinline__ void swar_step(uint32_t k12, uint32_t k13, uint32_t ab, uint32_t cd, uint32_t bd, uint32_t ad,
                                          uint32_t &c0, uint32_t &c1, uint32_t &c2) {
    constexpr uint32_t H = Swar<BITS>::H, ONES = Swar<BITS>::ONES;
    const uint32_t x = k12 + (cd - bd);
    const uint32_t t = x - ONES;
    if (MODE == MODE_BINARY_FULL) {
        popc_acc(t & H, c0);
        popc_acc(~x & H, c1);
    } else {
        const uint32_t w = k13 + (cd - ad);
        uint32_t hv = H;
        if (MODE == MODE_PARTIAL) hv = ~((ab | cd) << 2) & H; // flag bit (0x20 / 0x2000) -> top bit
        popc_acc(t & hv, c0);
        popc_acc(~x & hv, c1);
        popc_acc(x & ~(t | w) & hv, c2);
    }
}

__device__ __forceinline__ uint32_t upper_bound_le(const uint32_t *__restrict__ arr, uint32_t lo, uint32_t hi, uint32_t key) {
    // largest i in [lo, hi) with arr[i] <= key (arr non-decreasing, arr[lo] <= key)
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (arr[mid] <= key) lo = mid; else hi = mid;
    }
    return lo;
}


// Work item = one WAVEFRONT = (d-block of kDB largest ids, third id c, tile of (a,b) with a<b<c).
// Ids below c are cut into blocks of 8. An off-diagonal tile is (a-block at) x (b-block bt), at < bt:
// lane (ia, ib) owns a = 8*at+ia, b = 8*bt+ib. A diagonal tile packs TWO diagonal blocks (2k, 2k+1):
// lanes 0..31 / 32..63 enumerate the 28 pairs a<b inside block 2k / 2k+1. A lane owns the kDB quartets
// {a,b,c,d0..d0+7}; ranks C(d,4)+C(c,3)+C(b,2)+a are consecutive along a (12-byte tuples).
// Per 16-tree chunk the wave stages only the panel elements of the pairs (x,c), (x,d) for the 16 ids x
// of its two blocks: (1+kDB)*16 = 144 elements = 2.3 KB, independent of n; (c,d) comes through scalar
// loads. Staging is double-buffered through registers: the global loads of chunk t+1 are issued before
// the compute of chunk t and written to the other LDS buffer after it. The four waves of a workgroup
// are independent (own tile, own LDS region; DS operations of one wave execute in order), so there is
// no workgroup barrier anywhere in the kernel.
constexpr int kCols = kTA + kTB;                        // staged ids per wave
constexpr int kRowElems = (1 + kDB) * kCols;            // 144 staged elements
constexpr int kStagePerLane = (kRowElems + kWave - 1) / kWave; // 3

template <int BITS, int MODE, typename CT>
__global__ __launch_bounds__(kCountThreads) void count_gather_kernel(const uint4 *__restrict__ P, uint32_t npairs,
                                                                     uint32_t n_chunks, uint32_t m_trees, uint32_t n,
                                                                     uint32_t d_lo, uint32_t d_hi, uint64_t rank_lo,
                                                                     uint32_t n_dblk, uint32_t total_tiles,
                                                                     const uint32_t *__restrict__ dprefix,
                                                                     const uint32_t *__restrict__ cprefix,
                                                                     CT *__restrict__ table,
                                                                     uint32_t *__restrict__ overflow_flag, uint32_t overwrite) {
    __shared__ uint4 stage_all[kWavesPerBlock][2][kRowElems];

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const uint32_t tile = blockIdx.x * kWavesPerBlock + wave;
    if (tile >= total_tiles) return; // whole wave
    uint4(*stage)[kRowElems] = stage_all[wave];

    // ---- tile decode (wave-uniform) ----
    const uint32_t k = upper_bound_le(dprefix, 0, n_dblk, tile);
    const uint32_t local = tile - dprefix[k];
    const uint32_t d0 = d_lo + k * kDB;
    const uint32_t d1 = min(d0 + (uint32_t)kDB, d_hi);
    const uint32_t c = upper_bound_le(cprefix, 2, d1 - 1, local);
    const uint32_t T = (c + kTB - 1) / kTB, n_off = T * (T - 1) / 2;
    const uint32_t tl = local - cprefix[c];
    uint32_t a0, b0, a, b, colA, colB; // colA/colB: this lane's columns in the staged rows
    if (tl < n_off) {                  // off-diagonal tile (uniform branch)
        uint32_t at, bt;
        unrank2(tl, at, bt);           // at < bt
        a0 = at * kTA; b0 = bt * kTB;
        colA = lane & (kTA - 1); colB = kTA + lane / kTA;
        a = a0 + colA; b = b0 + (colB - kTA);
    } else {                           // two diagonal blocks
        const uint32_t kd = tl - n_off;
        a0 = (2 * kd) * kTA; b0 = (2 * kd + 1) * kTB;
        const uint32_t h = lane >> 5, q = lane & 31;
        uint32_t ia = 0, ib = 1;
        if (q < 28) unrank2(q, ia, ib); // ia < ib < 8
        colA = h * kTA + ia; colB = h * kTA + ib;
        a = (h ? b0 : a0) + ia; b = (h ? b0 : a0) + ib;
        if (q >= 28) b = 0xFFFFFFFFu;  // no pair
    }
    const bool lane_valid = (a < b) && (b < c);
    const uint32_t pi = lane_valid ? (uint32_t)binom2(b) + a : 0u; // pair index of (a,b)

    // staging map: element e = row * 16 + col <- panel pair (x, y): row 0: y = c, row 1+j: y = d0+j;
    // col < 8: x = a0+col, else x = b0+col-8. Elements no valid lane ever reads (x >= c, d outside the
    // block) load pair 0: the value is irrelevant and an unconditional load is cheaper than a select.
    uint32_t src[kStagePerLane];
#pragma unroll
    for (int s = 0; s < kStagePerLane; ++s) {
        const uint32_t e = lane + s * kWave;
        uint32_t p = 0u;
        if (e < (uint32_t)kRowElems) {
            const uint32_t row = e / kCols, col = e % kCols;
            const uint32_t x = col < (uint32_t)kTA ? a0 + col : b0 + (col - kTA);
            const uint32_t y = row == 0 ? c : d0 + (row - 1);
            if (x < c && y < d1 && (row == 0 || y > c)) p = (uint32_t)binom2(y) + x;
        }
        src[s] = p;
    }
    // (c, d0+j): the same element for every lane -> uniform index -> scalar loads, no LDS
    uint32_t cdi[kDB];
#pragma unroll
    for (int j = 0; j < kDB; ++j) {
        const uint32_t y = d0 + j;
        cdi[j] = (y < d1 && y > c) ? (uint32_t)binom2(y) + c : 0u;
    }
    // valid d slots are j in [jlo, jhi)
    const uint32_t jlo = c >= d0 ? c + 1 - d0 : 0u, jhi = d1 - d0;
    const bool all_slots = (jlo == 0) && (jhi == (uint32_t)kDB);

    uint32_t c0[kDB], c1[kDB], c2[kDB];
#pragma unroll
    for (int j = 0; j < kDB; ++j) c0[j] = c1[j] = c2[j] = 0;

    uint4 nxt[kStagePerLane];
#pragma unroll
    for (int s = 0; s < kStagePerLane; ++s) nxt[s] = make_uint4(0, 0, 0, 0);
    uint4 ab_next, cd_next[kDB];
    // prologue: chunk 0 -> buffer 0
#pragma unroll
    for (int j = 0; j < kDB; ++j) cd_next[j] = P[cdi[j]];
#pragma unroll
    for (int s = 0; s < kStagePerLane; ++s)
        if (lane + s * kWave < (uint32_t)kRowElems) nxt[s] = P[src[s]];
    ab_next = P[pi];
#pragma unroll
    for (int s = 0; s < kStagePerLane; ++s) {
        const uint32_t e = lane + s * kWave;
        if (e < (uint32_t)kRowElems) stage[0][e] = nxt[s];
    }

    for (uint32_t tc = 0; tc < n_chunks; ++tc) {
        const uint4 ab = ab_next;
        uint4 cdv[kDB];
#pragma unroll
        for (int j = 0; j < kDB; ++j) cdv[j] = cd_next[j];
        const uint4 *buf = stage[tc & 1];
        if (tc + 1 < n_chunks) { // issue the next chunk's loads; they complete under the compute below
            const uint4 *Pn = P + (size_t)(tc + 1) * npairs;
#pragma unroll
            for (int s = 0; s < kStagePerLane; ++s)
                if (lane + s * kWave < (uint32_t)kRowElems) nxt[s] = Pn[src[s]];
            ab_next = Pn[pi];
#pragma unroll
            for (int j = 0; j < kDB; ++j) cd_next[j] = Pn[cdi[j]];
        }
        const uint4 ac = buf[colA], bc = buf[colB];
        constexpr uint32_t H = Swar<BITS>::H;
        const uint4 k12 = make_uint4(ab.x + H - ac.x, ab.y + H - ac.y, ab.z + H - ac.z, ab.w + H - ac.w);
        const uint4 k13 = make_uint4(ab.x + H - bc.x, ab.y + H - bc.y, ab.z + H - bc.z, ab.w + H - bc.w);
#define QS_SLOT(j)                                                                                          \
    do {                                                                                                    \
        const uint4 bd = buf[(1 + (j)) * kCols + colB];                                                     \
        uint4 ad = bd;                                                                                      \
        if (MODE != MODE_BINARY_FULL) ad = buf[(1 + (j)) * kCols + colA];                                   \
        const uint4 cd = cdv[j];                                                                            \
        swar_step<BITS, MODE>(k12.x, k13.x, ab.x, cd.x, bd.x, ad.x, c0[j], c1[j], c2[j]);                   \
        swar_step<BITS, MODE>(k12.y, k13.y, ab.y, cd.y, bd.y, ad.y, c0[j], c1[j], c2[j]);                   \
        swar_step<BITS, MODE>(k12.z, k13.z, ab.z, cd.z, bd.z, ad.z, c0[j], c1[j], c2[j]);                   \
        swar_step<BITS, MODE>(k12.w, k13.w, ab.w, cd.w, bd.w, ad.w, c0[j], c1[j], c2[j]);                   \
    } while (0)
        if (all_slots) { // straight-line code for the common case
#pragma unroll
            for (int j = 0; j < kDB; ++j) QS_SLOT(j);
        } else {         // c inside the d-block, or the last block of the shard: skip the empty slots
#pragma unroll
            for (int j = 0; j < kDB; ++j)
                if ((uint32_t)j >= jlo && (uint32_t)j < jhi) QS_SLOT(j);
        }
#undef QS_SLOT
        if (tc + 1 < n_chunks) {
#pragma unroll
            for (int s = 0; s < kStagePerLane; ++s) {
                const uint32_t e = lane + s * kWave;
                if (e < (uint32_t)kRowElems) stage[(tc + 1) & 1][e] = nxt[s];
            }
        }
    }

    if (!lane_valid) return;
    const uint64_t rc = binom3(c) + pi;
#pragma unroll
    for (int j = 0; j < kDB; ++j) {
        const uint32_t d = d0 + j;
        if (d < d1 && d > c) {
            const uint64_t idx = (binom4(d) + rc - rank_lo) * 3;
            uint32_t n0 = c0[j], n1 = c1[j], n2 = (MODE == MODE_BINARY_FULL) ? (m_trees - c0[j] - c1[j]) : c2[j];
            uint32_t v0 = n0, v1 = n1, v2 = n2;
            if (!overwrite) { v0 += (uint32_t)table[idx]; v1 += (uint32_t)table[idx + 1]; v2 += (uint32_t)table[idx + 2]; }
            if (sizeof(CT) == 2 && ((v0 | v1 | v2) > 0xFFFFu)) atomicOr(overflow_flag, 1u);
            table[idx] = (CT)v0;
            table[idx + 1] = (CT)v1;
            table[idx + 2] = (CT)v2;
        }
    }
}

size_t gather_lds_bytes(uint32_t) { return sizeof(uint4) * 2 * kRowElems * kWavesPerBlock; }

uint32_t gather_tiles_for_c(uint32_t c) {
    const uint32_t T = (c + kTB - 1) / kTB;
    return T * (T - 1) / 2 + (T + 1) / 2; // off-diagonal tiles + pairs of diagonal blocks
}

hipError_t launch_count_gather(hipStream_t s, const CountGeometry &g, const void *panel, int panel_bits, int mode,
                               uint32_t n_chunks, uint32_t m_trees, void *table, int count_bits, uint32_t *overflow_flag,
                               bool overwrite) {
    if (g.total_tiles == 0) return hipSuccess;
    const uint32_t npairs = (uint32_t)binom2(g.n);
    dim3 grid((g.total_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kCountThreads);
#define QS_GATHER(B, M, CT)                                                                                         \
    hipLaunchKernelGGL((count_gather_kernel<B, M, CT>), grid, block, 0, s, (const uint4 *)panel, npairs, n_chunks,  \
                       m_trees, g.n, g.d_lo, g.d_hi, g.rank_lo, g.n_dblk, g.total_tiles, g.dprefix, g.cprefix, (CT *)table,     \
                       overflow_flag, overwrite ? 1u : 0u)
#define QS_GATHER_M(B, CT)                                                                                          \
    do {                                                                                                            \
        if (mode == MODE_BINARY_FULL) QS_GATHER(B, MODE_BINARY_FULL, CT);                                           \
        else if (mode == MODE_GENERAL_FULL) QS_GATHER(B, MODE_GENERAL_FULL, CT);                                    \
        else QS_GATHER(B, MODE_PARTIAL, CT);                                                                        \
    } while (0)
    if (panel_bits == 8) {
        if (count_bits == 32) QS_GATHER_M(8, uint32_t); else QS_GATHER_M(8, uint16_t);
    } else {
        if (count_bits == 32) QS_GATHER_M(16, uint32_t); else QS_GATHER_M(16, uint16_t);
    }
#undef QS_GATHER_M
#undef QS_GATHER
    return hipGetLastError();
}

// ======================================================================================
// bit-sliced gather path (default when LCA depths fit 7 bits)
// ======================================================================================
//
// The SWAR kernel above is bound by VALU issue: 7 integer ops per (quartet, 4 trees), two of them
// half-rate popcounts (tools/valu_rates.hip). Bit-slicing removes most of that work. The panel is
// transposed into bit planes: one element = the planes of one taxon pair for 32 trees (word k, bit t =
// bit k of the LCA depth in tree t; partial batches append a word "pair present in tree t"), stored
// compactly: only the B planes the batch needs exist in memory. With
//     L = M[ab] - M[ac] + 2^B   (per lane and c, once per 32 trees)
//     R = M[bd] - M[cd] + 2^B   (per staged row element, computed by the staging lanes once per wave)
// the topology test  ab|cd <=> M[ab]+M[cd] > M[ac]+M[bd] <=> L > R  is a (B+1)-bit magnitude
// comparison done for 32 trees at once with two 3-input boolean ops per bit (v_bitop3_b32, full rate):
//     gt = (l & ~r) | (~(l ^ r) & gt)      lt = (~l & r) | (~(l ^ r) & lt)        LSB -> MSB
// so one (quartet, 32 trees) step costs 2(B+1) boolean ops + 2 popcounts instead of 8 x 7 ops.
// The third topology of the general modes is the same comparison on  M[ad]-M[cd]  vs  M[ab]-M[bc].


// Bit-plane panel, general builder (any n). Workgroup = (group of 32 trees, 4096 pairs), 16 waves. The 32 trees are
// taken in rounds of as many trees as fit in LDS together (8 at 1024 taxa, 16 at 512): every wave builds the leaf
// positions and the sparse table (range minimum over adj_depth, u8: depths are < 128 whenever the bit-plane panel is
// used) of one tree of the round in its own LDS region -- DS operations of one wave execute in order, so no barrier is
// needed inside a build --, one barrier, then thread = 4 pairs: it answers its pairs' queries in the round's trees (4 LDS
// reads + min each) and keeps the planes in registers until all rounds are done. The tables are built once per 4096
// pairs; the round-1 builder rebuilt them for every 256 pairs (44 table entries written per query answered at 1024 taxa)
// and ran one workgroup of 4 waves per CU: 105 ms for 1024 taxa x 5000 trees, 21 % of a table-sharded step.
constexpr int kBPThreads = 1024;
constexpr int kBPPPT = 4;      // pairs per thread
constexpr int kBPPB = kBPThreads * kBPPPT;
template <bool PARTIAL, int NWC, typename DT>
__global__ __launch_bounds__(kBPThreads) void build_bitpanel_kernel(const uint32_t *__restrict__ leaf_off,
                                                                    const uint32_t *__restrict__ order, uint32_t slot0,
                                                                    const uint16_t *__restrict__ leaf_ids,
                                                                    const uint16_t *__restrict__ adj_depth,
                                                                    uint32_t n_trees, uint32_t n, uint32_t npairs,
                                                                    uint32_t levels, uint32_t tree_bytes, uint32_t per_round,
                                                                    uint4 *__restrict__ Pb) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const uint32_t g = blockIdx.y, p0 = blockIdx.x * kBPPB;
    constexpr int kWaves = kBPThreads / kWave;
    constexpr int kPlanes = PARTIAL ? NWC - 1 : NWC; // partial elements end with the presence word

    uint32_t px[kBPPPT], py[kBPPPT];
    uint32_t w[kBPPPT][kBitWords];
#pragma unroll
    for (int q = 0; q < kBPPPT; ++q) {
        const uint32_t p = p0 + q * kBPThreads + tid;
        px[q] = 0; py[q] = 1;
        if (p < npairs) unrank2(p, px[q], py[q]);
#pragma unroll
        for (int k = 0; k < kBitWords; ++k) w[q][k] = 0;
    }
    for (uint32_t r0 = 0; r0 < (uint32_t)kBitTrees; r0 += per_round) {
        __syncthreads(); // the previous round's queries are done
        for (uint32_t j = wave; j < per_round; j += kWaves) {
            const uint32_t slot = g * kBitTrees + r0 + j;
            if (slot >= n_trees) continue;
            const uint32_t t = order ? order[slot0 + slot] : slot0 + slot;
            uint16_t *pos = reinterpret_cast<uint16_t *>(smem + (size_t)j * tree_bytes); // [n]
            DT *st = reinterpret_cast<DT *>(pos + n);                                    // [levels][n]
            const uint32_t base = leaf_off[t], L = leaf_off[t + 1] - base;
            for (uint32_t x = lane; x < n; x += kWave) pos[x] = 0xFFFFu;
            for (uint32_t i = lane; i < L; i += kWave) {
                pos[leaf_ids[base + i]] = (uint16_t)i;
                // depths beyond the planes of this class are cut at the largest value they hold (trees placed in a class below
                // their own depth bits: depth clamp, clamp_fix_kernel); cut before the narrowing store, min() commutes with it
                st[i] = (DT)min((uint32_t)adj_depth[base + i], (1u << kPlanes) - 1u);
            }
            for (uint32_t k = 1; k < levels; ++k) {
                const uint32_t half = 1u << (k - 1), span = 1u << k;
                if (span + 1 <= L)
                    for (uint32_t i = lane; i + span <= L - 1; i += kWave)
                        st[k * n + i] = min(st[(k - 1) * n + i], st[(k - 1) * n + i + half]);
            }
        }
        __syncthreads();
        for (uint32_t j = 0; j < per_round; ++j) {
            const uint32_t bit = r0 + j, slot = g * kBitTrees + bit;
            if (slot >= n_trees) { // padding trees: depth 0 everywhere (resolve nothing), absent in partial mode
                if (!PARTIAL) {
#pragma unroll
                    for (int q = 0; q < kBPPPT; ++q) w[q][kPres] |= 1u << bit;
                }
                continue;
            }
            const uint16_t *pos = reinterpret_cast<const uint16_t *>(smem + (size_t)j * tree_bytes);
            const DT *st = reinterpret_cast<const DT *>(pos + n);
#pragma unroll
            for (int q = 0; q < kBPPPT; ++q) {
                const uint32_t a = pos[px[q]], b = pos[py[q]];
                if (PARTIAL && (a == 0xFFFFu || b == 0xFFFFu)) continue;
                const uint32_t lo = min(a, b), hi = max(a, b), len = hi - lo;
                const uint32_t k = 31u - (uint32_t)__clz((int)len);
                const uint32_t val = min((uint32_t)st[k * n + lo], (uint32_t)st[k * n + hi - (1u << k)]);
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) w[q][pl] |= ((val >> pl) & 1u) << bit;
                w[q][kPres] |= 1u << bit;
            }
        }
    }
    // compact layout: per tree group uint4 lo[npairs] (planes 0..3), then (NWC - 4) upper words per pair; partial
    // batches: the last word is the presence plane
    char *grp = reinterpret_cast<char *>(Pb) + (size_t)g * npairs * NWC * 4;
#pragma unroll
    for (int q = 0; q < kBPPPT; ++q) {
        const uint32_t p = p0 + q * kBPThreads + tid;
        if (p >= npairs) continue;
        if (PARTIAL) w[q][NWC - 1] = w[q][kPres];
        reinterpret_cast<uint4 *>(grp)[p] = make_uint4(w[q][0], w[q][1], w[q][2], w[q][3]);
        uint32_t *hi = reinterpret_cast<uint32_t *>(grp + (size_t)npairs * 16) + (size_t)p * (NWC > 4 ? NWC - 4 : 0);
#pragma unroll
        for (int k = 4; k < NWC; ++k) hi[k - 4] = w[q][k];
    }
}

// Small-n variant (the tables of all 32 trees of a group fit in LDS together, n <= ~256): workgroup = (group,
// 1024 pairs), 16 waves. Phase 1: every wave builds the leaf positions and the sparse table (u8: depths are
// < 128 whenever the bit-plane panel is used) of two trees in its own LDS region. Phase 2: thread = pair; it
// answers its pair's query in all 32 trees and assembles the planes in registers -- no byte staging, no
// transposition pass, and the tables are built 4x less often than with 256-pair workgroups. 128 taxa x 1000
// trees: 35 us -> see DESIGN.md.
constexpr int kBPSThreads = 1024;
template <bool PARTIAL, int NWC>
__global__ __launch_bounds__(kBPSThreads) void build_bitpanel_small_kernel(const uint32_t *__restrict__ leaf_off,
                                                                          const uint32_t *__restrict__ order, uint32_t slot0,
                                                                          const uint16_t *__restrict__ leaf_ids,
                                                                          const uint16_t *__restrict__ adj_depth,
                                                                          uint32_t n_trees, uint32_t n, uint32_t npairs,
                                                                          uint32_t levels, uint32_t tree_bytes,
                                                                          uint4 *__restrict__ Pb) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const uint32_t g = blockIdx.y, p0 = blockIdx.x * kBPSThreads;
    constexpr int kWaves = kBPSThreads / kWave;
    constexpr int kPlanes = PARTIAL ? NWC - 1 : NWC; // partial elements end with the presence word

    for (int j = (int)wave; j < kBitTrees; j += kWaves) {
        const uint32_t slot = g * kBitTrees + j;
        if (slot >= n_trees) continue;
        const uint32_t t = order ? order[slot0 + slot] : slot0 + slot; // tree behind this slot of the sub-batch
        uint16_t *pos = reinterpret_cast<uint16_t *>(smem + (size_t)j * tree_bytes); // [n]
        uint8_t *st = reinterpret_cast<uint8_t *>(pos + n);                          // [levels][n]
        const uint32_t base = leaf_off[t], L = leaf_off[t + 1] - base;
        for (uint32_t x = lane; x < n; x += kWave) pos[x] = 0xFFFFu;
        for (uint32_t i = lane; i < L; i += kWave) {
            pos[leaf_ids[base + i]] = (uint16_t)i;
            st[i] = (uint8_t)min((uint32_t)adj_depth[base + i], (1u << kPlanes) - 1u);   // (depth clamp, as in build_bitpanel_kernel)
        }
        for (uint32_t k = 1; k < levels; ++k) {
            const uint32_t half = 1u << (k - 1), span = 1u << k;
            if (span + 1 <= L)
                for (uint32_t i = lane; i + span <= L - 1; i += kWave)
                    st[k * n + i] = min(st[(k - 1) * n + i], st[(k - 1) * n + i + half]);
        }
    }
    __syncthreads();

    const uint32_t p = p0 + tid;
    if (p >= npairs) return;
    uint32_t x, y;
    unrank2(p, x, y);
    uint32_t w[kBitWords];
#pragma unroll
    for (int k = 0; k < kBitWords; ++k) w[k] = 0;
#pragma unroll
    for (int j = 0; j < kBitTrees; ++j) {
        const uint32_t t = g * kBitTrees + j;
        if (t >= n_trees) { // padding trees: depth 0 everywhere (resolve nothing), absent in partial mode
            if (!PARTIAL) w[kPres] |= 1u << j;
            continue;
        }
        const uint16_t *pos = reinterpret_cast<const uint16_t *>(smem + (size_t)j * tree_bytes);
        const uint8_t *st = reinterpret_cast<const uint8_t *>(pos + n);
        const uint32_t a = pos[x], b = pos[y];
        if (PARTIAL && (a == 0xFFFFu || b == 0xFFFFu)) continue;
        const uint32_t lo = min(a, b), hi = max(a, b), len = hi - lo;
        const uint32_t k = 31u - (uint32_t)__clz((int)len);
        const uint32_t val = min((uint32_t)st[k * n + lo], (uint32_t)st[k * n + hi - (1u << k)]);
#pragma unroll
        for (int q = 0; q < kPlanes; ++q) w[q] |= ((val >> q) & 1u) << j;
        w[kPres] |= 1u << j;
    }
    if (PARTIAL) w[NWC - 1] = w[kPres];
    char *grp = reinterpret_cast<char *>(Pb) + (size_t)g * npairs * NWC * 4;
    reinterpret_cast<uint4 *>(grp)[p] = make_uint4(w[0], w[1], w[2], w[3]);
    uint32_t *hi = reinterpret_cast<uint32_t *>(grp + (size_t)npairs * 16) + (size_t)p * (NWC > 4 ? NWC - 4 : 0);
#pragma unroll
    for (int k = 4; k < NWC; ++k) hi[k - 4] = w[k];
}

hipError_t launch_build_bitpanel(hipStream_t s, const DeviceBatch &b, uint32_t n, bool partial, void *panel,
                                 uint32_t n_groups, uint32_t compact_nw, bool force_general) {
    const uint32_t npairs = (uint32_t)binom2(n);
    const uint32_t levels = panel_levels(n);
    const uint32_t planes = compact_nw - (partial ? 1u : 0u);   // depth bits the panel carries
    if (compact_nw > (uint32_t)kMaxDepthBits + (partial ? 1u : 0u)) return hipErrorInvalidValue;
    if (planes <= 7 && compact_nw <= 7) {   // all 32 trees' tables (u8 depths) resident at once -> the small-n kernel
        const uint32_t tree_bytes = (n * 2 + levels * n + 3) & ~3u;
        const size_t lds_small = (size_t)kBitTrees * tree_bytes;
        if (lds_small <= 96 * 1024 && !force_general) { // force_general: qs_set_tuning(QS_TUNE_PANEL_KERNEL, 1) (tests / A-B runs)
            dim3 grid((npairs + kBPSThreads - 1) / kBPSThreads, n_groups), block(kBPSThreads);
#define QS_BPS(PART, NWC)                                                                                          \
    do {                                                                                                           \
        auto k = build_bitpanel_small_kernel<PART, NWC>;                                                           \
        if (lds_small > 48 * 1024) {                                                                               \
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small); \
            if (e != hipSuccess) return e;                                                                         \
        }                                                                                                          \
        hipLaunchKernelGGL(k, grid, block, lds_small, s, b.leaf_off, b.tree_order, b.slot0, b.leaf_ids, b.adj_depth, b.n_trees, n, npairs, levels, \
                           tree_bytes, (uint4 *)panel);                                                            \
    } while (0)
            if (partial) { // planes + presence word: 5..7
                if (compact_nw <= 5) QS_BPS(true, 5);
                else if (compact_nw == 6) QS_BPS(true, 6);
                else QS_BPS(true, 7);
            }
            else if (compact_nw <= 4) QS_BPS(false, 4);
            else if (compact_nw == 5) QS_BPS(false, 5);
            else if (compact_nw == 6) QS_BPS(false, 6);
            else QS_BPS(false, 7);
#undef QS_BPS
            return hipGetLastError();
        }
    }
    // general builder: as many trees per round as fit in 128 KB of LDS (a power of two, at most 16 = one per wave); the
    // range-minimum tables hold u8 depths up to 7 planes and u16 depths beyond (deep trees: 8..10 planes)
    const uint32_t dbytes = planes <= 7 ? 1u : 2u;
    const uint32_t tree_bytes = (n * 2 + levels * n * dbytes + 3) & ~3u;
    uint32_t per_round = 16;
    while (per_round > 1 && (size_t)per_round * tree_bytes > 128 * 1024) per_round >>= 1;
    const size_t lds = (size_t)per_round * tree_bytes;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((npairs + kBPPB - 1) / kBPPB, n_groups), block(kBPThreads);
#define QS_BPG(PART, NWC, DT)                                                                                      \
    do {                                                                                                           \
        auto k = build_bitpanel_kernel<PART, NWC, DT>;                                                             \
        if (lds > 48 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                         \
        }                                                                                                          \
        hipLaunchKernelGGL(k, grid, block, lds, s, b.leaf_off, b.tree_order, b.slot0, b.leaf_ids, b.adj_depth, b.n_trees, n, npairs, levels, \
                           tree_bytes, per_round, (uint4 *)panel);                                                 \
    } while (0)
    if (partial) { // planes + presence word: 5..11
        switch (compact_nw <= 5 ? 5u : compact_nw) {
            case 5: QS_BPG(true, 5, uint8_t); break;
            case 6: QS_BPG(true, 6, uint8_t); break;
            case 7: QS_BPG(true, 7, uint8_t); break;
            case 8: QS_BPG(true, 8, uint8_t); break;
            case 9: QS_BPG(true, 9, uint16_t); break;
            case 10: QS_BPG(true, 10, uint16_t); break;
            default: QS_BPG(true, 11, uint16_t); break;
        }
    } else {
        switch (compact_nw <= 4 ? 4u : compact_nw) {
            case 4: QS_BPG(false, 4, uint8_t); break;
            case 5: QS_BPG(false, 5, uint8_t); break;
            case 6: QS_BPG(false, 6, uint8_t); break;
            case 7: QS_BPG(false, 7, uint8_t); break;
            case 8: QS_BPG(false, 8, uint16_t); break;
            case 9: QS_BPG(false, 9, uint16_t); break;
            default: QS_BPG(false, 10, uint16_t); break;
        }
    }
#undef QS_BPG
    return hipGetLastError();
}


// ======================================================================================
// count_bitslice3_kernel: the bit-sliced count of every batch whose depths fit 7 bits (6 with missing taxa)
// ======================================================================================
// Wave = (d-block of 8, c, tile of (a,b)); lane = (a,b); the arithmetic of the section above.
//   binary_full instance (every tree binary and holding all taxa): two topologies are counted (the third is m minus
//   the other two) and a lane owns TWO a-columns: the tile is 16 a x 8 b, lane (ia, ib) owns a1 = 8*A1+ia,
//   a2 = 8*A2+ia and b = 8*Bk+ib (A1 < A2 < Bk are 8-blocks of ids below c; A2 is absent in the last tile of an
//   odd Bk). The staged R element of (b,d) is read from LDS once and compared against both L(a1,b,c) and L(a2,b,c).
//   general_full / partial instances: one a-column, tile 8 a x 8 b, three topologies counted, R staged for all 16
//   columns (the lane also reads R of (a,d)); partial batches carry a presence word per element.
//   Diagonal tiles (a and b from the same block) pack two diagonal blocks per wave, one a per lane.
// The step over 32 trees is built around what the ISA of its predecessor showed (60 % of the instructions were not
// comparison chains; profiles/r01_experiments.md):
//   * the panel is COMPACT: per tree group uint4 lo[npairs] (planes 0..3) followed by the B-4 upper planes per
//     pair, so a 16-byte load fetches only live data (build_bitpanel_kernel, compact_nw);
//   * panel reads are raw BUFFER loads: the per-lane offset inside one tree group is loop-invariant, the
//     group advances through the (scalar) base address of the resource, lanes with nothing to load use an
//     out-of-range offset and get zeros -- no 64-bit address arithmetic and no exec masking in the loop;
//   * staging is branch-free: lane l owns R element (d-row l/8, b-column l%8), lanes 0..15 also own the
//     M[x,c] element of a-column l;
//   * the step is unrolled twice with the two register sets / LDS buffers swapping roles, so the
//     double buffering costs no register moves;
//   * the step is instantiated for (second a-column present, all 8 d slots live, diagonal tile) and the wave
//     picks its instance once: the hot instance has no branches inside the step;
//   * d-blocks are aligned to the TOP of the shard (the partial block is the one with the smallest ids, where
//     c < d leaves few tiles): 7 % fewer tiles at 128 taxa, 81 % instead of 65 % of them with all slots live.
// LDS image per wave and buffer: slots 0..127 = R elements (d-row * 16 + b-column), 128..143 = M[x,c].
uint32_t bitslice3_tiles_for_c(uint32_t c) {
    const uint32_t T = (c + kTB - 1) / kTB;
    return (T * T) / 4 + (T + 1) / 2; // pairs of a-blocks below every b-block + pairs of diagonal blocks
}


// One class of trees per launch: tile decode, the class's groups past the tile (bs3_segment), the tuples to the table. The mixed
// batches' single launch over several classes is count_bitslice3_fused_kernel (qs_count_fused.hip).
template <int B, int MODE, typename CT>
__global__ __launch_bounds__(kCountThreads) __attribute__((amdgpu_waves_per_eu(bs3_waves<B, MODE>(), bs3_waves<B, MODE>()))) void count_bitslice3_kernel(const uint4 *__restrict__ P, uint32_t npairs,
                                                                        uint32_t n_groups, uint32_t m_trees,
                                                                        uint32_t d_start, uint32_t d_hi, uint64_t rank_lo,
                                                                        uint32_t n_dblk, uint32_t total_tiles,
                                                                        const uint32_t *__restrict__ dprefix,
                                                                        const uint32_t *__restrict__ cprefix,
                                                                        CT *__restrict__ table,
                                                                        uint32_t *__restrict__ overflow_flag, uint32_t overwrite,
                                                                        uint32_t xcd_remap, uint32_t *__restrict__ wire,
                                                                        const uint32_t *__restrict__ perm) {
    __shared__ uint4 stage_all[kWavesPerBlock][Bs3Layout<B, MODE>::kLdsUint4];
    Bs3Tile t;
    if (!bs3_decode_tile(t, d_start, d_hi, n_dblk, total_tiles, dprefix, cprefix, xcd_remap, perm)) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // counters per d slot (see bs3_segment)
    uint32_t x0[kDB], x1[kDB], y0[kDB], y1[kDB], z0[kDB], z1[kDB];
#pragma unroll
    for (int j = 0; j < kDB; ++j) x0[j] = x1[j] = y0[j] = y1[j] = z0[j] = z1[j] = 0;
    bs3_segment<B, MODE>(t, P, npairs, n_groups, xcd_remap, stage_all[wave], x0, x1, y0, y1, z0, z1);
    constexpr int THIRD = MODE == MODE_BINARY_FULL ? 0 : MODE == MODE_BINARY_PARTIAL ? 1 : 2;
    bs3_store<CT, THIRD>(t, rank_lo, table, overflow_flag, overwrite, m_trees, MODE == MODE_BINARY_FULL ? wire : nullptr, x0, x1, y0, y1, z0, z1);
}



// ======================================================================================
// count_bitslice4_kernel: binary_full batches, the waves of a workgroup SHARE their panel loads
// ======================================================================================
// What bounds count_bitslice3_kernel besides VALU issue is the vector-memory path, not the caches behind it: a 32-tree
// step of one wave issues 10 loads over 64 lanes (5 KB of 16-byte gathers plus their upper planes), every CU's L1 address
// path processes ~16 such waves per round, and a timing-only build whose buffer descriptor has zero records (every load
// answered without touching the L1) runs 12 % faster while changing the tile order for better L2 hits changes nothing
// (profiles/r03_experiments.md). This kernel cuts the requests themselves. The four waves of a workgroup take four
// tiles with the SAME a-blocks, b-block and d-block and consecutive third ids c (the host builds that launch order:
// tile_order in qs_abi.hip, groups padded with "shadow" tiles that compute but do not store). Of a wave's 216 panel
// elements per step only M[c,d] (8) and M[x,c] (16) depend on c; M[ab] (2 x 64) and M[bd] (64) are identical for the
// whole workgroup. So per step
//   wave w < 3 loads ONE of the three shared sets (one element per lane), every wave its own 24 private elements,
//   everything is written raw to LDS (double-buffered),
//   one s_barrier,
//   every lane reads back its own M[ab] (x2), M[bd], M[cd] and M[ac] (x2) elements and forms L1, L2 and R as before.
// 7 load pairs per workgroup-step instead of 20, 288 lane-elements instead of 1280; the raw elements of the next group
// wait in 10 VGPRs (not 25) during the compare chains. Same arithmetic, same table. Only off-diagonal tiles with both
// a-columns take this kernel (92 % of the tiles at 512 taxa); the rest goes through count_bitslice3_kernel.
// MEASURED (MI355X, 512 taxa x 10000 trees): 368 ms against 354 ms for count_bitslice3_kernel alone (349.8 ms when forced to
// 5 waves per SIMD, with spills) -- the barrier is now a true dependency between four waves on four SIMDs, and that costs
// more than the requests cost. The 12 % that the zero-record knock-out promised was mostly the clock rising on all-zero
// operands, not the memory path. Kept as an option (QS_TUNE_COOP = 1) and under test; off by default.
constexpr int kC4Shared = 3 * kWave;                    // raw slots of the three shared sets
constexpr int kC4Private = 24;                          // per wave: 16 M[x,c] + 8 M[c,d]
constexpr int kC4Raw = kC4Shared + kWavesPerBlock * kC4Private; // 288 raw slots per buffer
constexpr uint32_t kShadowTile = 0x80000000u;           // launch-slot flag: compute (loads, barriers) but do not store

template <int NWORDS> __device__ __forceinline__ void c4_put(uint4 *lo, uint32_t *hi, uint32_t slot, const Planes &x) {
    constexpr int H = NWORDS - 4;                       // upper words: 0..4, kept in records of 1, 2 or 4 words
    lo[slot] = make_uint4(x.w[0], x.w[1], x.w[2], x.w[3]);
    if (H == 1) hi[slot] = x.w[4];
    else if (H == 2) reinterpret_cast<uint2 *>(hi)[slot] = make_uint2(x.w[4], x.w[5]);
    else if (H >= 3) reinterpret_cast<uint4 *>(hi)[slot] = make_uint4(x.w[4], x.w[5], x.w[6], x.w[7]);
}
template <int NWORDS> __device__ __forceinline__ Planes c4_get(const uint4 *lo, const uint32_t *hi, uint32_t slot) {
    constexpr int H = NWORDS - 4;
    Planes r;
    const uint4 v = lo[slot];
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
#pragma unroll
    for (int k = 4; k < kBitWords; ++k) r.w[k] = 0;
    if (H == 1) r.w[4] = hi[slot];
    else if (H == 2) { const uint2 h = reinterpret_cast<const uint2 *>(hi)[slot]; r.w[4] = h.x; r.w[5] = h.y; }
    else if (H >= 3) { const uint4 h = reinterpret_cast<const uint4 *>(hi)[slot]; r.w[4] = h.x; r.w[5] = h.y; r.w[6] = h.z; r.w[7] = h.w; }
    return r;
}
constexpr int c4_hi_words(int nwords) { return nwords <= 4 ? 0 : nwords == 5 ? 1 : nwords == 6 ? 2 : 4; }


// 10 staging registers instead of 25: the 4-bit instance fits 5 waves per SIMD (90 VGPRs); the others would spill there
template <int B> constexpr int bs4_waves() { return B <= 4 ? 5 : 4; }
template <int B, typename CT>
__global__ __launch_bounds__(kCountThreads) __attribute__((amdgpu_waves_per_eu(bs4_waves<B>(), bs4_waves<B>()))) void count_bitslice4_kernel(
    const uint4 *__restrict__ P, uint32_t npairs, uint32_t n_groups, uint32_t m_trees, uint32_t d_start, uint32_t d_hi,
    uint64_t rank_lo, uint32_t n_dblk, uint32_t n_slots, const uint32_t *__restrict__ dprefix, const uint32_t *__restrict__ cprefix,
    CT *__restrict__ table, uint32_t *__restrict__ overflow_flag, uint32_t overwrite, uint32_t xcd_remap, uint32_t *__restrict__ wire,
    const uint32_t *__restrict__ perm) {
    constexpr int NB = B + 1;                           // planes of L and R
    constexpr int NWRAW = B < 4 ? 4 : B;                // words of a raw panel element (compact panel: max(B, 4) planes)
    constexpr int HR = c4_hi_words(NWRAW), HI = c4_hi_words(NB);
    __shared__ uint4 raw_lo[2][kC4Raw];
    __shared__ __align__(16) uint32_t raw_hi[2][HR ? kC4Raw * HR : 4];
    __shared__ uint4 img_lo[kWavesPerBlock][kWave];
    __shared__ __align__(16) uint32_t img_hi[kWavesPerBlock][HI ? kWave * HI : 4];

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t lb = blockIdx.x;
    if (xcd_remap & 1u) {
        const uint32_t nb = gridDim.x, q8 = nb / 8, r8 = nb % 8, xcd = lb % 8, y = lb / 8;
        lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + y;
    }
    // launch slot -> tile: the host pads every workgroup to four tiles of one (a-blocks, b-block, d-block)
    uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)perm[lb * kWavesPerBlock + wave]);
    const bool shadow = (tile & kShadowTile) != 0;
    tile &= ~kShadowTile;
    uint32_t dp_k, cp_c;
    const uint32_t k = wave_search_le(dprefix, 0, n_dblk, tile, lane, dp_k);
    const uint32_t local = tile - dp_k;
    const uint32_t d1 = d_hi - k * kDB;
    const uint32_t d0 = d1 > d_start + kDB ? d1 - kDB : d_start;
    const uint32_t c = wave_search_le(cprefix, 2, d1 - 1, local, lane, cp_c);
    const uint32_t tl = local - cp_c;                   // off-diagonal tile with both a-blocks (host guarantee)
    uint32_t Bk = (uint32_t)(2.0f * sqrtf((float)tl + 1.0f));
    while ((Bk * Bk) / 4 > tl) --Bk;
    while (((Bk + 1) * (Bk + 1)) / 4 <= tl) ++Bk;
    const uint32_t jt = tl - (Bk * Bk) / 4;
    const uint32_t blk0 = 2 * jt, blk1 = 2 * jt + 1;
    const uint32_t ia = lane & (kTA - 1), ib = lane / kTA;
    const uint32_t a1 = blk0 * kTA + ia, a2 = blk1 * kTA + ia, b = Bk * kTB + ib;
    const bool v1 = !shadow && b < c, v2 = v1;          // a1 < a2 < b by construction
    const uint32_t pi1 = (uint32_t)binom2(b) + a1, pi2 = (uint32_t)binom2(b) + a2;
    const uint32_t jlo = c >= d0 ? c + 1 - d0 : 0u, jhi = d1 - d0;

    // ---- what this lane requests per group: one shared element (waves 0..2) and one private element (lanes 0..23) ----
    const uint32_t r_j = lane >> 3, r_col = lane & 7, dE = d0 + r_j;
    uint32_t shoff = kS3Inv;
    if (wave == 0) shoff = pi1 * 16u;
    else if (wave == 1) shoff = pi2 * 16u;
    else if (wave == 2) { const uint32_t bE = Bk * kTB + r_col; if (dE < d1 && bE < dE) shoff = ((uint32_t)binom2(dE) + bE) * 16u; }
    uint32_t pvoff = kS3Inv;
    if (lane < 16) { const uint32_t xa = (lane < 8 ? blk0 : blk1) * kTA + (lane & 7); if (xa < c) pvoff = ((uint32_t)binom2(c) + xa) * 16u; }
    else if (lane < 24) { const uint32_t dy = d0 + (lane - 16); if (dy < d1 && dy > c) pvoff = ((uint32_t)binom2(dy) + c) * 16u; }
    const uint32_t group_bytes = npairs * (uint32_t)(NWRAW * 4), hi_base = npairs * 16u;
    const uint32_t pv_slot = kC4Shared + wave * kC4Private;

    uint32_t x0[kDB], x1[kDB], y0[kDB], y1[kDB];
#pragma unroll
    for (int j = 0; j < kDB; ++j) x0[j] = x1[j] = y0[j] = y1[j] = 0;

    auto rsrc_of = [&](uint32_t g) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(P) + (size_t)g * group_bytes), 0, (int)group_bytes, 0x00020000);
    };
    auto put_raw = [&](int bufi, const Planes &sh, const Planes &pv) {
        if (wave < 3) c4_put<NWRAW>(raw_lo[bufi], raw_hi[bufi], wave * kWave + lane, sh);
        if (lane < kC4Private) c4_put<NWRAW>(raw_lo[bufi], raw_hi[bufi], pv_slot + lane, pv);
    };
    // one 32-tree step: group g sits raw in buffer CUR; request group g_next, count g, park g_next raw in the other buffer
    auto step = [&](uint32_t g_next, auto cur_tag, auto full_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        // all waves have parked their part of this group (LDS writes complete before the barrier releases)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const Planes ab1 = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], lane);
        const Planes rowA1 = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], pv_slot + ia);
        const Planes ab2 = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], kWave + lane);
        const Planes rowA2 = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], pv_slot + 8 + ia);
        const Planes mbd = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], 2 * kWave + lane);
        const Planes mcd = c4_get<NWRAW>(raw_lo[CUR], raw_hi[CUR], pv_slot + 16 + r_j);
        const __amdgpu_buffer_rsrc_t r = rsrc_of(g_next);
        const Planes sh = buf_load_planes<NWRAW>(r, shoff, hi_base);
        const Planes pv = buf_load_planes<NWRAW>(r, pvoff, hi_base);
        c4_put<NB>(img_lo[wave], img_hi[wave], lane, sub_biased<B>(mbd, mcd));   // R element (d-row r_j, b-column r_col)
        const Planes L1 = sub_biased<B>(ab1, rowA1);
        const Planes L2 = sub_biased<B>(ab2, rowA2);
#pragma unroll
        for (int j = 0; j < kDB; ++j) {
            if (FULL || ((uint32_t)j >= jlo && (uint32_t)j < jhi)) {
                const Planes Rb = c4_get<NB>(img_lo[wave], img_hi[wave], j * 8 + ib);
                uint32_t gt, lt, gt2, lt2;
                cmp_planes<NB>(L1, Rb, gt, lt);
                popc_acc(gt, x0[j]);
                popc_acc(lt, x1[j]);
                cmp_planes<NB>(L2, Rb, gt2, lt2);
                popc_acc(gt2, y0[j]);
                popc_acc(lt2, y1[j]);
            }
        }
        put_raw(CUR ^ 1, sh, pv);
    };
    auto run = [&](auto full_tag) {
        using Z_ = std::integral_constant<int, 0>; using O_ = std::integral_constant<int, 1>;
        {   // group 0 -> buffer 0
            const __amdgpu_buffer_rsrc_t r = rsrc_of(0);
            put_raw(0, buf_load_planes<NWRAW>(r, shoff, hi_base), buf_load_planes<NWRAW>(r, pvoff, hi_base));
        }
        const uint32_t g_last = n_groups - 1;
        uint32_t g = 0;
        for (; g + 2 <= n_groups; g += 2) {
            step(g + 1, Z_{}, full_tag);
            step(min(g + 2, g_last), O_{}, full_tag);   // past the end: re-reads the last group (never used)
        }
        if (g < n_groups) step(g_last, Z_{}, full_tag);
    };
    // every wave of the workgroup must take part in every barrier: the two instances run the same number of steps
    if (jlo == 0 && jhi == (uint32_t)kDB) run(std::true_type{}); else run(std::false_type{});

    uint64_t bd4 = binom4(d0), bd3 = binom3(d0), bd2 = binom2(d0);
    const uint64_t rcb = binom3(c) - rank_lo;
    if (wire) {   // one word n0 | n1 << 16 per tuple (QS_COUNT_WIRE16X2)
#pragma unroll
        for (int j = 0; j < kDB; ++j) {
            const uint32_t d = d0 + j;
            const uint64_t base = bd4 + rcb;
            bd4 += bd3; bd3 += bd2; bd2 += d;
            if (d < d1 && d > c && v1) {
                uint32_t w = x0[j] | (x1[j] << 16), w2 = y0[j] | (y1[j] << 16);
                if (!overwrite) { w += wire[base + pi1]; w2 += wire[base + pi2]; }
                wire[base + pi1] = w;
                wire[base + pi2] = w2;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kDB; ++j) {
        const uint32_t d = d0 + j;
        const uint64_t base = bd4 + rcb;
        bd4 += bd3; bd3 += bd2; bd2 += d;
        if (d < d1 && d > c && v1) {
            {
                const uint64_t idx = (base + pi1) * 3;
                uint32_t w0 = x0[j], w1 = x1[j], w2 = m_trees - x0[j] - x1[j];
                if (!overwrite) { const Tuple3<CT> t = load_tuple(table + idx); w0 += t.a; w1 += t.b; w2 += t.c; }
                if (sizeof(CT) == 2 && ((w0 | w1 | w2) > 0xFFFFu)) atomicOr(overflow_flag, 1u);
                store_tuple(table + idx, w0, w1, w2);
            }
            if (v2) {
                const uint64_t idx = (base + pi2) * 3;
                uint32_t w0 = y0[j], w1 = y1[j], w2 = m_trees - y0[j] - y1[j];
                if (!overwrite) { const Tuple3<CT> t = load_tuple(table + idx); w0 += t.a; w1 += t.b; w2 += t.c; }
                if (sizeof(CT) == 2 && ((w0 | w1 | w2) > 0xFFFFu)) atomicOr(overflow_flag, 1u);
                store_tuple(table + idx, w0, w1, w2);
            }
        }
    }
}


hipError_t launch_count_bitslice3(hipStream_t s, const CountGeometry &g_in, const void *panel, int depth_bits, int mode,
                                  uint32_t n_groups, uint32_t m_trees, void *table, int count_bits, uint32_t *overflow_flag,
                                  bool overwrite, uint32_t *wire) {
    if (g_in.total_tiles == 0 || n_groups == 0) return hipSuccess;
    const uint32_t npairs = (uint32_t)binom2(g_in.n);
    CountGeometry g = g_in;
    // (count_bitslice4_kernel carries at most 7 planes per element: deeper classes take the plain kernel over g.perm with all tiles)
    if (mode == MODE_BINARY_FULL && g.perm_coop && g.n_coop && depth_bits <= 7) {
        // the tiles whose workgroups share their panel loads (count_bitslice4_kernel); the others follow below through
        // count_bitslice3_kernel with the rest list as its launch order (disjoint tiles: both write the same table)
        dim3 grid4(g.n_coop / kWavesPerBlock), block4(kCountThreads);
#define QS_BS4(BB, CT)                                                                                              \
    hipLaunchKernelGGL((count_bitslice4_kernel<BB, CT>), grid4, block4, 0, s, (const uint4 *)panel, npairs, n_groups, m_trees, \
                       g.d_lo, g.d_hi, g.rank_lo, g.n_dblk, g.n_coop, g.dprefix, g.cprefix, (CT *)table, overflow_flag,    \
                       overwrite ? 1u : 0u, 1u, wire, g.perm_coop)
#define QS_BS4_B(CT)                                                                                                \
    do {                                                                                                            \
        if (depth_bits <= 4) QS_BS4(4, CT);                                                                         \
        else if (depth_bits == 5) QS_BS4(5, CT);                                                                    \
        else if (depth_bits == 6) QS_BS4(6, CT);                                                                    \
        else QS_BS4(7, CT);                                                                                         \
    } while (0)
        if (count_bits == 32) QS_BS4_B(uint32_t); else QS_BS4_B(uint16_t);
#undef QS_BS4_B
#undef QS_BS4
        hipError_t e4 = hipGetLastError();
        if (e4 != hipSuccess) return e4;
        g.perm = g.perm_rest; g.total_tiles = g.n_rest;
        if (g.total_tiles == 0) return hipSuccess;
    }
    dim3 grid((g.total_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kCountThreads);
#define QS_BS3(BB, MM, CT)                                                                                          \
    hipLaunchKernelGGL((count_bitslice3_kernel<BB, MM, CT>), grid, block, 0, s, (const uint4 *)panel, npairs, n_groups, \
                       m_trees, g.d_lo, g.d_hi, g.rank_lo, g.n_dblk, g.total_tiles, g.dprefix, g.cprefix,           \
                       (CT *)table, overflow_flag, overwrite ? 1u : 0u, g.n >= 200 ? (((kSyncModes >> (MM)) & 1u) ? 3u : 1u) : 0u /* bit 0: XCD remap, bit 1: waves of a workgroup in step (QS_SYNC_MODES) */, wire, g.perm)
#define QS_BS3_B(MM, CT)                                                                                            \
    do {                                                                                                            \
        switch (depth_bits <= 4 ? 4 : depth_bits) {                                                                 \
            case 4: QS_BS3(4, MM, CT); break;                                                                       \
            case 5: QS_BS3(5, MM, CT); break;                                                                       \
            case 6: QS_BS3(6, MM, CT); break;                                                                       \
            case 7: QS_BS3(7, MM, CT); break;                                                                       \
            case 8: QS_BS3(8, MM, CT); break;                                                                       \
            case 9: QS_BS3(9, MM, CT); break;                                                                       \
            default: QS_BS3(10, MM, CT); break;                                                                     \
        }                                                                                                           \
    } while (0)
#define QS_BS3_BP(CT) QS_BS3_B(MODE_PARTIAL, CT)
    if (depth_bits > kMaxDepthBits) return hipErrorInvalidValue;
    if (count_bits == 32) {
        if (mode == MODE_BINARY_FULL) QS_BS3_B(MODE_BINARY_FULL, uint32_t);
        else if (mode == MODE_GENERAL_FULL) QS_BS3_B(MODE_GENERAL_FULL, uint32_t);
        else if (mode == MODE_BINARY_PARTIAL) QS_BS3_B(MODE_BINARY_PARTIAL, uint32_t);
        else QS_BS3_BP(uint32_t);
    } else {
        if (mode == MODE_BINARY_FULL) QS_BS3_B(MODE_BINARY_FULL, uint16_t);
        else if (mode == MODE_GENERAL_FULL) QS_BS3_B(MODE_GENERAL_FULL, uint16_t);
        else if (mode == MODE_BINARY_PARTIAL) QS_BS3_B(MODE_BINARY_PARTIAL, uint16_t);
        else QS_BS3_BP(uint16_t);
    }
#undef QS_BS3_BP
#undef QS_BS3_B
#undef QS_BS3
    return hipGetLastError();
}


// ======================================================================================
// scatter count kernel (tree-major, atomics)
// ======================================================================================

template <typename CT> __device__ __forceinline__ void table_atomic_inc(CT *table, uint64_t cell);
template <> __device__ __forceinline__ void table_atomic_inc<uint32_t>(uint32_t *table, uint64_t cell) {
    atomicAdd(&table[cell], 1u);
}
template <> __device__ __forceinline__ void table_atomic_inc<uint16_t>(uint16_t *table, uint64_t cell) {
    // packed half-word increment: cell totals stay < 2^16, so no carry crosses the half-word
    uint32_t *w = reinterpret_cast<uint32_t *>(table) + (cell >> 1);
    atomicAdd(w, (cell & 1) ? 0x10000u : 1u);
}

constexpr int kScatterMaxLeaves = 4096;

// One wavefront (= one workgroup) per inner node of one evaluation tree. The tree's tour is
// staged in LDS. For every oriented triple (pair side P; single sides Q, R) of the node's
// links the wave walks pairs (a,a') of P and members b of Q (wave-uniform), lanes enumerate
// c in R; a hit is counted only when min(a,a') < min(b,c).
template <typename CT>
__global__ __launch_bounds__(kWave) void count_scatter_kernel(const uint32_t *__restrict__ leaf_off,
                                                              const uint16_t *__restrict__ leaf_ids,
                                                              const uint32_t *__restrict__ node_tree,
                                                              const uint32_t *__restrict__ rng_off,
                                                              const uint16_t *__restrict__ ranges, uint32_t d_lo,
                                                              uint32_t d_hi, uint64_t rank_lo, CT *__restrict__ table) {
    __shared__ uint16_t ids[kScatterMaxLeaves];
    const uint32_t v = blockIdx.x, lane = threadIdx.x;
    const uint32_t t = node_tree[v];
    const uint32_t base = leaf_off[t], L = leaf_off[t + 1] - base;
    for (uint32_t i = lane; i < L; i += kWave) ids[i] = leaf_ids[base + i];
    __syncthreads();
    const uint32_t k0 = rng_off[v], k = rng_off[v + 1] - k0;
    for (uint32_t i1 = 0; i1 < k; ++i1)
        for (uint32_t i2 = i1 + 1; i2 < k; ++i2)
            for (uint32_t i3 = i2 + 1; i3 < k; ++i3) {
                const uint32_t li[3] = {i1, i2, i3};
                for (int o = 0; o < 3; ++o) { // which link is the pair side
                    const uint32_t lp = li[o], lq = li[(o + 1) % 3], lr = li[(o + 2) % 3];
                    const uint32_t ps = ranges[2 * (k0 + lp)], pe = ranges[2 * (k0 + lp) + 1];
                    const uint32_t qs_ = ranges[2 * (k0 + lq)], qe = ranges[2 * (k0 + lq) + 1];
                    const uint32_t rs = ranges[2 * (k0 + lr)], re = ranges[2 * (k0 + lr) + 1];
                    const uint32_t np = (pe + L - ps) % L, nq = (qe + L - qs_) % L, nr = (re + L - rs) % L;
                    for (uint32_t x = 0; x < np; ++x) {
                        const uint32_t a = ids[(ps + x) % L];
                        for (uint32_t y = x + 1; y < np; ++y) {
                            const uint32_t a2 = ids[(ps + y) % L];
                            const uint32_t amin = min(a, a2);
                            for (uint32_t z = 0; z < nq; ++z) {
                                const uint32_t bq = ids[(qs_ + z) % L];
                                if (bq < amin) continue; // uniform
                                for (uint32_t w = lane; w < nr; w += kWave) {
                                    const uint32_t cr = ids[(rs + w) % L];
                                    if (cr < amin) continue;
                                    // sort the four ids
                                    uint32_t s0 = min(a, a2), s1 = max(a, a2), s2 = min(bq, cr), s3 = max(bq, cr);
                                    // s0 is the global minimum by construction
                                    uint32_t lo = min(s1, s2), hi = max(s1, s2);
                                    uint32_t m1 = lo, m2 = min(hi, s3), m3 = max(hi, s3);
                                    if (m3 < d_lo || m3 >= d_hi) continue;
                                    const int slot = slot_of_pairing(a, a2, bq, cr);
                                    const uint64_t cell = (rank4(s0, m1, m2, m3) - rank_lo) * 3 + (uint64_t)slot;
                                    table_atomic_inc<CT>(table, cell);
                                }
                            }
                        }
                    }
                }
            }
}

hipError_t launch_count_scatter(hipStream_t s, const DeviceBatch &b, uint32_t n, uint32_t d_lo, uint32_t d_hi,
                                uint64_t rank_lo, void *table, int count_bits) {
    (void)n;
    if (b.n_nodes == 0) return hipSuccess;
    dim3 grid(b.n_nodes), block(kWave);
    if (count_bits == 32)
        hipLaunchKernelGGL(count_scatter_kernel<uint32_t>, grid, block, 0, s, b.leaf_off, b.leaf_ids, b.node_tree, b.rng_off,
                           b.ranges, d_lo, d_hi, rank_lo, (uint32_t *)table);
    else
        hipLaunchKernelGGL(count_scatter_kernel<uint16_t>, grid, block, 0, s, b.leaf_off, b.leaf_ids, b.node_tree, b.rng_off,
                           b.ranges, d_lo, d_hi, rank_lo, (uint16_t *)table);
    return hipGetLastError();
}

// ======================================================================================
// depth clamp: the corrections of trees counted in a class below their own depth bits
// ======================================================================================
// The bit-sliced kernel's work per (quartet, 32 trees) is 2(B+1)+2 instructions with B the depth bits of the CLASS. The
// four-point test does not need depths: any labelling of the inner nodes that grows strictly along every root-to-leaf path
// gives the same answers, and one that grows WEAKLY gives either the same answer or "all three sums equal" -- never a
// wrong topology (the two equal sums of a quartet are equal by node identity). So a tree whose deepest LCA needs B+1 bits
// can be counted in the B-bit class with every depth cut at 2^B - 1 (the panel builders do that), at the price of the
// quartets the cut ties: exactly those with at least THREE leaves below one node of depth 2^B - 1 -- in centred random
// trees a handful of small subtrees (512 taxa: 76 % of the trees need 5 bits, median 5 600 such quartets of 2.8e9 per
// tree). In the tour arrays such a subtree is a maximal run of adjacent-LCA depths >= 2^B - 1 (qs_abi.hip plan_depth_clamp
// finds them and decides per tree whether the cut pays). This kernel adds what the cut lost: for every triple (i < j < k)
// of a run and every fourth leaf -- outside the run: the triple's cherry pairs up, the topology follows from the two
// adjacent LCA depths of the triple alone; inside the run behind k: the four-point rule on the three adjacent depths --
// one atomic increment of the true topology's cell; in the binary modes the count kernel has put the tied quartet into
// the third cell (n2 = trees - n0 - n1), so that cell is decremented when the true topology is another one.
// Workgroup = one unit = (tree, run, a stretch of the run's triples in colex order); threads = the fourth leaves.
// Replaces nothing in the reference: its loop is shape-independent (QuartetCounterLookup.hpp:65-106).
constexpr int kFixMaxRun = 128;     // leaves of a run (longer runs: the tree keeps its own depth class)
constexpr int kFixThreads = 256;
enum FixRule { FIX_BINARY = 0, FIX_GENERAL = 1, FIX_WIRE = 2 };

template <typename CT> __device__ __forceinline__ void table_atomic_dec(CT *table, uint64_t cell);
template <> __device__ __forceinline__ void table_atomic_dec<uint32_t>(uint32_t *table, uint64_t cell) { atomicSub(&table[cell], 1u); }
template <> __device__ __forceinline__ void table_atomic_dec<uint16_t>(uint16_t *table, uint64_t cell) {
    // the cell holds at least the tied quartet's own count at this point (the count kernel of the slice ran before): no borrow
    uint32_t *w = reinterpret_cast<uint32_t *>(table) + (cell >> 1);
    atomicSub(w, (cell & 1) ? 0x10000u : 1u);
}

template <typename CT, int RULE>
__global__ __launch_bounds__(kFixThreads) void clamp_fix_kernel(const FixUnit *__restrict__ units, const uint32_t *__restrict__ leaf_off,
                                                                const uint16_t *__restrict__ leaf_ids, const uint16_t *__restrict__ adj_depth,
                                                                uint32_t d_lo, uint32_t d_hi, uint64_t rank_lo, CT *__restrict__ table,
                                                                uint32_t *__restrict__ wire) {
    __shared__ uint16_t ids[kScatterMaxLeaves];
    __shared__ uint16_t rm[kFixMaxRun][kFixMaxRun];   // rm[a][b], a < b: LCA depth of the run's leaves a and b
    const FixUnit u = units[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = leaf_off[u.tree], L = leaf_off[u.tree + 1] - base;
    const uint32_t i0 = u.run & 0xFFFFu, s = u.run >> 16;
    for (uint32_t i = tid; i < L; i += kFixThreads) ids[i] = leaf_ids[base + i];
    for (uint32_t a = tid; a + 1 < s; a += kFixThreads) {
        uint32_t m = 0xFFFFu;
        for (uint32_t b = a + 1; b < s; ++b) { m = min(m, (uint32_t)adj_depth[base + i0 + b - 1]); rm[a][b] = (uint16_t)m; }
    }
    __syncthreads();
    // triple t_lo in colex order: t = C(k,3) + C(j,2) + i, i < j < k
    uint32_t k = 2, rest = u.t_lo;
    while ((uint32_t)binom3(k + 1) <= rest) ++k;
    rest -= (uint32_t)binom3(k);
    uint32_t j = 1;
    while ((uint32_t)binom2(j + 1) <= rest) ++j;
    uint32_t i = rest - (uint32_t)binom2(j);
    const uint32_t n_out = L - s;
    for (uint32_t t = u.t_lo; t < u.t_hi; ++t) {
        const uint32_t mij = rm[i][j], mjk = rm[j][k];
        const uint32_t x1 = ids[i0 + i], x2 = ids[i0 + j], x3 = ids[i0 + k];
        const uint32_t n_in = s - 1 - k;
        // fourth leaf outside the run: cherry (x1,x2) if lca(i,j) is the deeper one, (x2,x3) if lca(j,k) is; equal: the triple is
        // unresolved in the tree itself (a multifurcation), nothing to add
        const uint32_t q0 = mij != mjk ? 0u : n_out;
        for (uint32_t q = q0 + tid; q < n_out + n_in; q += kFixThreads) {
            uint32_t pa, pb, oa, ob;   // the tree displays pa pb | oa ob
            if (q < n_out) {
                const uint32_t x = ids[q < i0 ? q : q + s];
                if (mij > mjk) { pa = x1; pb = x2; oa = x3; ob = x; } else { pa = x2; pb = x3; oa = x1; ob = x; }
            } else {
                const uint32_t l = k + 1 + (q - n_out), x4 = ids[i0 + l];
                const uint32_t mkl = rm[k][l], mx = max(mij, mkl);
                if (mjk < mx) { pa = x1; pb = x2; oa = x3; ob = x4; }
                else if (mjk > mx) { pa = x1; pb = x4; oa = x2; ob = x3; }
                else continue;
            }
            const uint32_t lo1 = min(pa, pb), hi1 = max(pa, pb), lo2 = min(oa, ob), hi2 = max(oa, ob);
            const uint32_t s0 = min(lo1, lo2), s3 = max(hi1, hi2), m1 = max(lo1, lo2), m2 = min(hi1, hi2);
            if (s3 < d_lo || s3 >= d_hi) continue;
            const uint32_t s1 = min(m1, m2), s2 = max(m1, m2);
            const int slot = slot_of_pairing(pa, pb, oa, ob);
            const uint64_t tup = rank4(s0, s1, s2, s3) - rank_lo;
            if (RULE == FIX_WIRE) { if (slot < 2) atomicAdd(&wire[tup], slot ? 0x10000u : 1u); }   // n0 | n1 << 16; n2 is implied
            else if (RULE == FIX_GENERAL) table_atomic_inc<CT>(table, tup * 3 + (uint64_t)slot);    // the tie counted nothing
            else if (slot != 2) { table_atomic_inc<CT>(table, tup * 3 + (uint64_t)slot); table_atomic_dec<CT>(table, tup * 3 + 2); }
        }
        if (++i == j) { i = 0; if (++j == k) { j = 1; ++k; } }
    }
}

hipError_t launch_clamp_fix(hipStream_t s, const DeviceBatch &b, const FixUnit *units, uint32_t n_units, uint32_t d_lo, uint32_t d_hi,
                            uint64_t rank_lo, void *table, int count_bits, int mode, uint32_t *wire) {
    if (n_units == 0) return hipSuccess;
    dim3 grid(n_units), block(kFixThreads);
    const bool gen = mode == MODE_GENERAL_FULL || mode == MODE_PARTIAL;
#define QS_FIX(CT, RULE) hipLaunchKernelGGL((clamp_fix_kernel<CT, RULE>), grid, block, 0, s, units, b.leaf_off, b.leaf_ids, b.adj_depth, d_lo, d_hi, rank_lo, (CT *)table, wire)
    if (wire) QS_FIX(uint32_t, FIX_WIRE);
    else if (count_bits == 32) { if (gen) QS_FIX(uint32_t, FIX_GENERAL); else QS_FIX(uint32_t, FIX_BINARY); }
    else { if (gen) QS_FIX(uint16_t, FIX_GENERAL); else QS_FIX(uint16_t, FIX_BINARY); }
#undef QS_FIX
    return hipGetLastError();
}

// ======================================================================================
// u32 table -> u16 table (wire format of the multi-GPU all-reduce when all totals stay below 2^16)
// ======================================================================================
__global__ __launch_bounds__(256) void pack16_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                     uint64_t n_words, uint64_t n_cells, uint32_t *__restrict__ overflow_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // one output word = two cells
    if (i >= n_words) return;
    const uint32_t lo = src[2 * i];
    const uint32_t hi = (2 * i + 1 < n_cells) ? src[2 * i + 1] : 0u;
    if ((lo | hi) > 0xFFFFu) atomicOr(overflow_flag, 1u);
    dst[i] = (lo & 0xFFFFu) | (hi << 16);
}

// Two-cell wire format for batches in which every tree resolves every quartet (binary trees holding all taxa):
// n0 + n1 + n2 = number of trees, so a tuple travels as ONE word n0 | n1 << 16 and n2 is restored after the
// collective from the total number of trees. flags[0]: a count does not fit 16 bits; flags[1]: a tuple does not sum
// to `trees` (the batch was not binary/full).
__global__ __launch_bounds__(256) void pack16x2_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                       uint64_t n_tuples, uint32_t trees, uint32_t *__restrict__ overflow_flag,
                                                       uint32_t *__restrict__ shape_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tuples) return;
    const uint32_t n0 = src[3 * i], n1 = src[3 * i + 1], n2 = src[3 * i + 2];
    if ((n0 | n1) > 0xFFFFu) atomicOr(overflow_flag, 1u);
    if (n0 + n1 + n2 != trees) atomicOr(shape_flag, 1u);
    dst[i] = (n0 & 0xFFFFu) | (n1 << 16);
}
__global__ __launch_bounds__(256) void unpack16x2_kernel(const uint32_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                         uint64_t n_tuples, uint32_t trees, uint32_t *__restrict__ shape_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tuples) return;
    const uint32_t w = src[i], n0 = w & 0xFFFFu, n1 = w >> 16;
    if (n0 + n1 > trees) atomicOr(shape_flag, 1u);
    dst[3 * i] = (uint16_t)n0; dst[3 * i + 1] = (uint16_t)n1; dst[3 * i + 2] = (uint16_t)(trees - n0 - n1);
}
// The same for totals of 65536 trees and more (BASELINE configs[3]: 100 000 trees): TWO u32 cells (n0, n1) per tuple on the
// wire instead of three, 8 bytes per quartet instead of 12; n2 = total trees - n0 - n1 after the collective.
__global__ __launch_bounds__(256) void pack32x2_kernel(const uint32_t *__restrict__ src, uint2 *__restrict__ dst, uint64_t n_tuples,
                                                       uint32_t trees, uint32_t *__restrict__ shape_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tuples) return;
    const uint32_t n0 = src[3 * i], n1 = src[3 * i + 1], n2 = src[3 * i + 2];
    if (n0 + n1 + n2 != trees) atomicOr(shape_flag, 1u);
    dst[i] = make_uint2(n0, n1);
}
__global__ __launch_bounds__(256) void unpack32x2_kernel(const uint2 *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n_tuples,
                                                         uint32_t trees, uint32_t *__restrict__ shape_flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tuples) return;
    const uint2 w = src[i];
    if ((uint64_t)w.x + w.y > trees) atomicOr(shape_flag, 1u);
    dst[3 * i] = w.x; dst[3 * i + 1] = w.y; dst[3 * i + 2] = trees - w.x - w.y;
}
hipError_t launch_pack32x2(hipStream_t s, const void *table_u32, void *dst, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag) {
    if (n_tuples == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_tuples + 255) / 256));
    hipLaunchKernelGGL(pack32x2_kernel, grid, block, 0, s, (const uint32_t *)table_u32, (uint2 *)dst, n_tuples, trees, shape_flag);
    return hipGetLastError();
}
hipError_t launch_unpack32x2(hipStream_t s, const void *src, void *dst_u32, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag) {
    if (n_tuples == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_tuples + 255) / 256));
    hipLaunchKernelGGL(unpack32x2_kernel, grid, block, 0, s, (const uint2 *)src, (uint32_t *)dst_u32, n_tuples, trees, shape_flag);
    return hipGetLastError();
}
hipError_t launch_pack16x2(hipStream_t s, const void *table_u32, void *dst, uint64_t n_tuples, uint32_t trees,
                           uint32_t *overflow_flag, uint32_t *shape_flag) {
    if (n_tuples == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_tuples + 255) / 256));
    hipLaunchKernelGGL(pack16x2_kernel, grid, block, 0, s, (const uint32_t *)table_u32, (uint32_t *)dst, n_tuples, trees, overflow_flag, shape_flag);
    return hipGetLastError();
}
hipError_t launch_unpack16x2(hipStream_t s, const void *src, void *dst_u16, uint64_t n_tuples, uint32_t trees, uint32_t *shape_flag) {
    if (n_tuples == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_tuples + 255) / 256));
    hipLaunchKernelGGL(unpack16x2_kernel, grid, block, 0, s, (const uint32_t *)src, (uint16_t *)dst_u16, n_tuples, trees, shape_flag);
    return hipGetLastError();
}

// dst[i] += src[0][i] + ... + src[n_src-1][i] over 32-bit words: the single-process reduce(-scatter) of the multi-GPU host
// (`--reduce p2p`): GPU g sums its chunk of every peer's table with plain 16-byte loads over xGMI (the sources are peer
// device memory, mapped by hipDeviceEnablePeerAccess), no communicator. Grid-stride; every load is read once: non-temporal.
constexpr int kSumMaxSrc = 15;
struct SumSources { const uint32_t *p[kSumMaxSrc]; };
__global__ __launch_bounds__(256) void sum_words_kernel(uint32_t *__restrict__ dst, SumSources src, uint32_t n_src, uint64_t n_words) {
    const uint64_t n4 = n_words / 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        qs_u32x4 acc = reinterpret_cast<const qs_u32x4 *>(dst)[i];
        for (uint32_t k = 0; k < n_src; ++k) acc += __builtin_nontemporal_load(reinterpret_cast<const qs_u32x4 *>(src.p[k]) + i);
        reinterpret_cast<qs_u32x4 *>(dst)[i] = acc;
    }
    const uint64_t t = 4 * n4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // the last n_words % 4 words
    if (t < n_words) {
        uint32_t acc = dst[t];
        for (uint32_t k = 0; k < n_src; ++k) acc += src.p[k][t];
        dst[t] = acc;
    }
}
hipError_t launch_sum_words(hipStream_t s, void *dst, const void *const *src, uint32_t n_src, uint64_t n_words, int n_cu) {
    if (n_words == 0 || n_src == 0) return hipSuccess;
    if (n_src > (uint32_t)kSumMaxSrc) return hipErrorInvalidValue;
    SumSources ss;
    for (uint32_t k = 0; k < (uint32_t)kSumMaxSrc; ++k) ss.p[k] = k < n_src ? (const uint32_t *)src[k] : nullptr;
    const uint64_t want = (n_words / 4 + 255) / 256;
    dim3 block(256), grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)std::max(n_cu, 1) * 16)));
    hipLaunchKernelGGL(sum_words_kernel, grid, block, 0, s, (uint32_t *)dst, ss, n_src, n_words);
    return hipGetLastError();
}

hipError_t launch_pack16(hipStream_t s, const void *table_u32, void *dst, uint64_t n_cells, uint32_t *overflow_flag) {
    const uint64_t n_words = (n_cells + 1) / 2;
    if (n_words == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_words + 255) / 256));
    hipLaunchKernelGGL(pack16_kernel, grid, block, 0, s, (const uint32_t *)table_u32, (uint32_t *)dst, n_words, n_cells, overflow_flag);
    return hipGetLastError();
}

// ======================================================================================
// lookup (countQuartetOccurrences, QuartetCounterLookup.hpp:299-318)
// ======================================================================================

template <typename CT>
__global__ void lookup_kernel(uint32_t d_lo, uint32_t d_hi, uint64_t rank_lo, const CT *__restrict__ table, uint64_t nq,
                              const uint16_t *__restrict__ abcd, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint32_t a = abcd[4 * i], b = abcd[4 * i + 1], c = abcd[4 * i + 2], d = abcd[4 * i + 3];
    uint32_t lo1 = min(a, b), hi1 = max(a, b), lo2 = min(c, d), hi2 = max(c, d);
    uint32_t s0 = min(lo1, lo2), s3 = max(hi1, hi2);
    uint32_t m1 = max(lo1, lo2), m2 = min(hi1, hi2);
    uint32_t s1 = min(m1, m2), s2 = max(m1, m2);
    uint64_t o0 = 0, o1 = 0, o2 = 0;
    if (s3 >= d_lo && s3 < d_hi && s0 != s1 && s1 != s2 && s2 != s3) {
        const uint64_t cell = (rank4(s0, s1, s2, s3) - rank_lo) * 3;
        o0 = table[cell + slot_of_pairing(a, b, c, d)];
        o1 = table[cell + slot_of_pairing(a, c, b, d)];
        o2 = table[cell + slot_of_pairing(a, d, b, c)];
    }
    out[3 * i] = o0; out[3 * i + 1] = o1; out[3 * i + 2] = o2;
}

hipError_t launch_lookup(hipStream_t s, uint32_t n, uint32_t d_lo, uint32_t d_hi, uint64_t rank_lo, const void *table,
                         int count_bits, uint64_t nq, const uint16_t *abcd_dev, uint64_t *out_dev) {
    (void)n;
    if (nq == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((nq + 255) / 256));
    if (count_bits == 32)
        hipLaunchKernelGGL(lookup_kernel<uint32_t>, grid, block, 0, s, d_lo, d_hi, rank_lo, (const uint32_t *)table, nq,
                           abcd_dev, out_dev);
    else
        hipLaunchKernelGGL(lookup_kernel<uint16_t>, grid, block, 0, s, d_lo, d_hi, rank_lo, (const uint16_t *)table, nq,
                           abcd_dev, out_dev);
    return hipGetLastError();
}

} // namespace qs
