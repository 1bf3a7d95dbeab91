// qs_bitslice3.hpp -- device code of the bit-sliced gather count shared by count_bitslice3_kernel (qs_count.hip: one class of
// trees per launch) and count_bitslice3_fused_kernel (qs_count_fused.hip: the classes of a mixed batch in ONE launch).
// Arithmetic, LDS image and the 32-tree step are described in qs_count.hip ("bit-sliced gather path"); this header holds
//   * the plane arithmetic (Planes, sub_biased, cmp_planes, gt_planes) and the panel / LDS element accessors,
//   * Bs3Tile + bs3_decode_tile: launch slot -> (d-block, c, a-blocks, b-block) and the lane's quartets,
//   * bs3_segment<B, MODE>: all 32-tree groups of ONE class streamed past the tile, counters in the caller's registers,
//   * bs3_store: the tile's tuples to the table.
// Replaces QuartetCounterLookup::updateQuartetsThreeClades (QuartetCounterLookup.hpp:65-106) together with the launchers.
#pragma once
#include "qs_common.hpp"
#include "qs_internal.hpp"

#include <type_traits>

namespace qs {

// acc += popcount(v) in ONE instruction (v_bcnt_u32_b32 has an accumulate operand; hipcc otherwise
// emits bcnt with 0 followed by an add)
__device__ __forceinline__ void popc_acc(uint32_t v, uint32_t &acc) {
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(v));
}

// The same search done by the whole wave in at most two memory latencies (instead of log2(hi - lo) dependent scalar
// loads): lane i first looks at arr[lo + i * stride] (stride = ceil(range / 64)), a ballot gives the segment that
// holds the answer, then the lanes look at that segment's (at most 64) consecutive entries. *value = arr[result],
// taken from the loaded registers with v_readlane. Ranges up to 4096 entries (n_taxa <= 4096).
__device__ __forceinline__ uint32_t wave_search_le(const uint32_t *__restrict__ arr, uint32_t lo, uint32_t hi, uint32_t key,
                                                   uint32_t lane, uint32_t &value) {
    const uint32_t range = hi - lo, stride = (range + kWave - 1) / kWave;
    uint32_t seg_lo = lo;
    if (stride > 1) {
        const uint32_t i = lo + lane * stride;
        const uint32_t v = i < hi ? arr[i] : 0xFFFFFFFFu;
        const unsigned long long m = __ballot(v <= key);      // lane 0 always hits (arr[lo] <= key)
        seg_lo = lo + ((uint32_t)__builtin_popcountll(m) - 1) * stride;
    }
    const uint32_t seg_hi = min(seg_lo + (stride > 1 ? stride : (uint32_t)kWave), hi);
    const uint32_t i = seg_lo + lane;
    const uint32_t v = i < seg_hi ? arr[i] : 0xFFFFFFFFu;
    const unsigned long long m = __ballot(v <= key);
    const int top = 63 - __builtin_clzll(m);
    value = (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(top));
    return seg_lo + (uint32_t)top;
}

constexpr int kTA = 8, kTB = 8;                         // tile sides (kTA * kTB == 64 lanes)

constexpr int kBitWords = 12;  // words of a Planes value in registers: up to 11 planes (R / L of 10-bit depths) + the presence word
constexpr int kPres = kBitWords - 1; // where the presence word ("pair present in tree t", partial batches) travels in registers
constexpr int kMaxDepthBits = 10;    // deepest bit-sliced class: LCA depths below 1024 (deeper trees: byte-SWAR kernel)
constexpr int kBitTrees = 32;  // trees per element

struct Planes { uint32_t w[kBitWords]; };

// One v_bitop3_b32: any boolean function of three words, given by its truth table
// TT = f(0xF0, 0xCC, 0xAA) (bit i of TT = f at (a,b,c) = (i>>2&1, i>>1&1, i&1)). hipcc only fuses some
// expression shapes into bitop3 (it fell back to xnor/and_or pairs here), so the hot chains use it
// explicitly. Full rate on gfx950 (tools/valu_rates.hip).
template <int TT> __device__ __forceinline__ uint32_t lut3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
constexpr int kTT_XOR3 = 0xF0 ^ 0xCC ^ 0xAA;                                  // a ^ b ^ c
constexpr int kTT_LT = ((0x0F & 0xCC) | (~(0xF0 ^ 0xCC) & 0xAA)) & 0xFF;      // (~a & b) | (~(a ^ b) & c): borrow / less-than step
constexpr int kTT_GT = ((0xF0 & 0x33) | (~(0xF0 ^ 0xCC) & 0xAA)) & 0xFF;      // (a & ~b) | (~(a ^ b) & c): greater-than step

// LDS image with HW (1, 2, 4 or 8) upper words per slot: words 0..3 at buf[e], the upper words in an array
// of HW-word records behind the `stride` 16-byte slots
template <int HW> __device__ __forceinline__ Planes lds_load_hw(const uint4 *buf, uint32_t e, int stride) {
    const uint4 lo = buf[e];
    Planes r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
#pragma unroll
    for (int k = 4; k < kBitWords; ++k) r.w[k] = 0;
    if (HW == 1) r.w[4] = reinterpret_cast<const uint32_t *>(buf + stride)[e];
    else if (HW == 2) { const uint2 h = reinterpret_cast<const uint2 *>(buf + stride)[e]; r.w[4] = h.x; r.w[5] = h.y; }
    else if (HW == 4) { const uint4 h = buf[stride + e]; r.w[4] = h.x; r.w[5] = h.y; r.w[6] = h.z; r.w[7] = h.w; }
    else if (HW == 5 || HW == 6) {   // 4 + 1 or 4 + 2 upper words: a second 16-byte array, then a 4- or 8-byte one (9- / 10-word operands)
        const uint4 h = buf[stride + e];
        r.w[4] = h.x; r.w[5] = h.y; r.w[6] = h.z; r.w[7] = h.w;
        if (HW == 5) r.w[8] = reinterpret_cast<const uint32_t *>(buf + 2 * stride)[e];
        else { const uint2 g = reinterpret_cast<const uint2 *>(buf + 2 * stride)[e]; r.w[8] = g.x; r.w[9] = g.y; }
    }
    else {   // 8 upper words: two 16-byte records per slot
        const uint4 h = buf[stride + 2 * e], g = buf[stride + 2 * e + 1];
        r.w[4] = h.x; r.w[5] = h.y; r.w[6] = h.z; r.w[7] = h.w; r.w[8] = g.x; r.w[9] = g.y; r.w[10] = g.z; r.w[11] = g.w;
    }
    return r;
}
template <int HW> __device__ __forceinline__ void lds_store_hw(uint4 *buf, uint32_t e, int stride, const Planes &r) {
    buf[e] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
    if (HW == 1) reinterpret_cast<uint32_t *>(buf + stride)[e] = r.w[4];
    else if (HW == 2) reinterpret_cast<uint2 *>(buf + stride)[e] = make_uint2(r.w[4], r.w[5]);
    else if (HW == 4) buf[stride + e] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
    else if (HW == 5 || HW == 6) {
        buf[stride + e] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
        if (HW == 5) reinterpret_cast<uint32_t *>(buf + 2 * stride)[e] = r.w[8];
        else reinterpret_cast<uint2 *>(buf + 2 * stride)[e] = make_uint2(r.w[8], r.w[9]);
    }
    else { buf[stride + 2 * e] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]); buf[stride + 2 * e + 1] = make_uint4(r.w[8], r.w[9], r.w[10], r.w[11]); }
}
// x - y + 2^B over B planes -> B+1 planes (unsigned, bias 2^B); word 7 = presence(x) & presence(y). The top plane is kept
// INVERTED (it holds the final borrow, 1 <=> x < y, instead of its complement): every consumer compares two such numbers,
// and [~p > ~q] = [p < q], so the comparisons swap their truth tables at the top plane (cmp_planes / gt_planes) and the
// v_not_b32 per difference is gone (3 of ~270 VALU instructions of a 32-tree step).
template <int B>
__device__ __forceinline__ Planes sub_biased(const Planes &x, const Planes &y) {
    Planes r;
    uint32_t br = 0;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const uint32_t a = x.w[k], b = y.w[k];
        r.w[k] = lut3<kTT_XOR3>(a, b, br);
        br = lut3<kTT_LT>(a, b, br);
    }
    r.w[B] = br;
#pragma unroll
    for (int k = B + 1; k < kPres; ++k) r.w[k] = 0;
    r.w[kPres] = x.w[kPres] & y.w[kPres];   // (B <= 10: the presence word never collides with a plane)
    return r;
}

// [l > r] and [l < r] for two results of sub_biased (NB = B+1 planes, the top one inverted), 32 trees at once
template <int NB>
__device__ __forceinline__ void cmp_planes(const Planes &l, const Planes &r, uint32_t &gt, uint32_t &lt) {
    gt = 0; lt = 0;
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        const uint32_t a = l.w[k], b = r.w[k];
        gt = lut3<kTT_GT>(a, b, gt);
        lt = lut3<kTT_LT>(a, b, lt);
    }
    const uint32_t a = l.w[NB - 1], b = r.w[NB - 1];   // inverted planes: the roles of the two tables swap
    gt = lut3<kTT_LT>(a, b, gt);
    lt = lut3<kTT_GT>(a, b, lt);
}
template <int NB>
__device__ __forceinline__ uint32_t gt_planes(const Planes &l, const Planes &r) {
    uint32_t gt = 0;
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
        const uint32_t a = l.w[k], b = r.w[k];
        gt = lut3<kTT_GT>(a, b, gt);
    }
    return lut3<kTT_LT>(l.w[NB - 1], r.w[NB - 1], gt);
}

// LDS image per wave: RC columns (16; 24 in the general / partial modes, which also stage R of the two a-blocks and M[b,c])
// x 8 d-rows of R elements, then RC elements M[x,c]
constexpr int s3_cols(bool gen) { return gen ? 24 : 16; }
constexpr int s3_row0(bool gen) { return kDB * s3_cols(gen); }              // R elements: slot = d-row * RC + column
constexpr int s3_slots(bool gen) { return s3_row0(gen) + s3_cols(gen); }    // 144 / 216
constexpr uint32_t kS3Inv = 0x80000000u;       // offset beyond any tree group (also after >> 2): the buffer load returns zeros

typedef uint32_t qs_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t qs_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t qs_u32x3 __attribute__((ext_vector_type(3)));
typedef qs_u32x3 qs_u32x3_a4 __attribute__((aligned(4)));   // 12-byte vector at a 4-byte aligned address

// panel element of the pair at 16-byte-slot offset voff (= pair * 16) of the tree group behind `r`: planes 0..3 from
// the group's lo array, the NW - 4 upper planes from the array behind it (hi_base = npairs * 16 bytes into the group)
template <int NW> __device__ __forceinline__ Planes buf_load_planes(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t hi_base) {
    Planes p;
    const qs_u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    p.w[0] = lo.x; p.w[1] = lo.y; p.w[2] = lo.z; p.w[3] = lo.w;
#pragma unroll
    for (int k = 4; k < kBitWords; ++k) p.w[k] = 0;
    constexpr int H = NW - 4;                                   // upper words: a record of H words per pair behind the lo array
    const uint32_t hoff = (voff >> 2) * (uint32_t)(H > 0 ? H : 1); // = pair * H * 4 bytes (voff = pair * 16)
    if (H == 1) p.w[4] = __builtin_amdgcn_raw_buffer_load_b32(r, hoff, hi_base, 0);
    else if (H == 2) { const qs_u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(r, hoff, hi_base, 0); p.w[4] = h.x; p.w[5] = h.y; }
    else if (H == 3) { const qs_u32x3 h = __builtin_amdgcn_raw_buffer_load_b96(r, hoff, hi_base, 0); p.w[4] = h.x; p.w[5] = h.y; p.w[6] = h.z; }
    else if (H >= 4) {
        const qs_u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(r, hoff, hi_base, 0);
        p.w[4] = h.x; p.w[5] = h.y; p.w[6] = h.z; p.w[7] = h.w;
        if (H == 5) p.w[8] = __builtin_amdgcn_raw_buffer_load_b32(r, hoff + 16, hi_base, 0);
        else if (H == 6) { const qs_u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(r, hoff + 16, hi_base, 0); p.w[8] = g.x; p.w[9] = g.y; }
        else if (H == 7) { const qs_u32x3 g = __builtin_amdgcn_raw_buffer_load_b96(r, hoff + 16, hi_base, 0); p.w[8] = g.x; p.w[9] = g.y; p.w[10] = g.z; }
    }
    return p;
}

#ifndef QS_SYNC_MODES
#define QS_SYNC_MODES 0xDu   /* modes (bit = CountMode) whose four waves of a workgroup walk their 32-tree steps in step (one s_barrier per step, launched
                             * with xcd_remap bit 1): binary_full -5 % at 512 taxa (round 2), partial and binary_partial unchanged, general_full
                             * 3.5 % FASTER without it (round 6: profiles/r06_experiments.md 7) -> every mode but general_full (bit 1).
                             * The flag must stay a RUN-TIME value (a kernel argument): with the barrier compiled out the general_full step
                             * takes 153 instead of 84-87 ms -- the s_barrier at the top of a step is also the fence that keeps LLVM from sinking
                             * the step's panel loads to their first use */
#endif
constexpr uint32_t kSyncModes = QS_SYNC_MODES;
#ifndef QS_BIN_PRIO
#define QS_BIN_PRIO 1       /* binary modes: s_setprio 1 around the compare chains of a step (A/B: 0) */
#endif
#ifndef QS_PROBE_ROWS
#define QS_PROBE_ROWS kDB
#endif
#ifndef QS_PROBE_GEN_WAVES
#define QS_PROBE_GEN_WAVES 0   /* probe builds: waves per SIMD of the general / partial instances up to QS_GEN3_MAXB bits */
#endif
#ifndef QS_BS3_WAVES
#define QS_BS3_WAVES 4
#endif
#ifndef QS_GEN3_MAXB
#define QS_GEN3_MAXB 6   /* general / partial instances up to this many depth bits are held to 3 waves per SIMD (168 VGPRs, a few spills) */
#endif

// A/B switches of the general / partial instances (profiles/r05_experiments.md 3; 512 taxa x 1500 trees, 20 % of the edges collapsed,
// ms per step on one box): OPAQUE 0 / 1 = 81.4 / 90.6 -- the opaque columns take the hot instance from 168 VGPRs + 9 spills to 147
// and bring all seven panel loads to the top of the step, and it is 10 % SLOWER (with the row-ahead prefetch 85.0): measured, not
// adopted. WBAR 0 / 1 / 2 = 81.4 / 81.3 / 82.2: the wave barrier is free, the wavefront fence is not.
#ifndef QS_BP4_WAVES
#define QS_BP4_WAVES 1      /* binary_partial at 4 depth bits is held to 4 waves per SIMD (128 VGPRs + 6 spilled; it would take 133): 512 taxa x
                             * 1500 trees with 10 % of the taxa dropped 67.0 / 67.2 -> 63.7 / 63.4 ms (round 5; at 5 bits, 152 VGPRs, round 4 measured no gain) */
#endif
#ifndef QS_GEN_OPAQUE
#define QS_GEN_OPAQUE 0     /* 1: general modes re-define the lane's LDS columns per step behind an empty asm (see the step) */
#endif
#ifndef QS_GEN_WBAR
#define QS_GEN_WBAR 1       /* single-buffered LDS image: 0 = nothing between a step's reads and its writes, 1 = wave barrier (orders every instruction), 2 = wavefront-scope fence (orders memory operations only) */
#endif
template <int KIND> __device__ __forceinline__ void lds_order() {
    if (KIND == 1) __builtin_amdgcn_wave_barrier();
    else if (KIND == 2) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
#ifndef QS_GEN_PREFETCH
#define QS_GEN_PREFETCH 0   /* 1: general modes request the next d-row's R elements one row ahead; 2: ... behind a scheduling barrier */
#endif
#define QS_BS3_OCC __attribute__((amdgpu_waves_per_eu(QS_BS3_WAVES, QS_BS3_WAVES)))


// one table tuple = three cells: moved with ONE 12-byte access for u32 cells (global_load/store_dwordx3; a tuple
// is 4-byte aligned) instead of three 4-byte ones -- the epilogue of a wave is 16 tuples per lane
template <typename CT> struct Tuple3 { uint32_t a, b, c; };
template <typename CT> __device__ __forceinline__ Tuple3<CT> load_tuple(const CT *p) {
    Tuple3<CT> t;
    // read once per launch and never again: non-temporal, so the tuples do not evict panel lines from the L2
    // (profiles/r02_experiments.md: -1.3 % at 512 taxa, neutral at 256; non-temporal STORES measured slower)
    if (sizeof(CT) == 4) { const qs_u32x3 v = __builtin_nontemporal_load(reinterpret_cast<const qs_u32x3_a4 *>(p)); t.a = v.x; t.b = v.y; t.c = v.z; }
    else { t.a = p[0]; t.b = p[1]; t.c = p[2]; }
    return t;
}
template <typename CT> __device__ __forceinline__ void store_tuple(CT *p, uint32_t a, uint32_t b, uint32_t c) {
    if (sizeof(CT) == 4) { qs_u32x3 v; v.x = a; v.y = b; v.z = c; *reinterpret_cast<qs_u32x3_a4 *>(p) = v; }
    else { p[0] = (CT)a; p[1] = (CT)b; p[2] = (CT)c; }
}

// waves per SIMD the register allocation aims at: 4 (<= 128 VGPRs); the two instances that do not fit without spilling
// (7 depth bits, binary: 8-plane operands in two a-columns; 4 bits, general) take 3 -- a spill means scratch memory
// Deep trees (8..10 depth bits: ladders of up to ~2000 taxa) carry 9..12-word operands: 2 waves per SIMD.
// (8 and 9 bits in the binary / general modes: 3 waves -- <= 168 VGPRs and 42-46 KB of LDS per workgroup.)
template <int B, int MODE> constexpr int bs3_waves() {
    constexpr bool gen = MODE == MODE_GENERAL_FULL || MODE == MODE_PARTIAL;
    if (gen && QS_PROBE_GEN_WAVES && B <= (MODE == MODE_PARTIAL ? QS_GEN3_MAXB - 1 : QS_GEN3_MAXB)) return QS_PROBE_GEN_WAVES;
    if (gen) return B <= (MODE == MODE_PARTIAL ? QS_GEN3_MAXB - 1 : QS_GEN3_MAXB) ? 3 : 2;   // two a-columns + three counters per quartet: 172-192 VGPRs unconstrained (B <= 6)
    if (B >= 10 || (B >= 7 && MODE == MODE_BINARY_PARTIAL)) return 2;
    if (MODE == MODE_BINARY_PARTIAL) return (B <= 4 && QS_BP4_WAVES) ? 4 : 3;
    if (B >= 8) return 3;
    if (B == 7 && MODE == MODE_BINARY_FULL) return 3;
    return QS_BS3_WAVES;
}

constexpr int kWavesPerBlock = kCountThreads / kWave;   // 4: a workgroup = four independent wave tiles (consecutive in the launch order)

// ======================================================================================
// the tile of a wave
// ======================================================================================
// Wave = (d-block of 8 largest ids, third id c, tile of 16 a x 8 b, a < b < c). Off-diagonal tiles: lane (ia, ib) owns
// a1 = 8 blk0 + ia, a2 = 8 blk1 + ia and b = 8 blkB + ib; diagonal tiles (a and b from the same block) pack two diagonal
// blocks per wave, one a per lane.
struct Bs3Tile {
    uint32_t lane;
    uint32_t c, d0, d1, jlo, jhi;       // third id; the d-block [d0, d1); its live slots [jlo, jhi) (d > c)
    uint32_t blk0, blk1, blkB;          // the two a-blocks (blk1 = 0xFFFFFFFF: absent) and the b-block
    bool offdiag, has_a2;               // wave-uniform
    bool v1, v2;                        // the lane's quartets (a1,b,c,.) / (a2,b,c,.) exist
    uint32_t colA1, colA2, colB16;      // the lane's columns in the LDS image (colB16: in the 16-column image of the binary modes; the
                                        // general modes keep the b-block in columns 16..23 of off-diagonal tiles)
    uint32_t pi1, pi2;                  // pair index C(b,2) + a of (a1,b) / (a2,b); 0 when the quartet does not exist
};

// launch slot -> tile (wave-uniform), false = this wave has no tile. d-block k counts down from the top of the shard.
// Workgroups with the same blockIdx % 8 share an XCD (observed dispatch rule, used for speed only). With xcd_remap bit 0
// every XCD walks its own contiguous eighth of the tile list, so its 4 MB L2 holds the panel rows of one (d-block, c)
// neighbourhood instead of all eight: +4 % at 512 taxa, +1.5 % at 256, -2 % at 128 (there the whole panel fits every L2),
// so the launcher sets it from 200 taxa on.
__device__ __forceinline__ bool bs3_decode_tile(Bs3Tile &t, uint32_t d_start, uint32_t d_hi, uint32_t n_dblk, uint32_t total_tiles,
                                                const uint32_t *__restrict__ dprefix, const uint32_t *__restrict__ cprefix,
                                                uint32_t xcd_remap, const uint32_t *__restrict__ perm) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t lb = blockIdx.x;
    if (xcd_remap & 1u) {
        const uint32_t nb = gridDim.x, q8 = nb / 8, r8 = nb % 8, xcd = lb % 8, y = lb / 8;
        lb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + y; // bijective on [0, nb)
    }
    uint32_t tile = lb * kWavesPerBlock + wave;
    if (tile >= total_tiles) return false;
    if (perm) tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)perm[tile]);
    uint32_t dp_k, cp_c;
    const uint32_t k = wave_search_le(dprefix, 0, n_dblk, tile, lane, dp_k);
    const uint32_t local = tile - dp_k;
    const uint32_t d1 = d_hi - k * kDB;
    const uint32_t d0 = d1 > d_start + kDB ? d1 - kDB : d_start;
    const uint32_t c = wave_search_le(cprefix, 2, d1 - 1, local, lane, cp_c);
    const uint32_t T = (c + kTB - 1) / kTB;
    const uint32_t n_off = (T * T) / 4;
    const uint32_t tl = local - cp_c;
    const bool offdiag = tl < n_off;
    uint32_t blk0, blk1, blkB;
    uint32_t a1, a2, b, colA1, colA2, colB;
    if (offdiag) {
        uint32_t Bk = (uint32_t)(2.0f * sqrtf((float)tl + 1.0f)); // largest Bk with floor(Bk^2 / 4) <= tl
        while ((Bk * Bk) / 4 > tl) --Bk;
        while (((Bk + 1) * (Bk + 1)) / 4 <= tl) ++Bk;
        const uint32_t j = tl - (Bk * Bk) / 4;
        blk0 = 2 * j; blk1 = (2 * j + 1 < Bk) ? 2 * j + 1 : 0xFFFFFFFFu; blkB = Bk;
        const uint32_t ia = lane & (kTA - 1), ib = lane / kTA;
        colA1 = ia; colA2 = kTA + ia; colB = ib;
        a1 = blk0 * kTA + ia;
        a2 = blk1 == 0xFFFFFFFFu ? 0xFFFFFFFFu : blk1 * kTA + ia;
        b = Bk * kTB + ib;
    } else {
        const uint32_t kd = tl - n_off;
        blk0 = 2 * kd; blk1 = 2 * kd + 1; blkB = blk0;
        const uint32_t h = lane >> 5, q = lane & 31;
        uint32_t ia = 0, ib = 1;
        if (q < 28) unrank2(q, ia, ib);
        colA1 = h * kTA + ia; colA2 = colA1; colB = h * kTA + ib;
        a1 = (h ? blk1 : blk0) * kTA + ia; a2 = 0xFFFFFFFFu;
        b = q < 28 ? (h ? blk1 : blk0) * kTA + ib : 0xFFFFFFFFu;
    }
    t.lane = lane;
    t.c = c; t.d0 = d0; t.d1 = d1;
    t.jlo = c >= d0 ? c + 1 - d0 : 0u; t.jhi = d1 - d0;
    t.blk0 = blk0; t.blk1 = blk1; t.blkB = blkB;
    t.offdiag = offdiag;
    t.has_a2 = offdiag && blk1 != 0xFFFFFFFFu;
    t.v1 = (a1 < b) && (b < c);
    t.v2 = t.has_a2 && (a2 < b) && (b < c);
    t.colA1 = colA1; t.colA2 = colA2; t.colB16 = colB;
    t.pi1 = t.v1 ? (uint32_t)binom2(b) + a1 : 0u;
    t.pi2 = t.v2 ? (uint32_t)binom2(b) + a2 : 0u;
    return true;
}

// ======================================================================================
// one class of trees past one tile
// ======================================================================================
// What depends on (depth bits, mode) of a class: the words of a panel element and of an LDS element, the columns of the LDS
// image, one or two images per wave.
//   BIN: binary trees (two comparisons decide a quartet); GEN = !BIN: multifurcations possible, a third comparison tells ad|bc
//   from "unresolved". PART: elements carry a presence word (taxa may be missing). binary_partial = BIN and PART: the third
//   topology is what is left of the trees that hold all four taxa, v & ~(gt | lt) -- no third comparison (gene trees: binary,
//   with missing taxa). Every mode uses the same tiling: a lane owns TWO a-columns.
template <int B, int MODE> struct Bs3Layout {
    static constexpr bool BIN = MODE == MODE_BINARY_FULL || MODE == MODE_BINARY_PARTIAL;
    static constexpr bool GEN = !BIN;
    static constexpr bool PART = MODE == MODE_PARTIAL || MODE == MODE_BINARY_PARTIAL;
    static constexpr bool BP = MODE == MODE_BINARY_PARTIAL;
    static constexpr int RC = s3_cols(GEN), kS3Row0 = s3_row0(GEN), kS3Slots = s3_slots(GEN);
    // LDS operations of one wave execute in order and no wave reads another's image, so ONE buffer per wave would do; the
    // binary modes keep two (the stores of the next image then carry no dependence on the current image's reads)
    static constexpr int NBUF = GEN ? 1 : 2;
    static constexpr int NB = B + 1;
    static constexpr int NWP = B + (PART ? 1 : 0);                 // words of a compact panel element (planes [+ presence])
    static constexpr int NW = NWP < 4 ? 4 : NWP;
    static constexpr int RW = NB + (PART ? 1 : 0);                 // words of an LDS element (R has B+1 planes [+ presence])
    static constexpr int HW = RW <= 5 ? 1 : (RW == 6 ? 2 : (RW <= 8 ? 4 : (RW == 9 ? 5 : (RW == 10 ? 6 : 8)))); // upper words of an LDS slot
    static constexpr int PW = B + 1;                               // where the presence word travels in an LDS element
    static_assert(B <= kMaxDepthBits && RW <= kBitWords, "at most 10 depth bits (11 planes + presence in 12 words)");
    static constexpr int kImg = (HW == 5 || HW == 6) ? 2 * kS3Slots + (kS3Slots * (HW - 4) + 3) / 4 : kS3Slots + (kS3Slots * HW + 3) / 4; // uint4 per wave and buffer
    static constexpr int kLdsUint4 = NBUF * kImg;                  // uint4 of LDS per wave
};

// All n_groups 32-tree groups of ONE class (panel P: compact elements of Bs3Layout<B, MODE>::NW words) streamed past the tile:
// the counters are the caller's registers and are only ever added to.
//   x0 / x1 = ab|cd, ac|bd of (a1,b); y0 / y1 the same of (a2,b); z0 / z1: general modes = ad|bc of (a1,b) / (a2,b); binary_partial
//   = the trees that hold all four taxa (the third count is that minus the other two); binary_full: untouched (the third count is
//   the number of trees minus the other two).
// lds: this wave's region, Bs3Layout<B, MODE>::kLdsUint4 uint4.
// VAR: -1 = the wave picks the instance of its tile here (has_a2 / all 8 d slots live / diagonal); 0..3 = the caller has done
// so (bs3_tile_variant) and runs several segments under ONE dispatch (the fused kernel: the counters then flow through straight-line
// code from segment to segment instead of through a four-way merge after every segment).
__device__ __forceinline__ int bs3_tile_variant(const Bs3Tile &t) {
    const bool full = t.jlo == 0 && t.jhi == (uint32_t)kDB;
    return t.has_a2 ? (full ? 0 : 1) : (t.offdiag ? 2 : 3);
}
template <int B, int MODE, int VAR = -1>
__device__ __forceinline__ void bs3_segment(const Bs3Tile &t, const uint4 *__restrict__ P, uint32_t npairs, uint32_t n_groups, uint32_t xcd_remap,
                                            uint4 *lds, uint32_t (&x0)[kDB], uint32_t (&x1)[kDB], uint32_t (&y0)[kDB], uint32_t (&y1)[kDB],
                                            uint32_t (&z0)[kDB], uint32_t (&z1)[kDB]) {
    using Lay = Bs3Layout<B, MODE>;
    constexpr bool BIN = Lay::BIN, GEN = Lay::GEN, PART = Lay::PART, BP = Lay::BP;
    constexpr int RC = Lay::RC, kS3Row0 = Lay::kS3Row0, kS3Slots = Lay::kS3Slots, NBUF = Lay::NBUF, NB = Lay::NB, NW = Lay::NW, HW = Lay::HW, PW = Lay::PW;
    uint4 *buf0 = lds, *buf1 = lds + (NBUF - 1) * Lay::kImg;
    const uint32_t lane = t.lane, c = t.c, d0 = t.d0, d1 = t.d1, jlo = t.jlo, jhi = t.jhi, blk0 = t.blk0, blk1 = t.blk1, blkB = t.blkB;
    const bool offdiag = t.offdiag, has_a2 = t.has_a2, v1 = t.v1, v2 = t.v2;
    const uint32_t pi1 = t.pi1, pi2 = t.pi2;
    // staged columns: M[x,c] of blk0 in columns 0..7 and of blk1 in 8..15; the binary modes stage the R elements of the b-block in
    // columns 0..7, the general modes R of blk0 / blk1 / blkB in columns 0..7 / 8..15 / 16..23 (they also compare R of (a,d)) and
    // M[b,c] in row columns 16..23. Diagonal tiles: 16 columns in every mode.
    const uint32_t colA1_ = t.colA1, colA2_ = t.colA2, colB_ = t.colB16 + ((GEN && offdiag) ? 2u * kTA : 0u);
    const uint32_t n_r = offdiag ? (GEN ? 3u : 1u) : 2u;        // blocks of R elements staged per d-row (wave-uniform)

    // ---- loop-invariant offsets inside one tree group (16 bytes per pair in the lo array), and LDS slots ----
    const uint32_t r_j = lane >> 3, r_col = lane & 7, dE = d0 + r_j;
    const bool dok = dE < d1 && dE > c;
    // lane (r_j, r_col) stages the R elements of d-row r_j and column r_col of up to three blocks: slots r_j * RC + r_col + {0, 8, 16}
    const uint32_t bE0 = (n_r == 1 ? blkB : blk0) * kTB + r_col, bE1 = blk1 * kTB + r_col, bE2 = blkB * kTB + r_col;
    const bool ok0 = dok && bE0 < c, ok1 = dok && n_r >= 2 && blk1 != 0xFFFFFFFFu && bE1 < c, ok2 = dok && n_r == 3 && bE2 < c;
    const uint32_t rowd = (uint32_t)binom2(dE);
    const uint32_t x0off = ok0 ? (rowd + bE0) * 16u : kS3Inv, x1off = ok1 ? (rowd + bE1) * 16u : kS3Inv;
    const uint32_t x2off = ok2 ? (rowd + bE2) * 16u : kS3Inv;
    const uint32_t yoff = dok ? (rowd + c) * 16u : kS3Inv;      // M[c,d]: shared by all R elements of the lane
    const uint32_t slot0 = r_j * RC + r_col, slot1 = slot0 + 8, slot2 = slot0 + 16;
    // the RC elements M[x,c] (x in blk0, blk1 and -- general modes -- the b-block) are staged by ALL lanes, in copies (lane & 15
    // resp. lane & 31 picks the element; the copies load the same address and store the same value to the same slot). With the
    // store under `if (lane < 16)` the compiler sank the load into that branch, i.e. behind the whole compute of the step, and
    // waited for it with vmcnt(0) right there: one exposed memory latency per 32-tree step (knock-out: profiles/r02_experiments.md).
    const uint32_t lrow = GEN ? ((lane & 31) < 24 ? (lane & 31) : (lane & 31) - 8) : (lane & 15);   // (0..23: lanes 24..31 repeat 16..23)
    uint32_t xa = 0xFFFFFFFFu;
    if (lrow < 8) xa = blk0 * kTA + lrow;
    else if (lrow < 16) { if (blk1 != 0xFFFFFFFFu) xa = blk1 * kTA + (lrow - 8); }
    else if (lrow < (uint32_t)RC && offdiag) xa = blkB * kTB + (lrow - 16);
    const uint32_t rowoff = xa < c ? ((uint32_t)binom2(c) + xa) * 16u : kS3Inv;
    const uint32_t rowslot = kS3Row0 + lrow;
    const uint32_t ab1off = v1 ? pi1 * 16u : kS3Inv, ab2off = v2 ? pi2 * 16u : kS3Inv;
    const uint32_t group_bytes = npairs * (uint32_t)(NW * 4), hi_base = npairs * 16u;

    auto rsrc_of = [&](uint32_t g) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const char *>(P) + (size_t)g * group_bytes), 0, (int)group_bytes, 0x00020000);
    };
    // compact panel element -> planes in w[0..B-1], presence (partial) in w[kPres]
    auto gload = [&](__amdgpu_buffer_rsrc_t r, uint32_t voff) {
        Planes p = buf_load_planes<NW>(r, voff, hi_base);
        if (PART && B != kPres) { p.w[kPres] = p.w[B]; p.w[B] = 0; }
        return p;
    };
    // LDS element: words 0..B [+ presence at word B+1]
    auto lstore = [&](uint4 *buf, uint32_t slot, Planes x) {
        if (PART && PW != kPres) x.w[PW] = x.w[kPres];
        lds_store_hw<HW>(buf, slot, kS3Slots, x);
    };
    auto lload = [&](const uint4 *buf, uint32_t slot) {
        Planes r = lds_load_hw<HW>(buf, slot, kS3Slots);
        if (PART && PW != kPres) { r.w[kPres] = r.w[PW]; r.w[PW] = 0; }
        return r;
    };
    auto row0_load = [&](const uint4 *buf, uint32_t col) {
        if (B <= 4 && !PART) {
            const uint4 lo = buf[kS3Row0 + col];
            Planes r;
            r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
#pragma unroll
            for (int k = 4; k < kBitWords; ++k) r.w[k] = 0;
            return r;
        }
        return lload(buf, kS3Row0 + col);
    };
    struct Staged { Planes x0, x1, x2, y, row; };

    // one 32-tree step: request group g_next into (st, abn1, abn2), count group g from (cur, abc1, abc2), then
    // turn the requested elements into the LDS image `nxt`
    auto step = [&](uint32_t g_next, const uint4 *cur, uint4 *nxt, const Planes &abc1, const Planes &abc2, Planes &abn1,
                    Planes &abn2, auto a2_tag, auto full_tag, auto nr_tag) {
        constexpr bool A2 = decltype(a2_tag)::value, FULL = decltype(full_tag)::value;
        constexpr int NR = decltype(nr_tag)::value;                   // blocks of R elements this tile stages (n_r)
        const __amdgpu_buffer_rsrc_t r = rsrc_of(g_next);
        // The four waves of a workgroup are independent tiles, but consecutive ones: same b-block, c and d-block. Keeping
        // them in step (one s_barrier per 32-tree group; every wave runs the same number of steps, a wave without a tile
        // has ended and does not count) lets their identical panel loads -- M[b,d], M[c,d], M[b,c] -- meet in the L1
        // instead of becoming separate requests to the L2, whose number bounds the kernel together with VALU issue
        // (profiles/r02_experiments.md): -5 % at 512 taxa, -9 % with NNI trees, -5 % on a 1024-taxon shard, -1 % at 256;
        // at 128 taxa and below (the panel sits in the L2 anyway) it costs 4-6 %, so the launcher sets bit 1 from 200 on.
        if (xcd_remap & 2u) __builtin_amdgcn_s_barrier();
        Staged st;
        st.x0 = gload(r, x0off);
        st.y = gload(r, yoff);
        if (NR >= 2) st.x1 = gload(r, x1off);
        if (NR == 3) st.x2 = gload(r, x2off);
        st.row = gload(r, rowoff);
        abn1 = gload(r, ab1off);
        if (A2) abn2 = gload(r, ab2off);

        // QS_GEN_OPAQUE (A/B only, off): in the straight-line (FULL) instance of the general modes LLVM hoists every (d-row, column)
        // LDS address out of the loop as a value of its own -- ~48 address registers in an instance that lives at the 168-VGPR
        // limit, which pushes four of the seven panel loads behind the last d-row. With the columns re-defined per step behind an
        // empty asm the rows become immediate offsets from three bases (147 VGPRs, no spills, loads at the top) -- and the kernel
        // runs 10 % slower (table above): what looks like an exposed L2 round trip is covered by the other two waves.
        uint32_t colA1 = colA1_, colA2 = colA2_, colB = colB_;
        if (GEN && QS_GEN_OPAQUE) asm volatile("" : "+v"(colA1), "+v"(colA2), "+v"(colB));
        // binary modes: the compare chains run at issue priority 1, the request / staging part of the step at 0 (round 6: -1 % on
        // configs[2], -1.2 % for binary_partial; the general modes LOSE 0.6-4.5 % with it and keep priority 0: profiles/r06_experiments.md 8)
        if (BIN && QS_BIN_PRIO) __builtin_amdgcn_s_setprio(1);
        const Planes L1 = sub_biased<B>(abc1, row0_load(cur, colA1)); // M[a1 b] - M[a1 c] + 2^B
        Planes L2 = L1, G1 = L1, G2 = L1;
        if (A2) L2 = sub_biased<B>(abc2, row0_load(cur, colA2));      // M[a2 b] - M[a2 c] + 2^B
        if (GEN) {                                                    // M[ab] - M[bc] + 2^B: the other side of the third comparison
            const Planes rb = row0_load(cur, colB);
            G1 = sub_biased<B>(abc1, rb);
            if (A2) G2 = sub_biased<B>(abc2, rb);
        }
        // general modes, hot instance: the three R elements of d-row j + 1 are requested before row j is compared (the LDS
        // round trip is ~100+ cycles, a row's chains ~80, and only 3 waves per SIMD are there to cover the difference)
        constexpr bool PFR = QS_GEN_PREFETCH && MODE == MODE_GENERAL_FULL && B <= 4 && FULL && A2;
        Planes nRb = L1, nRa = L1, nRa2 = L1;
        if (PFR) { nRb = lload(cur, colB); nRa = lload(cur, colA1); nRa2 = lload(cur, colA2); }
#pragma unroll
        for (int j = 0; j < QS_PROBE_ROWS; ++j) {   // (QS_PROBE_ROWS < 8: timing probe of a narrower d-block, profiles/r06_experiments.md 4 -- NOT a product build)
            if (FULL || ((uint32_t)j >= jlo && (uint32_t)j < jhi)) { // wave-uniform
                const Planes pRb = nRb, pRa = nRa, pRa2 = nRa2;
                if (PFR && j + 1 < kDB) {
                    nRb = lload(cur, (j + 1) * RC + colB); nRa = lload(cur, (j + 1) * RC + colA1); nRa2 = lload(cur, (j + 1) * RC + colA2);
                    if (QS_GEN_PREFETCH == 2) __builtin_amdgcn_sched_barrier(0);
                }
                const Planes Rb = PFR ? pRb : lload(cur, j * RC + colB);   // M[bd] - M[cd] + 2^B
                uint32_t gt, lt;
                cmp_planes<NB>(L1, Rb, gt, lt);
                if (BIN) {
                    if (BP) {
                        const uint32_t v = L1.w[kPres] & Rb.w[kPres];   // a1, b, c, d all present
                        gt &= v; lt &= v;
                        popc_acc(v, z0[j]);   // trees that hold all four: a binary one resolves the quartet, so ad|bc = this count - the other two (epilogue)
                    }
                    popc_acc(gt, x0[j]);
                    popc_acc(lt, x1[j]);
                    if (A2) {
                        uint32_t gt2, lt2;
                        cmp_planes<NB>(L2, Rb, gt2, lt2);
                        if (BP) {
                            const uint32_t v2 = L2.w[kPres] & Rb.w[kPres];
                            gt2 &= v2; lt2 &= v2;
                            popc_acc(v2, z1[j]);
                        }
                        popc_acc(gt2, y0[j]);
                        popc_acc(lt2, y1[j]);
                    }
                } else {
                    const Planes Ra = PFR ? pRa : lload(cur, j * RC + colA1);    // M[a1 d] - M[cd] + 2^B
                    // [S3 > S1]. In a tree the two smaller of the three sums are equal (four-point condition), so S3 > S1 already
                    // implies S1 == S2: no masking with ~(gt | lt) (round 4: one instruction per quartet less)
                    uint32_t g3 = gt_planes<NB>(Ra, G1);
                    if (PART) {
                        const uint32_t v = L1.w[kPres] & Rb.w[kPres]; // a1, b, c, d all present
                        gt &= v; lt &= v; g3 &= v;
                    }
                    popc_acc(gt, x0[j]);
                    popc_acc(lt, x1[j]);
                    popc_acc(g3, z0[j]);
                    if (A2) {                                        // the second a-column against the same R of (b,d)
                        const Planes Ra2 = PFR ? pRa2 : lload(cur, j * RC + colA2);
                        uint32_t gt2, lt2;
                        cmp_planes<NB>(L2, Rb, gt2, lt2);
                        uint32_t h3 = gt_planes<NB>(Ra2, G2);
                        if (PART) {
                            const uint32_t v2 = L2.w[kPres] & Rb.w[kPres];
                            gt2 &= v2; lt2 &= v2; h3 &= v2;
                        }
                        popc_acc(gt2, y0[j]);
                        popc_acc(lt2, y1[j]);
                        popc_acc(h3, z1[j]);
                    }
                }
            }
        }
        // single-buffered image (cur == nxt): every read of this step above, every write of the next image below. LDS operations
        // of a wave execute in program order; the scheduling barrier keeps the compiler from moving a store above a load it
        // cannot prove disjoint (no instruction is emitted)
        if (BIN && QS_BIN_PRIO) __builtin_amdgcn_s_setprio(0);   // (kept high through the staging stores as well: +2 %: worse)
        if (NBUF == 1) lds_order<QS_GEN_WBAR>();
        lstore(nxt, slot0, sub_biased<B>(st.x0, st.y));
        if (NR >= 2) lstore(nxt, slot1, sub_biased<B>(st.x1, st.y));
        if (NR == 3) lstore(nxt, slot2, sub_biased<B>(st.x2, st.y));
        if (B <= 4 && !PART) nxt[rowslot] = make_uint4(st.row.w[0], st.row.w[1], st.row.w[2], st.row.w[3]);
        else lstore(nxt, rowslot, st.row);
        if (NBUF == 1) lds_order<QS_GEN_WBAR>();   // ... and the next step's reads stay behind these stores
    };

    auto run = [&](auto a2_tag, auto full_tag, auto nr_tag) {
        constexpr bool A2 = decltype(a2_tag)::value;
        constexpr int NR = decltype(nr_tag)::value;
        Planes abA1, abA2, abB1, abB2;
#pragma unroll
        for (int w = 0; w < kBitWords; ++w) abA2.w[w] = abB2.w[w] = 0;
        {   // group 0 -> buf0 / set A
            const __amdgpu_buffer_rsrc_t r = rsrc_of(0);
            const Planes y = gload(r, yoff);
            lstore(buf0, slot0, sub_biased<B>(gload(r, x0off), y));
            if (NR >= 2) lstore(buf0, slot1, sub_biased<B>(gload(r, x1off), y));
            if (NR == 3) lstore(buf0, slot2, sub_biased<B>(gload(r, x2off), y));
            const Planes row = gload(r, rowoff);
            lstore(buf0, rowslot, row);
            abA1 = gload(r, ab1off);
            if (A2) abA2 = gload(r, ab2off);
        }
        const uint32_t g_last = n_groups - 1;
        // Both steps of a trip are unconditional and the odd last group is peeled: with the second step under
        // `if (g + 1 < n_groups)` LLVM sinks the first step's M[ab] loads (used only by the second) into that branch,
        // i.e. behind the whole compute of the first step and directly in front of their first use -- one exposed L2
        // round trip per two steps, taken by all waves of a SIMD at about the same time.
        uint32_t g = 0;
        for (; g + 2 <= n_groups; g += 2) {
            step(g + 1, buf0, buf1, abA1, abA2, abB1, abB2, a2_tag, full_tag, nr_tag);
            step(min(g + 2, g_last), buf1, buf0, abB1, abB2, abA1, abA2, a2_tag, full_tag, nr_tag); // past the end: re-reads the last group (unused)
        }
        if (g < n_groups) step(g_last, buf0, buf1, abA1, abA2, abB1, abB2, a2_tag, full_tag, nr_tag);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>; using N3 = std::integral_constant<int, 3>;
    using NO = std::integral_constant<int, BIN ? 1 : 3>;   // R blocks of an off-diagonal tile
    if (VAR == 0) { run(T_{}, T_{}, NO{}); return; }
    if (VAR == 1) { run(T_{}, F_{}, NO{}); return; }
    if (VAR == 2) { run(F_{}, F_{}, NO{}); return; }
    if (VAR == 3) { run(F_{}, F_{}, N2{}); return; }
    const bool full = jlo == 0 && jhi == (uint32_t)kDB;
    if (BIN) {
        if (has_a2) { if (full) run(T_{}, T_{}, N1{}); else run(T_{}, F_{}, N1{}); }
        else if (offdiag) run(F_{}, F_{}, N1{});
        else run(F_{}, F_{}, N2{});
    } else {
        if (has_a2) { if (full) run(T_{}, T_{}, N3{}); else run(T_{}, F_{}, N3{}); }
        else if (offdiag) run(F_{}, F_{}, N3{});
        else run(F_{}, F_{}, N2{});
    }
}

// The tile's tuples to the table. rank of {a,b,c,d} = C(d,4) + C(c,3) + C(b,2) + a; along the d slots C(d+1,4) = C(d,4) + C(d,3)
// etc., so the 64-bit products and divisions are done once per wave instead of once per slot. THIRD: where the third cell
// comes from -- 0: trees - n0 - n1 (binary_full), 1: z - n0 - n1 (binary_partial: z = trees holding all four), 2: z itself.
// WIRE (binary_full with QS_COUNT_WIRE16X2): instead of the [rank][3] table ONE word n0 | n1 << 16 per tuple -- the two-cell
// format the multi-GPU collective moves (n2 = trees - n0 - n1 is restored by qs_unpack16x2). No table write, no pack kernel.
template <typename CT, int THIRD>
__device__ __forceinline__ void bs3_store(const Bs3Tile &t, uint64_t rank_lo, CT *__restrict__ table, uint32_t *__restrict__ overflow_flag,
                                          uint32_t overwrite, uint32_t m_trees, uint32_t *__restrict__ wire, const uint32_t (&x0)[kDB],
                                          const uint32_t (&x1)[kDB], const uint32_t (&y0)[kDB], const uint32_t (&y1)[kDB],
                                          const uint32_t (&z0)[kDB], const uint32_t (&z1)[kDB]) {
    const uint32_t c = t.c, d0 = t.d0, d1 = t.d1, pi1 = t.pi1, pi2 = t.pi2;
    const bool v1 = t.v1, v2 = t.v2;
    uint64_t bd4 = binom4(d0), bd3 = binom3(d0), bd2 = binom2(d0);
    const uint64_t rcb = binom3(c) - rank_lo;
    if (THIRD == 0 && wire) {
#pragma unroll
        for (int j = 0; j < kDB; ++j) {
            const uint32_t d = d0 + j;
            const uint64_t base = bd4 + rcb;
            bd4 += bd3; bd3 += bd2; bd2 += d;
            if (d < d1 && d > c) {
                if (v1) {
                    uint32_t w = x0[j] | (x1[j] << 16);
                    if (!overwrite) w += wire[base + pi1];      // carry-free: the host keeps the totals below 2^16
                    wire[base + pi1] = w;
                }
                if (v2) {
                    uint32_t w = y0[j] | (y1[j] << 16);
                    if (!overwrite) w += wire[base + pi2];
                    wire[base + pi2] = w;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kDB; ++j) {
        const uint32_t d = d0 + j;
        const uint64_t base = bd4 + rcb;
        bd4 += bd3; bd3 += bd2; bd2 += d;
        if (d < d1 && d > c) {
            if (v1) {
                const uint64_t idx = (base + pi1) * 3;
                uint32_t w0 = x0[j], w1 = x1[j], w2 = THIRD == 0 ? m_trees - x0[j] - x1[j] : THIRD == 1 ? z0[j] - x0[j] - x1[j] : z0[j];
                if (!overwrite) { const Tuple3<CT> tp = load_tuple(table + idx); w0 += tp.a; w1 += tp.b; w2 += tp.c; }
                if (sizeof(CT) == 2 && ((w0 | w1 | w2) > 0xFFFFu)) atomicOr(overflow_flag, 1u);
                store_tuple(table + idx, w0, w1, w2);
            }
            if (v2) {
                const uint64_t idx = (base + pi2) * 3;
                uint32_t w0 = y0[j], w1 = y1[j], w2 = THIRD == 0 ? m_trees - y0[j] - y1[j] : THIRD == 1 ? z1[j] - y0[j] - y1[j] : z1[j];
                if (!overwrite) { const Tuple3<CT> tp = load_tuple(table + idx); w0 += tp.a; w1 += tp.b; w2 += tp.c; }
                if (sizeof(CT) == 2 && ((w0 | w1 | w2) > 0xFFFFu)) atomicOr(overflow_flag, 1u);
                store_tuple(table + idx, w0, w1, w2);
            }
        }
    }
}

} // namespace qs
