// qs_count_fused.hip -- the classes of a mixed batch in ONE launch of the bit-sliced count kernel.
//
// A batch is counted class by class (class = kernel mode x depth bits per TREE, qs_abi.hip plan_classes): full binary trees,
// binary trees with missing taxa, multifurcating trees and trees with both take different 32-tree steps (2(B+1)+2 ... 3(B+1)+7
// instructions per quartet). With one launch per class every class costs a pass over the table (34 GB at 512 taxa: the first
// stores it, every further one reads and writes it) plus the fixed cost of ~3 M waves (tile decode, prologue, epilogue): 25 ms
// per class at 512 taxa, where the whole step of a 1500-tree gene-tree batch is 60-90 ms (profiles/r05_experiments.md 1).
// The reference's loop is shape-independent (QuartetCounterLookup.hpp:65-106,166-188): it never pays for a mix of shapes.
//
// Here the classes that share their depth bits B run as SEGMENTS of one launch: a wave decodes its tile once, streams the
// tree groups of the binary_full class past it with that class's step, then the binary_partial groups with theirs, then
// general_full, then partial -- every class keeps its own panel layout and LDS image (bs3_segment<B, MODE>, qs_bitslice3.hpp) --
// and the counters stay in registers across the segments: ONE epilogue, ONE pass over the table.
//   counters: x / y = the two topologies every mode counts directly. The third one differs: the binary modes do not count it (a
//   binary tree resolves every quartet it holds: third = trees holding all four - the other two), the general modes count it.
//   After the binary segments z := (binary_full trees) + z (the binary_partial segment counted "holds all four" there) - x0 - x1
//   turns z into a plain count, which the general segments then add to.
// Two kernels per (B, cell type): binary classes only (binary_full + binary_partial: the occupancy of the binary_partial
// instance) and any mix with a general class (the occupancy of the partial instance; absent segments are skipped by a
// wave-uniform branch). The four waves of a workgroup walk the same segments with the same number of steps, so the step's
// s_barrier (waves in step, qs_bitslice3.hpp) stays balanced.
// Integer / bit work only: no MFMA.
#include "qs_bitslice3.hpp"

namespace qs {

#ifndef QS_FUSED_SEGMASK
#define QS_FUSED_SEGMASK 15   /* debug: segments compiled in (bit 0 binary_full .. bit 3 partial) */
#endif
#ifndef QS_FUSED_BIN4_WAVES
#define QS_FUSED_BIN4_WAVES 0   /* 0 = as the binary_partial instance of the same depth bits (bs3_waves); else waves per SIMD of the binary-only fused kernel at 4 bits */
#endif

#ifndef QS_FUSED_BIN5_WAVES
#define QS_FUSED_BIN5_WAVES 4   /* binary-only fused kernel at 5 bits: 121-127 VGPRs without spills (the one-class binary_partial instance needs 140-152: 3 waves);
                                 * at 6 bits a fourth wave would cost 6-13 spilled registers, some inside the hot loop: not taken */
#endif
#ifndef QS_FUSED_GEN_WAVES
#define QS_FUSED_GEN_WAVES 0    /* 0 = as the partial instance of the same depth bits; else waves per SIMD of the fused kernel with general segments (4 .. 5 bits) */
#endif
template <int B, bool GENK> constexpr int fused_waves() {
    if (GENK && B <= 5 && QS_FUSED_GEN_WAVES) return QS_FUSED_GEN_WAVES;
    if (GENK) return bs3_waves<B, MODE_PARTIAL>();
    if (B <= 4 && QS_FUSED_BIN4_WAVES) return QS_FUSED_BIN4_WAVES;
    if (B == 5) return QS_FUSED_BIN5_WAVES;
    if (B == 7) return 3;   // (the one-class binary_partial instance takes 2 at 7 bits; under one dispatch the pair fits 168 VGPRs)
    return bs3_waves<B, MODE_BINARY_PARTIAL>();
}
template <int B, bool GENK> constexpr int fused_lds_uint4() {
    int m = Bs3Layout<B, MODE_BINARY_FULL>::kLdsUint4;
    if (Bs3Layout<B, MODE_BINARY_PARTIAL>::kLdsUint4 > m) m = Bs3Layout<B, MODE_BINARY_PARTIAL>::kLdsUint4;
    if (GENK) {
        if (Bs3Layout<B, MODE_GENERAL_FULL>::kLdsUint4 > m) m = Bs3Layout<B, MODE_GENERAL_FULL>::kLdsUint4;
        if (Bs3Layout<B, MODE_PARTIAL>::kLdsUint4 > m) m = Bs3Layout<B, MODE_PARTIAL>::kLdsUint4;
    }
    return m;
}

struct FusedSegs {                  // segment order: binary_full, binary_partial, general_full, partial
    const uint4 *P[4];              // the class's panel slice (compact elements of its own width); unused when n_groups = 0
    uint32_t n_groups[4];           // 32-tree groups of the slice
    uint32_t m_bf;                  // trees of the binary_full segment (its padding trees resolve nothing)
};

template <int B, bool GENK, typename CT>
__global__ __launch_bounds__(kCountThreads) __attribute__((amdgpu_waves_per_eu(fused_waves<B, GENK>(), fused_waves<B, GENK>()))) void count_bitslice3_fused_kernel(
    FusedSegs segs, uint32_t npairs, uint32_t d_start, uint32_t d_hi, uint64_t rank_lo, uint32_t n_dblk, uint32_t total_tiles,
    const uint32_t *__restrict__ dprefix, const uint32_t *__restrict__ cprefix, CT *__restrict__ table, uint32_t *__restrict__ overflow_flag,
    uint32_t overwrite, uint32_t xcd_remap, const uint32_t *__restrict__ perm, uint32_t seg_sync) {
    __shared__ uint4 stage_all[kWavesPerBlock][fused_lds_uint4<B, GENK>()];
    Bs3Tile t;
    if (!bs3_decode_tile(t, d_start, d_hi, n_dblk, total_tiles, dprefix, cprefix, xcd_remap, perm)) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint4 *lds = stage_all[wave];
    uint32_t x0[kDB], x1[kDB], y0[kDB], y1[kDB], z0[kDB], z1[kDB];
#pragma unroll
    for (int j = 0; j < kDB; ++j) x0[j] = x1[j] = y0[j] = y1[j] = z0[j] = z1[j] = 0;
    // ONE dispatch on the tile's instance (both a-columns / all d slots live / diagonal) around all segments: the counters flow
    // through straight-line code from segment to segment. With the dispatch inside every segment (as the one-class kernel has it)
    // 48 counters meet in a four-way merge after each segment, and the register allocator answers with 30-50 more VGPRs than
    // the dearest segment needs alone -- scratch traffic inside the hot loops (profiles/r06_experiments.md).
    // (LDS operations of one wave execute in order: a segment's first stores follow the previous segment's last reads; the wave
    // barrier keeps the compiler from interleaving the two images)
    // waves of a workgroup in step (the step's s_barrier) per segment: a run-time flag on purpose (QS_SYNC_MODES in qs_bitslice3.hpp)
    auto remap_of = [&](int seg) { return ((seg_sync >> seg) & 1u) ? xcd_remap : (xcd_remap & ~2u); };
    auto all_segments = [&](auto var_tag) {
        constexpr int V = decltype(var_tag)::value;
        if ((QS_FUSED_SEGMASK & 1) && segs.n_groups[0]) bs3_segment<B, MODE_BINARY_FULL, V>(t, segs.P[0], npairs, segs.n_groups[0], remap_of(0), lds, x0, x1, y0, y1, z0, z1);
        __builtin_amdgcn_wave_barrier();
        if ((QS_FUSED_SEGMASK & 2) && segs.n_groups[1]) bs3_segment<B, MODE_BINARY_PARTIAL, V>(t, segs.P[1], npairs, segs.n_groups[1], remap_of(1), lds, x0, x1, y0, y1, z0, z1);
        // third topology of the binary segments: the trees that hold all four taxa minus the two counted
#pragma unroll
        for (int j = 0; j < kDB; ++j) {
            z0[j] = segs.m_bf + z0[j] - x0[j] - x1[j];
            z1[j] = segs.m_bf + z1[j] - y0[j] - y1[j];
        }
        if (GENK) {
            __builtin_amdgcn_wave_barrier();
            if ((QS_FUSED_SEGMASK & 4) && segs.n_groups[2]) bs3_segment<B, MODE_GENERAL_FULL, V>(t, segs.P[2], npairs, segs.n_groups[2], remap_of(2), lds, x0, x1, y0, y1, z0, z1);
            __builtin_amdgcn_wave_barrier();
            if ((QS_FUSED_SEGMASK & 8) && segs.n_groups[3]) bs3_segment<B, MODE_PARTIAL, V>(t, segs.P[3], npairs, segs.n_groups[3], remap_of(3), lds, x0, x1, y0, y1, z0, z1);
        }
    };
    switch (bs3_tile_variant(t)) {
        case 0: all_segments(std::integral_constant<int, 0>{}); break;
        case 1: all_segments(std::integral_constant<int, 1>{}); break;
        case 2: all_segments(std::integral_constant<int, 2>{}); break;
        default: all_segments(std::integral_constant<int, 3>{}); break;
    }
    bs3_store<CT, 2>(t, rank_lo, table, overflow_flag, overwrite, 0u, nullptr, x0, x1, y0, y1, z0, z1);
}

// seg_panel / seg_groups / seg_trees: per mode in the order of CountMode (MODE_BINARY_FULL = 0 ... see qs_common.hpp); a mode
// without trees in this launch has seg_groups = 0.
hipError_t launch_count_bitslice3_fused(hipStream_t s, const CountGeometry &g, const void *const seg_panel[4], const uint32_t seg_groups[4],
                                        const uint32_t seg_trees[4], int depth_bits, void *table, int count_bits, uint32_t *overflow_flag,
                                        bool overwrite) {
    if (g.total_tiles == 0) return hipSuccess;
    if (depth_bits < 4 || depth_bits > kFusedMaxBits) return hipErrorInvalidValue;
    static const int order[4] = {MODE_BINARY_FULL, MODE_BINARY_PARTIAL, MODE_GENERAL_FULL, MODE_PARTIAL};
    FusedSegs fs;
    for (int i = 0; i < 4; ++i) { fs.P[i] = (const uint4 *)seg_panel[order[i]]; fs.n_groups[i] = seg_groups[order[i]]; }
    fs.m_bf = seg_groups[MODE_BINARY_FULL] ? seg_trees[MODE_BINARY_FULL] : 0u;
    const bool genk = fs.n_groups[2] || fs.n_groups[3];
    uint32_t seg_sync = 0;       // bit i: segment i keeps the four waves of a workgroup in step (QS_SYNC_MODES, by CountMode)
    for (int i = 0; i < 4; ++i) seg_sync |= ((kSyncModes >> order[i]) & 1u) << i;
    const uint32_t npairs = (uint32_t)binom2(g.n);
    dim3 grid((g.total_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kCountThreads);
#define QS_FUSED(BB, GG, CT)                                                                                                          \
    hipLaunchKernelGGL((count_bitslice3_fused_kernel<BB, GG, CT>), grid, block, 0, s, fs, npairs, g.d_lo, g.d_hi, g.rank_lo, g.n_dblk,  \
                       g.total_tiles, g.dprefix, g.cprefix, (CT *)table, overflow_flag, overwrite ? 1u : 0u, g.n >= 200 ? 3u : 0u, g.perm, seg_sync)
#define QS_FUSED_B(GG, CT)                                                                                                            \
    do {                                                                                                                              \
        switch (depth_bits) {                                                                                                         \
            case 4: QS_FUSED(4, GG, CT); break;                                                                                       \
            case 5: QS_FUSED(5, GG, CT); break;                                                                                       \
            case 6: QS_FUSED(6, GG, CT); break;                                                                                       \
            default: QS_FUSED(7, GG, CT); break;                                                                                      \
        }                                                                                                                             \
    } while (0)
    if (count_bits == 32) { if (genk) QS_FUSED_B(true, uint32_t); else QS_FUSED_B(false, uint32_t); }
    else { if (genk) QS_FUSED_B(true, uint16_t); else QS_FUSED_B(false, uint16_t); }
#undef QS_FUSED_B
#undef QS_FUSED
    return hipGetLastError();
}

} // namespace qs
