// valu_banks.hip -- micro-benchmark: does v_bitop3_b32 / v_bcnt pay for VGPR bank placement or dependent chains?
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_banks.hip -o gpurun_out/valu_banks ; run on the GPU box.
// Every kernel runs 16 instructions per loop trip with explicit registers (v0..v40), 4 or 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define KERNEL(NAME, BODY)                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed, int iters) {       \
        uint32_t acc;                                                                            \
        asm volatile(                                                                            \
            "v_mov_b32 v0, %1\n v_mov_b32 v1, %1\n v_mov_b32 v2, %1\n v_mov_b32 v3, %1\n"        \
            "v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n"        \
            "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n"      \
            "v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"    \
            "v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v18, %1\n v_mov_b32 v19, %1\n"    \
            "v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n v_mov_b32 v22, %1\n v_mov_b32 v23, %1\n"    \
            "v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %1\n v_mov_b32 v27, %1\n"    \
            "v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %1\n v_mov_b32 v31, %1\n"    \
            "v_mov_b32 v32, %1\n v_mov_b32 v33, %1\n v_mov_b32 v34, %1\n v_mov_b32 v35, %1\n"    \
            "s_mov_b32 s20, %2\n"                                                                \
            "1:\n" BODY                                                                          \
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                  \
            "v_xor_b32 %0, v0, v1\n v_xor_b32 %0, %0, v2\n v_xor_b32 %0, %0, v3\n"               \
            : "=v"(acc) : "v"(seed + threadIdx.x), "s"(iters)                                    \
            : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
              "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","s20","scc"); \
        if (acc == 0x12345678u) out[0] = acc;                                                    \
    }

// 16 independent chains, sources in three DIFFERENT banks (reg index mod 4): dst/src0 = v(i), src1 = v(i+1), src2 = v(i+2)
#define B3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0xd4\n"
KERNEL(k_diffbank,
    B3(0,0,17,18) B3(1,1,18,19) B3(2,2,19,16) B3(3,3,16,17) B3(4,4,21,22) B3(5,5,22,23) B3(6,6,23,20) B3(7,7,20,21)
    B3(8,8,25,26) B3(9,9,26,27) B3(10,10,27,24) B3(11,11,24,25) B3(12,12,29,30) B3(13,13,30,31) B3(14,14,31,28) B3(15,15,28,29))
// all three sources in the SAME bank (indices equal mod 4)
KERNEL(k_samebank,
    B3(0,0,16,20) B3(1,1,17,21) B3(2,2,18,22) B3(3,3,19,23) B3(4,4,16,24) B3(5,5,17,25) B3(6,6,18,26) B3(7,7,19,27)
    B3(8,8,20,28) B3(9,9,21,29) B3(10,10,22,30) B3(11,11,23,31) B3(12,12,24,32) B3(13,13,25,33) B3(14,14,26,34) B3(15,15,27,35))
// two sources in the same bank
KERNEL(k_twobank,
    B3(0,0,16,21) B3(1,1,17,22) B3(2,2,18,23) B3(3,3,19,20) B3(4,4,20,25) B3(5,5,21,26) B3(6,6,22,27) B3(7,7,23,24)
    B3(8,8,24,29) B3(9,9,25,30) B3(10,10,26,31) B3(11,11,27,28) B3(12,12,28,33) B3(13,13,29,34) B3(14,14,30,35) B3(15,15,31,32))
// 4 chains of depth 4 (each instruction depends on the previous one of its chain: distance 4)
KERNEL(k_chain4,
    B3(0,0,17,18) B3(1,1,18,19) B3(2,2,19,16) B3(3,3,16,17) B3(0,0,21,22) B3(1,1,22,23) B3(2,2,23,20) B3(3,3,20,21)
    B3(0,0,25,26) B3(1,1,26,27) B3(2,2,27,24) B3(3,3,24,25) B3(0,0,29,30) B3(1,1,30,31) B3(2,2,31,28) B3(3,3,28,29))
// 2 chains (distance 2)
KERNEL(k_chain2,
    B3(0,0,17,18) B3(1,1,18,19) B3(0,0,19,16) B3(1,1,16,17) B3(0,0,21,22) B3(1,1,22,23) B3(0,0,23,20) B3(1,1,20,21)
    B3(0,0,25,26) B3(1,1,26,27) B3(0,0,27,24) B3(1,1,24,25) B3(0,0,29,30) B3(1,1,30,31) B3(0,0,31,28) B3(1,1,28,29))
// 1 chain (fully dependent)
KERNEL(k_chain1,
    B3(0,0,17,18) B3(0,0,18,19) B3(0,0,19,16) B3(0,0,16,17) B3(0,0,21,22) B3(0,0,22,23) B3(0,0,23,20) B3(0,0,20,21)
    B3(0,0,25,26) B3(0,0,26,27) B3(0,0,27,24) B3(0,0,24,25) B3(0,0,29,30) B3(0,0,30,31) B3(0,0,31,28) B3(0,0,28,29))
// the kernel's own pattern: gt/lt chains over 6 planes for 2 comparisons sharing R (4 chains, each 6 deep), then 4 bcnt
#define GT(d, l, r) "v_bitop3_b32 v" #d ", v" #l ", v" #r ", v" #d " bitop3:0xb2\n"
#define LT(d, l, r) "v_bitop3_b32 v" #d ", v" #l ", v" #r ", v" #d " bitop3:0x2b\n"
#define BC(d, s) "v_bcnt_u32_b32 v" #d ", v" #s ", v" #d "\n"
KERNEL(k_slot,
    GT(0,16,24) LT(1,16,24) GT(2,8,24) LT(3,8,24) GT(0,17,25) LT(1,17,25) GT(2,9,25) LT(3,9,25)
    GT(0,18,26) LT(1,18,26) GT(2,10,26) LT(3,10,26) GT(0,19,27) LT(1,19,27) GT(2,11,27) LT(3,11,27)
    GT(0,20,28) LT(1,20,28) GT(2,12,28) LT(3,12,28) GT(0,21,29) LT(1,21,29) GT(2,13,29) LT(3,13,29)
    BC(4,0) BC(5,1) BC(6,2) BC(7,3))
// v_bcnt only, different banks
KERNEL(k_bcnt,
    BC(0,16) BC(1,17) BC(2,18) BC(3,19) BC(4,20) BC(5,21) BC(6,22) BC(7,23) BC(8,24) BC(9,25) BC(10,26) BC(11,27) BC(12,28) BC(13,29) BC(14,30) BC(15,31))
// v_bcnt with dst/src1 and src0 in the same bank
KERNEL(k_bcnt_same,
    BC(0,16) BC(1,17) BC(2,18) BC(3,19) BC(4,16) BC(5,17) BC(6,18) BC(7,19) BC(8,20) BC(9,21) BC(10,22) BC(11,23) BC(12,24) BC(13,25) BC(14,26) BC(15,27))
// 2-operand forms: v_and_b32 (VOP2) in different banks
#define A2(d, a, b) "v_and_b32 v" #d ", v" #a ", v" #b "\n"
KERNEL(k_and2,
    A2(0,0,17) A2(1,1,18) A2(2,2,19) A2(3,3,16) A2(4,4,21) A2(5,5,22) A2(6,6,23) A2(7,7,20) A2(8,8,25) A2(9,9,26) A2(10,10,27) A2(11,11,24) A2(12,12,29) A2(13,13,30) A2(14,14,31) A2(15,15,28))

template <typename K> void run(const char *name, K kern, int per_trip, uint32_t *d, int waves_per_simd) {
    const int iters = 16384;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 12345u, 64);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 12345u, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    double winstr = (double)grid.x * 4 * iters * per_trip;
    double per_simd_per_s = winstr / (best * 1e-3) / 1024.0;
    printf("%-12s waves/SIMD=%d  %8.3f ms  %.3f G wave-instr/s/SIMD  (cycles per instr at 2.4 GHz: %.2f)\n", name, waves_per_simd, best,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    uint32_t *d; hipMalloc(&d, 64);
    for (int w : {1, 2, 4, 5, 8}) {
        run("diffbank", k_diffbank, 16, d, w); run("samebank", k_samebank, 16, d, w); run("twobank", k_twobank, 16, d, w);
        run("chain4", k_chain4, 16, d, w); run("chain2", k_chain2, 16, d, w); run("chain1", k_chain1, 16, d, w);
        run("slot(24+4)", k_slot, 28, d, w); run("bcnt", k_bcnt, 16, d, w); run("bcnt_same", k_bcnt_same, 16, d, w); run("and2", k_and2, 16, d, w);
        printf("\n");
    }
    return 0;
}
