// valu_yardstick.hip -- the yardstick for the count kernel's VALU-issue roofline: cycles per wave-instruction per SIMD
// of explicit-register instruction streams on gfx950, measured IN CYCLES (s_memtime around the loop of every wave),
// not wall time over a nominal clock, after a pre-warm, with >= 200 ms per variant.
//   tools/bin/valu_yardstick [target_ms=250]
// Per variant and occupancy (waves per SIMD) it prints
//   cyc/instr = per XCD: (last end - first start of its waves, s_memtime) x 128 SIMDs / instructions issued there
//   resident  = sum of the waves' lifetimes / (span x 128): how many waves a SIMD really held on average
//   clock     = d s_memtime / d s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
// Sanity line of the guide: one wave per SIMD of independent v_and_b32 = 4 cycles, >= 2 waves -> 2.
// Build: make -C tools bin/valu_yardstick
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every kernel: 40 VGPRs initialised from (seed + lane), then `iters` trips of BODY (kPerTrip instructions), stamps
// around the loop, lane 0 of every wave writes (cycles, realtime ticks)
#define KERNEL(NAME, BODY) KERNELX(NAME, BODY BODY BODY BODY)
#define KERNELX(NAME, BODY)                                                                                           \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *stamps, uint32_t *out, uint32_t seed, int iters) { \
        uint32_t acc;                                                                                                \
        unsigned long long t0, t1, r0, r1;                                                                           \
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");     \
        asm volatile(                                                                                                \
            "v_mov_b32 v0, %1\n v_add_u32 v1, 1, %1\n v_add_u32 v2, 2, %1\n v_add_u32 v3, 3, %1\n"                   \
            "v_add_u32 v4, 4, %1\n v_add_u32 v5, 5, %1\n v_add_u32 v6, 6, %1\n v_add_u32 v7, 7, %1\n"                \
            "v_add_u32 v8, 8, %1\n v_add_u32 v9, 9, %1\n v_add_u32 v10, 10, %1\n v_add_u32 v11, 11, %1\n"            \
            "v_add_u32 v12, 12, %1\n v_add_u32 v13, 13, %1\n v_add_u32 v14, 14, %1\n v_add_u32 v15, 15, %1\n"        \
            "v_mul_u32_u24 v16, 3, %1\n v_mul_u32_u24 v17, 5, %1\n v_mul_u32_u24 v18, 7, %1\n v_mul_u32_u24 v19, 9, %1\n" \
            "v_mul_u32_u24 v20, 11, %1\n v_mul_u32_u24 v21, 13, %1\n v_mul_u32_u24 v22, 17, %1\n v_mul_u32_u24 v23, 19, %1\n" \
            "v_mul_u32_u24 v24, 23, %1\n v_mul_u32_u24 v25, 29, %1\n v_mul_u32_u24 v26, 31, %1\n v_mul_u32_u24 v27, 37, %1\n" \
            "v_mul_u32_u24 v28, 41, %1\n v_mul_u32_u24 v29, 43, %1\n v_mul_u32_u24 v30, 47, %1\n v_mul_u32_u24 v31, 53, %1\n" \
            "v_mul_u32_u24 v32, 59, %1\n v_mul_u32_u24 v33, 61, %1\n v_mul_u32_u24 v34, 62, %1\n v_mul_u32_u24 v35, 63, %1\n" \
            "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"                            \
            "s_mov_b32 s20, %2\n s_mov_b32 s21, 0x33cc55aa\n s_mov_b32 s22, 0x5a5a1234\n s_mov_b32 s23, 0x0f0f3c3c\n"        \
            "1:\n" BODY                                                                                          \
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                      \
            "v_xor_b32 %0, v0, v1\n v_xor_b32 %0, %0, v2\n v_xor_b32 %0, %0, v3\n v_xor_b32 %0, %0, v36\n v_xor_b32 %0, %0, v37\n" \
            : "=v"(acc) : "v"(seed + threadIdx.x * 2654435761u), "s"(iters)                                          \
            : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
              "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
              "s20","s21","s22","s23","scc","memory");                                                               \
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");     \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            const uint32_t w = blockIdx.x * 4 + threadIdx.x / 64;                                                    \
            uint32_t xcc;                                                                                            \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                       \
            stamps[4 * w] = t0; stamps[4 * w + 1] = t1; stamps[4 * w + 2] = r1 - r0; stamps[4 * w + 3] = xcc & 15;   \
        }                                                                                                            \
        if (acc == 0x12345678u) out[0] = acc;                                                                        \
    }

constexpr int kPerTrip = 128;   // the body (32 instructions) four times per loop trip: the taken branch costs a lone wave ~44 cycles

#define A2(d, a, b) "v_and_b32 v" #d ", v" #a ", v" #b "\n"
#define AD(d, a, b) "v_add_u32 v" #d ", v" #a ", v" #b "\n"
#define B3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0xd4\n"
#define B3S(d, a, s, c) "v_bitop3_b32 v" #d ", v" #a ", s" #s ", v" #c " bitop3:0xd4\n"
#define GT(d, l, r) "v_bitop3_b32 v" #d ", v" #l ", v" #r ", v" #d " bitop3:0xb2\n"
#define LT(d, l, r) "v_bitop3_b32 v" #d ", v" #l ", v" #r ", v" #d " bitop3:0x8e\n"
#define BC(d, s) "v_bcnt_u32_b32 v" #d ", v" #s ", v" #d "\n"
#define BC0(d, s) "v_bcnt_u32_b32 v" #d ", v" #s ", 0\n"
#define LA(d, a) "v_lshl_add_u32 v" #d ", v" #a ", 16, v" #d "\n"
#define MAD(d, a, b) "v_mad_u32_u24 v" #d ", v" #a ", v" #b ", v" #d "\n"

// 1. independent v_and_b32 (VOP2, two VGPR sources in different banks), 16 destinations
KERNEL(k_and,
    A2(0,0,17) A2(1,1,18) A2(2,2,19) A2(3,3,16) A2(4,4,21) A2(5,5,22) A2(6,6,23) A2(7,7,20) A2(8,8,25) A2(9,9,26) A2(10,10,27) A2(11,11,24) A2(12,12,29) A2(13,13,30) A2(14,14,31) A2(15,15,28)
    A2(0,0,18) A2(1,1,19) A2(2,2,16) A2(3,3,17) A2(4,4,22) A2(5,5,23) A2(6,6,20) A2(7,7,21) A2(8,8,26) A2(9,9,27) A2(10,10,24) A2(11,11,25) A2(12,12,30) A2(13,13,31) A2(14,14,28) A2(15,15,29))
// 2. independent v_add_u32
KERNEL(k_add,
    AD(0,0,17) AD(1,1,18) AD(2,2,19) AD(3,3,16) AD(4,4,21) AD(5,5,22) AD(6,6,23) AD(7,7,20) AD(8,8,25) AD(9,9,26) AD(10,10,27) AD(11,11,24) AD(12,12,29) AD(13,13,30) AD(14,14,31) AD(15,15,28)
    AD(0,0,18) AD(1,1,19) AD(2,2,16) AD(3,3,17) AD(4,4,22) AD(5,5,23) AD(6,6,20) AD(7,7,21) AD(8,8,26) AD(9,9,27) AD(10,10,24) AD(11,11,25) AD(12,12,30) AD(13,13,31) AD(14,14,28) AD(15,15,29))
// 3. v_bitop3_b32, three VGPR sources in three different banks (index mod 4), 16 independent destinations
KERNEL(k_b3_diff,
    B3(0,0,17,18) B3(1,1,18,19) B3(2,2,19,16) B3(3,3,16,17) B3(4,4,21,22) B3(5,5,22,23) B3(6,6,23,20) B3(7,7,20,21)
    B3(8,8,25,26) B3(9,9,26,27) B3(10,10,27,24) B3(11,11,24,25) B3(12,12,29,30) B3(13,13,30,31) B3(14,14,31,28) B3(15,15,28,29)
    B3(0,0,21,22) B3(1,1,22,23) B3(2,2,23,20) B3(3,3,20,21) B3(4,4,25,26) B3(5,5,26,27) B3(6,6,27,24) B3(7,7,24,25)
    B3(8,8,29,30) B3(9,9,30,31) B3(10,10,31,28) B3(11,11,28,29) B3(12,12,17,18) B3(13,13,18,19) B3(14,14,19,16) B3(15,15,16,17))
// 4. v_bitop3_b32, all three sources in ONE bank
KERNEL(k_b3_same,
    B3(0,0,16,20) B3(1,1,17,21) B3(2,2,18,22) B3(3,3,19,23) B3(4,4,16,24) B3(5,5,17,25) B3(6,6,18,26) B3(7,7,19,27)
    B3(8,8,20,28) B3(9,9,21,29) B3(10,10,22,30) B3(11,11,23,31) B3(12,12,24,32) B3(13,13,25,33) B3(14,14,26,34) B3(15,15,27,35)
    B3(0,0,24,28) B3(1,1,25,29) B3(2,2,26,30) B3(3,3,27,31) B3(4,4,20,32) B3(5,5,21,33) B3(6,6,22,34) B3(7,7,23,35)
    B3(8,8,16,24) B3(9,9,17,25) B3(10,10,18,26) B3(11,11,19,27) B3(12,12,16,20) B3(13,13,17,21) B3(14,14,18,22) B3(15,15,19,23))
// 5. v_bitop3_b32, ONE fully dependent chain
KERNEL(k_b3_chain1,
    B3(0,0,17,18) B3(0,0,18,19) B3(0,0,19,16) B3(0,0,16,17) B3(0,0,21,22) B3(0,0,22,23) B3(0,0,23,20) B3(0,0,20,21)
    B3(0,0,25,26) B3(0,0,26,27) B3(0,0,27,24) B3(0,0,24,25) B3(0,0,29,30) B3(0,0,30,31) B3(0,0,31,28) B3(0,0,28,29)
    B3(0,0,17,18) B3(0,0,18,19) B3(0,0,19,16) B3(0,0,16,17) B3(0,0,21,22) B3(0,0,22,23) B3(0,0,23,20) B3(0,0,20,21)
    B3(0,0,25,26) B3(0,0,26,27) B3(0,0,27,24) B3(0,0,24,25) B3(0,0,29,30) B3(0,0,30,31) B3(0,0,31,28) B3(0,0,28,29))
// 6. v_bitop3_b32 with one SGPR source
KERNEL(k_b3_sgpr,
    B3S(0,0,21,18) B3S(1,1,22,19) B3S(2,2,23,16) B3S(3,3,21,17) B3S(4,4,22,22) B3S(5,5,23,23) B3S(6,6,21,20) B3S(7,7,22,21)
    B3S(8,8,23,26) B3S(9,9,21,27) B3S(10,10,22,24) B3S(11,11,23,25) B3S(12,12,21,30) B3S(13,13,22,31) B3S(14,14,23,28) B3S(15,15,21,29)
    B3S(0,0,22,18) B3S(1,1,23,19) B3S(2,2,21,16) B3S(3,3,22,17) B3S(4,4,23,22) B3S(5,5,21,23) B3S(6,6,22,20) B3S(7,7,23,21)
    B3S(8,8,21,26) B3S(9,9,22,27) B3S(10,10,23,24) B3S(11,11,21,25) B3S(12,12,22,30) B3S(13,13,23,31) B3S(14,14,21,28) B3S(15,15,22,29))
// 7. v_bcnt_u32_b32 with accumulate (the count kernel's form), 16 accumulators
KERNEL(k_bcnt,
    BC(0,16) BC(1,17) BC(2,18) BC(3,19) BC(4,20) BC(5,21) BC(6,22) BC(7,23) BC(8,24) BC(9,25) BC(10,26) BC(11,27) BC(12,28) BC(13,29) BC(14,30) BC(15,31)
    BC(0,17) BC(1,18) BC(2,19) BC(3,16) BC(4,21) BC(5,22) BC(6,23) BC(7,20) BC(8,25) BC(9,26) BC(10,27) BC(11,24) BC(12,29) BC(13,30) BC(14,31) BC(15,28))
// 8. v_bcnt_u32_b32 with the constant 0 as addend
KERNEL(k_bcnt0,
    BC0(0,16) BC0(1,17) BC0(2,18) BC0(3,19) BC0(4,20) BC0(5,21) BC0(6,22) BC0(7,23) BC0(8,24) BC0(9,25) BC0(10,26) BC0(11,27) BC0(12,28) BC0(13,29) BC0(14,30) BC0(15,31)
    BC0(0,17) BC0(1,18) BC0(2,19) BC0(3,16) BC0(4,21) BC0(5,22) BC0(6,23) BC0(7,20) BC0(8,25) BC0(9,26) BC0(10,27) BC0(11,24) BC0(12,29) BC0(13,30) BC0(14,31) BC0(15,28))
// 9. the count kernel's slot at B = 5: two comparisons sharing R (4 chains of 6 v_bitop3) + 4 v_bcnt = 28, plus 4 more
//    chain steps to fill the trip (32): L1 = v16..21, L2 = v8..13, R = v24..29
KERNEL(k_slot,
    GT(0,16,24) LT(1,16,24) GT(2,8,24) LT(3,8,24) GT(0,17,25) LT(1,17,25) GT(2,9,25) LT(3,9,25)
    GT(0,18,26) LT(1,18,26) GT(2,10,26) LT(3,10,26) GT(0,19,27) LT(1,19,27) GT(2,11,27) LT(3,11,27)
    GT(0,20,28) LT(1,20,28) GT(2,12,28) LT(3,12,28) GT(0,21,29) LT(1,21,29) GT(2,13,29) LT(3,13,29)
    BC(36,0) BC(37,1) BC(38,2) BC(39,3)
    GT(4,16,30) LT(5,16,30) GT(6,8,30) LT(7,8,30))
// 10. the same slot with two 16-bit counters per register: bcnt of gt accumulates, bcnt of lt to a temp + v_lshl_add
KERNEL(k_slot_pk,
    GT(0,16,24) LT(1,16,24) GT(2,8,24) LT(3,8,24) GT(0,17,25) LT(1,17,25) GT(2,9,25) LT(3,9,25)
    GT(0,18,26) LT(1,18,26) GT(2,10,26) LT(3,10,26) GT(0,19,27) LT(1,19,27) GT(2,11,27) LT(3,11,27)
    GT(0,20,28) LT(1,20,28) GT(2,12,28) LT(3,12,28) GT(0,21,29) LT(1,21,29) GT(2,13,29) LT(3,13,29)
    BC(36,0) BC0(14,1) BC(38,2) BC0(15,3) LA(36,14) LA(38,15)
    GT(4,16,30) LT(5,16,30))
// 10b. the same 96 v_bitop3 + 16 v_bcnt + 16 v_bitop3 per trip, but the popcounts in groups: after every TWO slots (8 in a row)
//      and after every FOUR slots (16 in a row): does interleaving 4 half-rate popcounts per 24 full-rate ops cost extra?
#define CH6(g, l, r) GT(g,l,r) LT(g##1,l,r)
#define SLOTCH(a0,a1,a2,a3, R0,R1,R2,R3,R4,R5) \
    GT(a0,16,R0) LT(a1,16,R0) GT(a2,8,R0) LT(a3,8,R0) GT(a0,17,R1) LT(a1,17,R1) GT(a2,9,R1) LT(a3,9,R1) \
    GT(a0,18,R2) LT(a1,18,R2) GT(a2,10,R2) LT(a3,10,R2) GT(a0,19,R3) LT(a1,19,R3) GT(a2,11,R3) LT(a3,11,R3) \
    GT(a0,20,R4) LT(a1,20,R4) GT(a2,12,R4) LT(a3,12,R4) GT(a0,21,R5) LT(a1,21,R5) GT(a2,13,R5) LT(a3,13,R5)
#define FILL4 GT(14,16,30) LT(15,16,30) GT(14,8,30) LT(15,8,30)
KERNELX(k_slot_g1,
    SLOTCH(0,1,2,3, 24,25,26,27,28,29) BC(36,0) BC(37,1) BC(38,2) BC(39,3) FILL4
    SLOTCH(4,5,6,7, 24,25,26,27,28,29) BC(36,4) BC(37,5) BC(38,6) BC(39,7) FILL4
    SLOTCH(0,1,2,3, 25,26,27,28,29,24) BC(36,0) BC(37,1) BC(38,2) BC(39,3) FILL4
    SLOTCH(4,5,6,7, 25,26,27,28,29,24) BC(36,4) BC(37,5) BC(38,6) BC(39,7) FILL4)
KERNELX(k_slot_g2,
    SLOTCH(0,1,2,3, 24,25,26,27,28,29) SLOTCH(4,5,6,7, 24,25,26,27,28,29)
    BC(36,0) BC(37,1) BC(38,2) BC(39,3) BC(36,4) BC(37,5) BC(38,6) BC(39,7) FILL4 FILL4
    SLOTCH(0,1,2,3, 25,26,27,28,29,24) SLOTCH(4,5,6,7, 25,26,27,28,29,24)
    BC(36,0) BC(37,1) BC(38,2) BC(39,3) BC(36,4) BC(37,5) BC(38,6) BC(39,7) FILL4 FILL4)
KERNELX(k_slot_g4,
    SLOTCH(0,1,2,3, 24,25,26,27,28,29) SLOTCH(4,5,6,7, 24,25,26,27,28,29) SLOTCH(32,33,34,35, 25,26,27,28,29,24) SLOTCH(22,23,31,30, 25,26,27,28,29,24)
    BC(36,0) BC(37,1) BC(38,2) BC(39,3) BC(36,4) BC(37,5) BC(38,6) BC(39,7) BC(36,32) BC(37,33) BC(38,34) BC(39,35) BC(36,22) BC(37,23) BC(38,31) BC(39,30)
    FILL4 FILL4 FILL4 FILL4)
// 12. VGPR source banks (register number mod 4). v_add3_u32 d = d + a + b with the three sources in three banks / a and b in one / all in one
#define ADD3(d, a, b) "v_add3_u32 v" #d ", v" #d ", v" #a ", v" #b "\n"
KERNEL(k_add3_nc,
    ADD3(0,17,22) ADD3(1,18,23) ADD3(2,19,20) ADD3(3,16,21) ADD3(4,25,30) ADD3(5,26,31) ADD3(6,27,28) ADD3(7,24,29) ADD3(8,17,22) ADD3(9,18,23) ADD3(10,19,20) ADD3(11,16,21) ADD3(12,25,30) ADD3(13,26,31) ADD3(14,27,28) ADD3(15,24,29)
    ADD3(0,17,22) ADD3(1,18,23) ADD3(2,19,20) ADD3(3,16,21) ADD3(4,25,30) ADD3(5,26,31) ADD3(6,27,28) ADD3(7,24,29) ADD3(8,17,22) ADD3(9,18,23) ADD3(10,19,20) ADD3(11,16,21) ADD3(12,25,30) ADD3(13,26,31) ADD3(14,27,28) ADD3(15,24,29))
KERNEL(k_add3_c2,
    ADD3(0,17,21) ADD3(1,18,22) ADD3(2,19,23) ADD3(3,16,20) ADD3(4,25,29) ADD3(5,26,30) ADD3(6,27,31) ADD3(7,24,28) ADD3(8,17,21) ADD3(9,18,22) ADD3(10,19,23) ADD3(11,16,20) ADD3(12,25,29) ADD3(13,26,30) ADD3(14,27,31) ADD3(15,24,28)
    ADD3(0,17,21) ADD3(1,18,22) ADD3(2,19,23) ADD3(3,16,20) ADD3(4,25,29) ADD3(5,26,30) ADD3(6,27,31) ADD3(7,24,28) ADD3(8,17,21) ADD3(9,18,22) ADD3(10,19,23) ADD3(11,16,20) ADD3(12,25,29) ADD3(13,26,30) ADD3(14,27,31) ADD3(15,24,28))
KERNEL(k_add3_c3,
    ADD3(0,16,20) ADD3(1,17,21) ADD3(2,18,22) ADD3(3,19,23) ADD3(4,24,28) ADD3(5,25,29) ADD3(6,26,30) ADD3(7,27,31) ADD3(8,16,20) ADD3(9,17,21) ADD3(10,18,22) ADD3(11,19,23) ADD3(12,24,28) ADD3(13,25,29) ADD3(14,26,30) ADD3(15,27,31)
    ADD3(0,16,20) ADD3(1,17,21) ADD3(2,18,22) ADD3(3,19,23) ADD3(4,24,28) ADD3(5,25,29) ADD3(6,26,30) ADD3(7,27,31) ADD3(8,16,20) ADD3(9,17,21) ADD3(10,18,22) ADD3(11,19,23) ADD3(12,24,28) ADD3(13,25,29) ADD3(14,26,30) ADD3(15,27,31))
// v_and_b32 with both sources (and the destination) in one bank
KERNEL(k_and_c,
    A2(0,0,16) A2(1,1,17) A2(2,2,18) A2(3,3,19) A2(4,4,20) A2(5,5,21) A2(6,6,22) A2(7,7,23) A2(8,8,24) A2(9,9,25) A2(10,10,26) A2(11,11,27) A2(12,12,28) A2(13,13,29) A2(14,14,30) A2(15,15,31)
    A2(0,0,16) A2(1,1,17) A2(2,2,18) A2(3,3,19) A2(4,4,20) A2(5,5,21) A2(6,6,22) A2(7,7,23) A2(8,8,24) A2(9,9,25) A2(10,10,26) A2(11,11,27) A2(12,12,28) A2(13,13,29) A2(14,14,30) A2(15,15,31))
// the slot's mix with NO two sources of an instruction in one bank / L and R planes in one bank / accumulator in R's bank
KERNEL(k_slot_nc,
    GT(2,16,25) LT(3,16,25) GT(6,8,25) LT(7,8,25)
    GT(3,17,26) LT(0,17,26) GT(7,9,26) LT(4,9,26)
    GT(0,18,27) LT(1,18,27) GT(4,10,27) LT(5,10,27)
    GT(1,19,28) LT(2,19,28) GT(5,11,28) LT(6,11,28)
    GT(2,20,29) LT(3,20,29) GT(6,12,29) LT(7,12,29)
    GT(3,21,30) LT(0,21,30) GT(7,13,30) LT(4,13,30)
    BC(36,1) BC(37,2) BC(38,3) BC(39,0)
    GT(2,16,25) LT(3,16,25) GT(6,8,25) LT(7,8,25))
KERNEL(k_slot_clr,
    GT(1,16,24) LT(2,16,24) GT(5,8,24) LT(6,8,24)
    GT(2,17,25) LT(3,17,25) GT(6,9,25) LT(7,9,25)
    GT(3,18,26) LT(0,18,26) GT(7,10,26) LT(4,10,26)
    GT(0,19,27) LT(1,19,27) GT(4,11,27) LT(5,11,27)
    GT(1,20,28) LT(2,20,28) GT(5,12,28) LT(6,12,28)
    GT(2,21,29) LT(3,21,29) GT(6,13,29) LT(7,13,29)
    BC(36,1) BC(37,2) BC(38,3) BC(39,0)
    GT(1,16,24) LT(2,16,24) GT(5,8,24) LT(6,8,24))
KERNEL(k_slot_cacc,
    GT(1,16,25) LT(5,16,25) GT(33,8,25) LT(1,8,25)
    GT(2,17,26) LT(6,17,26) GT(34,9,26) LT(2,9,26)
    GT(3,18,27) LT(7,18,27) GT(35,10,27) LT(3,10,27)
    GT(0,19,28) LT(4,19,28) GT(32,11,28) LT(0,11,28)
    GT(1,20,29) LT(5,20,29) GT(33,12,29) LT(1,12,29)
    GT(2,21,30) LT(6,21,30) GT(34,13,30) LT(2,13,30)
    BC(36,1) BC(37,2) BC(38,3) BC(39,0)
    GT(1,16,25) LT(5,16,25) GT(33,8,25) LT(1,8,25))
// 11. 24 v_bitop3 + 8 v_add_u32 (is the slot's excess over 2 cycles the popcount or the mix?)
KERNEL(k_slot_add,
    GT(0,16,24) LT(1,16,24) GT(2,8,24) LT(3,8,24) GT(0,17,25) LT(1,17,25) GT(2,9,25) LT(3,9,25)
    GT(0,18,26) LT(1,18,26) GT(2,10,26) LT(3,10,26) GT(0,19,27) LT(1,19,27) GT(2,11,27) LT(3,11,27)
    GT(0,20,28) LT(1,20,28) GT(2,12,28) LT(3,12,28) GT(0,21,29) LT(1,21,29) GT(2,13,29) LT(3,13,29)
    AD(36,36,0) AD(37,37,1) AD(38,38,2) AD(39,39,3) AD(4,4,30) AD(5,5,31) AD(6,6,32) AD(7,7,33))
// 12. v_lshl_add_u32 and v_mad_u32_u24 (VOP3 integer forms)
KERNEL(k_lshladd,
    LA(0,16) LA(1,17) LA(2,18) LA(3,19) LA(4,20) LA(5,21) LA(6,22) LA(7,23) LA(8,24) LA(9,25) LA(10,26) LA(11,27) LA(12,28) LA(13,29) LA(14,30) LA(15,31)
    LA(0,17) LA(1,18) LA(2,19) LA(3,16) LA(4,21) LA(5,22) LA(6,23) LA(7,20) LA(8,25) LA(9,26) LA(10,27) LA(11,24) LA(12,29) LA(13,30) LA(14,31) LA(15,28))
KERNEL(k_mad24,
    MAD(0,16,32) MAD(1,17,33) MAD(2,18,34) MAD(3,19,35) MAD(4,20,32) MAD(5,21,33) MAD(6,22,34) MAD(7,23,35) MAD(8,24,32) MAD(9,25,33) MAD(10,26,34) MAD(11,27,35) MAD(12,28,32) MAD(13,29,33) MAD(14,30,34) MAD(15,31,35)
    MAD(0,17,32) MAD(1,18,33) MAD(2,19,34) MAD(3,16,35) MAD(4,21,32) MAD(5,22,33) MAD(6,23,34) MAD(7,20,35) MAD(8,25,32) MAD(9,26,33) MAD(10,27,34) MAD(11,24,35) MAD(12,29,32) MAD(13,30,33) MAD(14,31,34) MAD(15,28,35))

typedef void (*kern_t)(unsigned long long *, uint32_t *, uint32_t, int);

struct Result { double cyc_per_instr, clock_ghz, wall_cyc_per_instr, ms, xcd_cyc_per_instr, resident; };

static Result run(kern_t kern, unsigned long long *d_st, uint32_t *d_out, int wps, double target_ms) {
    const int blocks = 256 * wps; // 4 waves per block, one per SIMD: wps blocks per CU
    const int waves = blocks * 4;
    // Placement: the dispatcher may put several blocks on one CU and none on another. Every block asks for 1/wps of a CU's
    // 160 KiB of LDS (never touched), so that exactly wps blocks fit on a CU and a full grid puts wps waves on every SIMD.
    const int lds = ((160 * 1024) / wps) & ~1023;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // calibrate: a short run gives the trip time, then size the measured run for target_ms
    int iters = 20000;
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_st, d_out, 12345u, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    iters = (int)std::min(2.0e9, std::max(20000.0, iters * target_ms / std::max(ms, 0.01f)));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_st, d_out, 12345u, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> st(4 * (size_t)waves);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc(waves), clk(waves);
    // per XCD (s_memtime is compared only inside one XCD): span = first start .. last end of its waves
    unsigned long long lo[16], hi[16], dur[16], cnt[16];
    for (int x = 0; x < 16; ++x) { lo[x] = ~0ull; hi[x] = 0; dur[x] = 0; cnt[x] = 0; }
    for (int w = 0; w < waves; ++w) {
        const unsigned long long t0 = st[4 * w], t1 = st[4 * w + 1];
        const int x = (int)(st[4 * w + 3] & 15);
        cyc[w] = (double)(t1 - t0) / ((double)iters * kPerTrip);
        clk[w] = (double)(t1 - t0) / (double)st[4 * w + 2] * 0.1; // GHz: realtime ticks at 100 MHz
        lo[x] = std::min(lo[x], t0); hi[x] = std::max(hi[x], t1); dur[x] += t1 - t0; cnt[x]++;
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    std::vector<double> xr, xo;
    for (int x = 0; x < 16; ++x)
        if (cnt[x]) {
            const double span = (double)(hi[x] - lo[x]);
            xr.push_back(span * 128.0 / ((double)cnt[x] * iters * kPerTrip));   // 32 CUs x 4 SIMDs per XCD
            xo.push_back((double)dur[x] / (span * 128.0));
        }
    std::sort(xr.begin(), xr.end()); std::sort(xo.begin(), xo.end());
    Result r;
    r.cyc_per_instr = cyc[waves / 2] / wps;
    r.clock_ghz = clk[waves / 2];
    r.ms = ms;
    r.wall_cyc_per_instr = (ms * 1e-3) * r.clock_ghz * 1e9 / ((double)iters * kPerTrip) / wps;
    r.xcd_cyc_per_instr = xr.empty() ? 0 : xr[xr.size() / 2];
    r.resident = xo.empty() ? 0 : xo[xo.size() / 2];
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return r;
}

int main(int argc, char **argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 250.0;
    unsigned long long *d_st; uint32_t *d_out;
    CK(hipMalloc(&d_st, 4 * 8 * 256 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_out, 64));
    struct V { const char *name; kern_t k; } vs[] = {
        {"v_and_b32 indep", k_and}, {"v_add_u32 indep", k_add}, {"v_bitop3 3 banks", k_b3_diff}, {"v_bitop3 1 bank", k_b3_same},
        {"v_bitop3 1 chain", k_b3_chain1}, {"v_bitop3 sgpr src", k_b3_sgpr}, {"v_bcnt acc", k_bcnt}, {"v_bcnt +0", k_bcnt0},
        {"slot 28/32 (B=5)", k_slot}, {"slots, bcnt x4", k_slot_g1}, {"slots, bcnt x8", k_slot_g2}, {"slots, bcnt x16", k_slot_g4}, {"slot, no bank conflict", k_slot_nc}, {"slot, L/R one bank", k_slot_clr}, {"slot, acc in R bank", k_slot_cacc}, {"v_add3 3 banks", k_add3_nc}, {"v_add3 2 in a bank", k_add3_c2}, {"v_add3 1 bank", k_add3_c3}, {"v_and_b32 1 bank", k_and_c}, {"slot packed ctrs", k_slot_pk}, {"slot, add for bcnt", k_slot_add},
        {"v_lshl_add_u32", k_lshladd}, {"v_mad_u32_u24", k_mad24}};
    // pre-warm: 1.5 s of VALU work so that the clock has settled before the first measured variant
    for (int i = 0; i < 3; ++i) run(k_slot, d_st, d_out, 4, 500.0);
    printf("# cycles per wave-instruction per SIMD-32 (s_memtime): XCD figure = (last end - first start of an XCD's waves) x 128 SIMDs / its\n"
           "# instructions, median over the XCDs; {average resident waves per SIMD over that span}; [clock GHz = d s_memtime / d s_memrealtime];\n"
           "# (median wave: its own cycles per instruction / waves per SIMD -- below the XCD figure when fewer waves than planned were resident)\n");
    printf("%-20s", "waves/SIMD planned");
    const int occ[] = {1, 2, 3, 4, 5, 6, 8};
    for (int w : occ) printf("  %26d", w);
    printf("\n");
    for (const V &v : vs) {
        printf("%-20s", v.name);
        for (int w : occ) {
            const Result r = run(v.k, d_st, d_out, w, target_ms);
            printf("  %5.2f {%3.1f} [%4.2f] (%4.2f)", r.xcd_cyc_per_instr, r.resident, r.clock_ghz, r.cyc_per_instr);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
