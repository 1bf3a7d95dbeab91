// valu_rates.hip -- micro-benchmark: issue rate of the integer VALU ops the count kernel is built from.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t s = seed | 0x80808080u, t = seed * 3u, u = seed * 5u + threadIdx.x, w = seed * 7u;
    for (int it = 0; it < iters; ++it) {
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(s));
#define BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc" : "+v"(r[i]) : "v"(s), "v"(t));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "v"(t));
#define SAD(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(s), "v"(t));
#define DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(s), "v"(t));
#define PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define PKSUB(i) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define PKMAX(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(s));
#define ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "v"(t));
#define SUBSGPR(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r[i]) : "s"(seed));
#define ADDLIT(i) asm volatile("v_add_u32 %0, 0xfefefeff, %0" : "+v"(r[i]));
#define ANDLIT(i) asm volatile("v_and_b32 %0, 0x80808080, %0" : "+v"(r[i]));
#define ANDSGPR(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "s"(seed));
#define BITOP3S(i) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc" : "+v"(r[i]) : "s"(seed));
#define BITOP3V(i) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc" : "+v"(r[i]) : "v"(s));
#define ADDINL(i) asm volatile("v_add_u32 %0, 17, %0" : "+v"(r[i]));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define LSHR(i) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(r[i]));
#define BCNT0(i) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(r[i]));
#define STEP7(i) asm volatile("v_sub_u32 %0, %2, %1\n v_add_u32 %0, %0, %3\n v_add_u32 %1, %4, %0\n v_bitop3_b32 %0, %0, %5, %0 bitop3:0xc\n v_bcnt_u32_b32 %6, %0, %6\n v_and_b32 %1, %5, %1\n v_bcnt_u32_b32 %7, %1, %7" : "+v"(r[i]), "+v"(r[(i+1)&15]) : "v"(s), "v"(t), "v"(u), "v"(w), "v"(r[(i+2)&15]), "v"(r[(i+3)&15]));
#define STEP7S(i) asm volatile("v_sub_u32 %0, %2, %1\n v_add_u32 %0, %0, %3\n v_add_u32 %1, 0xfefefeff, %0\n v_bitop3_b32 %0, %0, %4, %0 bitop3:0xc\n v_bcnt_u32_b32 %5, %0, %5\n v_and_b32 %1, 0x80808080, %1\n v_bcnt_u32_b32 %6, %1, %6" : "+v"(r[i]), "+v"(r[(i+1)&15]) : "s"(seed), "v"(t), "s"(iters), "v"(r[(i+2)&15]), "v"(r[(i+3)&15]));
        if (OP == 0) { REP16(ADD) }
        if (OP == 1) { REP16(SUB) }
        if (OP == 2) { REP16(AND) }
        if (OP == 3) { REP16(BCNT) }
        if (OP == 4) { REP16(BITOP3) }
        if (OP == 5) { REP16(ADD3) }
        if (OP == 6) { REP16(SAD) }
        if (OP == 7) { REP16(DOT4) }
        if (OP == 8) { REP16(PKADD) }
        if (OP == 9) { REP16(PKSUB) }
        if (OP == 10) { REP16(PKMAX) }
        if (OP == 11) { REP16(LSHLADD) }
        if (OP == 12) { REP16(ANDOR) }
        if (OP == 13) { REP16(SUBSGPR) }
        if (OP == 14) { REP16(ADDLIT) }
        if (OP == 15) { REP16(ANDLIT) }
        if (OP == 16) { REP16(ANDSGPR) }
        if (OP == 17) { REP16(BITOP3S) }
        if (OP == 18) { REP16(BITOP3V) }
        if (OP == 19) { REP16(ADDINL) }
        if (OP == 20) { REP16(XOR) }
        if (OP == 21) { REP16(LSHR) }
        if (OP == 22) { REP16(BCNT0) }
        if (OP == 23) { STEP7(0) STEP7(4) STEP7(8) STEP7(12) }
        if (OP == 24) { STEP7S(0) STEP7S(4) STEP7S(8) STEP7S(12) }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int OP> double run(const char *name, uint32_t *d, int waves_per_simd) {
    const int iters = 16384;
    dim3 grid(256 * waves_per_simd), block(256); // 4 waves per block -> waves_per_simd per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 12345u, 64);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 12345u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)grid.x * 4 * iters * 16; // wave-instructions
    double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
    printf("%-10s waves/SIMD=%d  %8.3f ms  %.3f G wave-instr/s/SIMD  (cycles per instr at 2.4 GHz: %.2f)\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    return per_simd_per_s;
}

int main() {
    uint32_t *d; hipMalloc(&d, 64);
    for (int w : {4, 8}) {
        run<0>("v_add_u32", d, w); run<2>("v_and_b32", d, w); run<3>("v_bcnt", d, w);
        run<4>("v_bitop3", d, w); run<13>("v_sub_sgpr", d, w);
        run<14>("add_lit", d, w); run<15>("and_lit", d, w); run<16>("and_sgpr", d, w); run<17>("bitop3_s", d, w);
        run<18>("bitop3_v", d, w); run<19>("add_inl", d, w); run<20>("v_xor", d, w); run<21>("v_lshr", d, w); run<22>("bcnt_0", d, w);
        run<23>("STEP7 vgpr (x28/16)", d, w); run<24>("STEP7 sgpr+lit (x28/16)", d, w);
        printf("\n");
    }
    return 0;
}
