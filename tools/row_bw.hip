// row_bw.hip -- read bandwidth of the bundle score kernel's ACCESS PATTERN without its arithmetic: a wave owns 64 rows of
// `row_bytes` (consecutive rows are `pitch` bytes apart, as the rows (b, c, d) of consecutive pairs (c, d) are) and walks them
// in lockstep. Which shape of the loads reaches the streaming rate (tools/read_bw.hip: 6-7 TB/s)?
//   A: lane = row, a chunk of 96 bytes = six 16-byte loads per lane (the kernel's shape), K chunks in flight
//   B: four lanes share a 64-byte segment of a row: one instruction = 16 rows x 64 bytes, four instructions = 64 rows x 64 bytes
//   C: sixteen lanes share 256 bytes of a row: one instruction = 4 rows x 256 bytes
// with plain or non-temporal loads.   tools/bin/row_bw [GB = 24] [row_bytes = 1536]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// wave w of the grid owns rows [64 w, 64 w + 64); persistent: waves stride over the row groups
template <int K, bool NT> __global__ void pat_a(const unsigned char *base, size_t n_rows, size_t row_bytes, unsigned long long *out) {
    const size_t waves = (size_t)gridDim.x * blockDim.x / 64, lane = threadIdx.x & 63;
    unsigned int acc = 0;
    for (size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64; g * 64 < n_rows; g += waves) {
        const unsigned char *row = base + (g * 64 + lane) * row_bytes;
        for (size_t off = 0; off + 96 * K <= row_bytes; off += 96 * K) {
            u32x4 v[6 * K];
#pragma unroll
            for (int j = 0; j < 6 * K; ++j) v[j] = ld<NT>((const u32x4 *)(row + off) + j);
#pragma unroll
            for (int j = 0; j < 6 * K; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}
// LPR lanes per row segment of LPR * 16 bytes; one instruction covers 64 / LPR rows; 64 rows need LPR instructions per segment column
template <int LPR, int U, bool NT> __global__ void pat_b(const unsigned char *base, size_t n_rows, size_t row_bytes, unsigned long long *out) {
    const size_t waves = (size_t)gridDim.x * blockDim.x / 64, lane = threadIdx.x & 63;
    constexpr int RPI = 64 / LPR;            // rows per instruction
    unsigned int acc = 0;
    for (size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64; g * 64 < n_rows; g += waves) {
        const unsigned char *r0 = base + (g * 64 + lane / LPR) * row_bytes + (lane % LPR) * 16;
        for (size_t off = 0; off + (size_t)LPR * 16 * U <= row_bytes; off += (size_t)LPR * 16 * U) {
            u32x4 v[LPR * U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < LPR; ++i) v[u * LPR + i] = ld<NT>((const u32x4 *)(r0 + (size_t)i * RPI * row_bytes + off + (size_t)u * LPR * 16));
#pragma unroll
            for (int j = 0; j < LPR * U; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}
template <typename F> static float timed(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(a, 0); f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 24;
    const size_t row_bytes = argc > 2 ? (size_t)atoi(argv[2]) : 1536;
    const size_t bytes = gb << 30, n_rows = bytes / row_bytes / 64 * 64;
    void *d = nullptr; unsigned long long *out = nullptr;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc((void **)&out, 8)); CK(hipMemset(d, 0x5A, bytes)); CK(hipMemset(out, 0, 8));
    const double moved = (double)n_rows * row_bytes;
    printf("# %zu GB as %zu rows of %zu bytes; a wave owns 64 consecutive rows; TB/s = bytes read / time (best of 3)\n", gb, n_rows, row_bytes);
    const unsigned char *p = (const unsigned char *)d;
    for (int wpc : {8, 16, 32}) {
        const int threads = 512, blocks = 256 * wpc * 64 / threads;
#define RUN(name, kern) { const float ms = timed([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, p, n_rows, row_bytes, out); }); printf("%2d waves/CU  %-58s %.2f TB/s\n", wpc, name, moved / ms * 1e-9); fflush(stdout); }
        RUN("A lane = row, 96-byte chunk (6 x 16 B), 1 chunk in flight", (pat_a<1, false>));
        RUN("A ... 2 chunks in flight", (pat_a<2, false>));
        RUN("A ... 4 chunks in flight", (pat_a<4, false>));
        RUN("A ... 2 chunks in flight, non-temporal", (pat_a<2, true>));
        RUN("B 4 lanes x 16 B = 64 B of a row, 16 rows per instruction, U=1", (pat_b<4, 1, false>));
        RUN("B ... U=2 (128 B of every row in flight)", (pat_b<4, 2, false>));
        RUN("B ... U=4", (pat_b<4, 4, false>));
        RUN("B ... U=2, non-temporal", (pat_b<4, 2, true>));
        RUN("C 16 lanes x 16 B = 256 B of a row, 4 rows per instruction, U=1", (pat_b<16, 1, false>));
        RUN("C ... U=1, non-temporal", (pat_b<16, 1, true>));
        RUN("D 8 lanes x 16 B = 128 B of a row, 8 rows per instruction, U=1", (pat_b<8, 1, false>));
        RUN("D ... U=2", (pat_b<8, 2, false>));
    }
    return 0;
}
