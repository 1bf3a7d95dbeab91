// read_bw.hip -- what does a plain read-only streaming kernel reach on this box? (the practical roof of the score passes)
//   tools/bin/read_bw [GB = 32]
// Variants: uint4 loads per lane, U loads in flight per lane, plain or non-temporal, W waves per workgroup, persistent grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void read_kernel(const u32x4 *__restrict__ p, size_t n16, unsigned long long *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}
template <int U, bool NT> static float run(const u32x4 *p, size_t n16, unsigned long long *out, int blocks, int threads) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(threads), 0, 0, p, n16, out);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char **argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 32;
    const size_t bytes = gb << 30, n16 = bytes / 16;
    void *d = nullptr; unsigned long long *out = nullptr;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc((void **)&out, 8));
    CK(hipMemset(d, 0x5A, bytes)); CK(hipMemset(out, 0, 8));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("# %zu GB, %d CUs; best of 4; TB/s = bytes / time\n", gb, cus);
    for (int threads : {256, 512, 1024})
        for (int per_cu : {1, 2, 4, 8}) {
            const int blocks = cus * per_cu;
            if ((size_t)threads * per_cu > 2048) continue;
            const float t1 = run<1, false>((const u32x4 *)d, n16, out, blocks, threads);
            const float t4 = run<4, false>((const u32x4 *)d, n16, out, blocks, threads);
            const float t8 = run<8, false>((const u32x4 *)d, n16, out, blocks, threads);
            const float n4 = run<4, true>((const u32x4 *)d, n16, out, blocks, threads);
            const float n8 = run<8, true>((const u32x4 *)d, n16, out, blocks, threads);
            printf("threads %4d x %d per CU (%2d waves/CU): U=1 %.2f  U=4 %.2f  U=8 %.2f  U=4 nt %.2f  U=8 nt %.2f TB/s\n", threads, per_cu, threads * per_cu / 64,
                   bytes / t1 * 1e-9, bytes / t4 * 1e-9, bytes / t8 * 1e-9, bytes / n4 * 1e-9, bytes / n8 * 1e-9);
            fflush(stdout);
        }
    return 0;
}
