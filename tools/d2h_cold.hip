// d2h_cold.hip -- what does the FIRST device-to-host copy of a process / of a stream cost? (qs_score's cold "wait + copies" phase)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(unsigned *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main() {
    const size_t bytes = 22u << 20;
    void *d = nullptr, *h = nullptr, *h2 = nullptr;
    hipStream_t s1, s2;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h2, bytes, hipHostMallocDefault));
    hipLaunchKernelGGL(fill, dim3((unsigned)((bytes / 4 + 255) / 256)), dim3(256), 0, s1, (unsigned *)d, bytes / 4);
    CK(hipStreamSynchronize(s1));
    // some host-to-device traffic first (as the batch uploads of the product)
    CK(hipMemcpyAsync(d, h2, 1 << 20, hipMemcpyHostToDevice, s2));
    CK(hipStreamSynchronize(s2));
    auto timed = [&](const char *what, hipStream_t s, void *dst, size_t n) {
        const double t0 = now();
        hipError_t e = hipMemcpyAsync(dst, d, n, hipMemcpyDeviceToHost, s);
        const double t1 = now();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        const double t2 = now();
        printf("%-64s enqueue %7.3f ms, until done %7.3f ms%s\n", what, t1 - t0, t2 - t0, e == hipSuccess ? "" : "  FAILED");
    };
    timed("1. first D2H of the process: 4 bytes, non-blocking stream", s2, h2, 4);
    timed("2. 22 MB D2H, non-blocking stream, pinned buffer A (first touch)", s2, h, bytes);
    timed("3. the same again", s2, h, bytes);
    timed("4. 22 MB D2H, OTHER stream (its first D2H), pinned buffer A", s1, h, bytes);
    timed("5. the same again", s1, h, bytes);
    timed("6. 22 MB D2H, pinned buffer B (first touch by the device)", s1, h2, bytes);
    timed("7. the same again", s1, h2, bytes);
    unsigned fl = 0;
    { const double t0 = now(); CK(hipMemcpyAsync(&fl, d, 4, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); printf("%-64s %7.3f ms\n", "8. 4 bytes D2H into pageable memory (first)", now() - t0); }
    { const double t0 = now(); CK(hipMemcpyAsync(&fl, d, 4, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); printf("%-64s %7.3f ms\n", "9. the same again", now() - t0); }
    return 0;
}
