// qs_common.hpp -- shared host/device helpers for the gfx950 quartet engine.
//
// Index arithmetic of the count table (what quartet_lookup_table.hpp:141-212 computes
// per increment on the CPU) is done here once per lane and then kept in registers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QS_HD __host__ __device__ __forceinline__

namespace qs {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCountThreads = 256; // threads per count-kernel workgroup
constexpr int kDB = 8;             // largest-id values (d) handled per workgroup

QS_HD uint64_t binom2(uint64_t x) { return x * (x - 1) / 2; }
QS_HD uint64_t binom3(uint64_t x) { return x < 3 ? 0 : x * (x - 1) * (x - 2) / 6; }
QS_HD uint64_t binom4(uint64_t x) { return x < 4 ? 0 : x * (x - 1) * (x - 2) * (x - 3) / 24; }

// rank of sorted ids s0<s1<s2<s3 in the C(n,4) table
QS_HD uint64_t rank4(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    return binom4(s3) + binom3(s2) + binom2(s1) + s0;
}

// pair index p = C(b,2) + a  (a<b)  ->  (a,b)
QS_HD void unrank2(uint32_t p, uint32_t &a, uint32_t &b) {
    uint32_t bb = (uint32_t)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while ((uint64_t)bb * (bb - 1) / 2 > p) --bb;
    while ((uint64_t)(bb + 1) * bb / 2 <= p) ++bb;
    b = bb;
    a = p - (uint32_t)((uint64_t)bb * (bb - 1) / 2);
}

// rank r -> sorted ids (s0<s1<s2<s3); host and device
QS_HD void unrank4(uint64_t r, uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3) {
    // largest d with C(d,4) <= r
    uint32_t d = (uint32_t)(sqrt(sqrt(24.0 * (double)r + 1.0)) + 1.5);
    while (binom4(d) > r) --d;
    while (binom4((uint64_t)d + 1) <= r) ++d;
    r -= binom4(d);
    uint32_t c = (uint32_t)(cbrt(6.0 * (double)r + 1.0) + 1.0);
    while (binom3(c) > r) --c;
    while (binom3((uint64_t)c + 1) <= r) ++c;
    r -= binom3(c);
    uint32_t a, b;
    unrank2((uint32_t)r, a, b);
    s0 = a; s1 = b; s2 = c; s3 = d;
}

// Table slot of the pairing {x,y}|{rest} among four distinct ids: 0 when the partner of the
// smallest id is the 2nd smallest, 1 when the 3rd, 2 when the largest
// (same convention as the reference's tuple_index, quartet_lookup_table.hpp:87-111).
QS_HD int slot_of_pairing(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    // partner of the minimum
    uint32_t mn = x, partner = y;
    if (y < mn) { mn = y; partner = x; }
    if (z < mn) { mn = z; partner = w; }
    if (w < mn) { mn = w; partner = z; }
    int smaller_than_partner = (x < partner) + (y < partner) + (z < partner) + (w < partner);
    return smaller_than_partner - 1; // position of partner in sorted order minus 1
}

// order-preserving map double -> int64 (signed compare == double compare), so that atomicMin on the
// device and a MIN all-reduce over int64 (RCCL / gloo) both work on scores
QS_HD long long f64_to_sortable(double v) {
    union { double d; long long i; } c; c.d = v;
    return c.i < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)c.i) : c.i;
}
QS_HD double sortable_to_f64(long long s) {
    union { double d; long long i; } c;
    c.i = s < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)s) : s;
    return c.d;
}
constexpr long long kSortableMax = 0x7F7F7F7F7F7F7F7Fll; // "no score yet" (hipMemset byte 0x7F): a huge finite double

} // namespace qs
