// ref_table_shim.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/qs_oracle.c header).
//
// Thin extern "C" wrapper around the UNMODIFIED reference header
// /root/reference/src/quartet_lookup_table.hpp, included from where it lies
// (-I/root/reference/src on the compiler command line, oracle/Makefile). It is
// the one reference source on the hot path that compiles from its own file
// (it depends on the C++ standard library only); everything else on the path
// includes genesis v0.16.0, which is absent from this image, so the rest of the
// reference is unbuildable here and is restated in qs_oracle.c instead.
//
// Output goes to oracle/_ref/libqs_reftable.so (git-ignored, travels with gpurun).
// Used by tests/test_oracle_reftable.py to pin qso_rank/qso_slot and the compact
// table's increment/lookup semantics (incl. the savemem 2x quirk).
#include "quartet_lookup_table.hpp"

#include <cstdint>
#include <cstddef>
#include <cstring>
#include <stdexcept>

namespace {
template <typename T> struct Box { QuartetLookupTable<T> t; };
}

extern "C" {

// tuple_index is a pure function of (a,b,c,d): quartet_lookup_table.hpp:87-111
int qsref_tuple_index(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    static QuartetLookupTable<uint8_t> t(4);
    return (int)t.tuple_index(a, b, c, d);
}

void *qsref_table_create(uint64_t n, int bits) {
    switch (bits) {
    case 8: { auto *b = new Box<uint8_t>(); b->t.init(n); return b; }
    case 16: { auto *b = new Box<uint16_t>(); b->t.init(n); return b; }
    default: { auto *b = new Box<uint32_t>(); b->t.init(n); return b; }
    }
}

void qsref_table_destroy(void *h, int bits) {
    switch (bits) {
    case 8: delete (Box<uint8_t> *)h; break;
    case 16: delete (Box<uint16_t> *)h; break;
    default: delete (Box<uint32_t> *)h; break;
    }
}

uint64_t qsref_table_size(void *h, int bits) {
    switch (bits) {
    case 8: return ((Box<uint8_t> *)h)->t.size();
    case 16: return ((Box<uint16_t> *)h)->t.size();
    default: return ((Box<uint32_t> *)h)->t.size();
    }
}

// lookup_index_ is private; recover it as the element distance from tuple 0
// (get_tuple(0,1,2,3) is element 0 of the vector: rank = 0).
#define QSREF_INDEX(T)                                                              \
    {                                                                               \
        auto &tab = ((Box<T> *)h)->t;                                               \
        auto *base = &tab.get_tuple(0, 1, 2, 3);                                    \
        auto *p = &tab.get_tuple(a, b, c, d);                                       \
        return (uint64_t)(p - base);                                                \
    }
uint64_t qsref_lookup_index(void *h, int bits, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    switch (bits) {
    case 8: QSREF_INDEX(uint8_t)
    case 16: QSREF_INDEX(uint16_t)
    default: QSREF_INDEX(uint32_t)
    }
}

// The savemem increment exactly as QuartetCounterLookup.hpp:84-87 performs it.
#define QSREF_INC(T)                                                                \
    {                                                                               \
        auto &tab = ((Box<T> *)h)->t;                                               \
        auto &tuple = tab.get_tuple(a, a2, b, c);                                   \
        size_t idx = tab.tuple_index(a, a2, b, c);                                  \
        tuple[idx]++;                                                               \
        return;                                                                     \
    }
void qsref_table_increment(void *h, int bits, uint64_t a, uint64_t a2, uint64_t b, uint64_t c) {
    switch (bits) {
    case 8: QSREF_INC(uint8_t)
    case 16: QSREF_INC(uint16_t)
    default: QSREF_INC(uint32_t)
    }
}

// The savemem lookup exactly as QuartetCounterLookup.hpp:303-311 performs it.
#define QSREF_OCC(T)                                                                \
    {                                                                               \
        auto const &tab = ((Box<T> *)h)->t;                                         \
        const auto &tuple = tab.get_tuple(a, b, c, d);                              \
        out3[0] = tuple[tab.tuple_index(a, b, c, d)];                               \
        out3[1] = tuple[tab.tuple_index(a, c, b, d)];                               \
        out3[2] = tuple[tab.tuple_index(a, d, b, c)];                               \
        return;                                                                     \
    }
void qsref_table_occurrences(void *h, int bits, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t *out3) {
    switch (bits) {
    case 8: QSREF_OCC(uint8_t)
    case 16: QSREF_OCC(uint16_t)
    default: QSREF_OCC(uint32_t)
    }
}

// The same lookup for ids that may REPEAT (a rooted reference tree makes QuartetScoreComputer.hpp:393-396 call
// countQuartetOccurrences with b == c or b == d): the const get_tuple (quartet_lookup_table.hpp:79-85) throws
// std::runtime_error when the index of the sorted ids falls behind the table. Returns 0 and the three cells, or 1 and
// the exception's what() in msg -- no exception crosses the C boundary.
#define QSREF_OCC_CHECKED(T)                                                        \
    {                                                                               \
        auto const &tab = ((Box<T> *)h)->t;                                         \
        const auto &tuple = tab.get_tuple(a, b, c, d);                              \
        out3[0] = tuple[tab.tuple_index(a, b, c, d)];                               \
        out3[1] = tuple[tab.tuple_index(a, c, b, d)];                               \
        out3[2] = tuple[tab.tuple_index(a, d, b, c)];                               \
        if (index_out) *index_out = (uint64_t)(&tuple - &tab.get_tuple(0, 1, 2, 3)); \
        return 0;                                                                   \
    }
int qsref_table_occurrences_checked(void *h, int bits, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t *out3,
                                    uint64_t *index_out, char *msg, uint64_t msg_len) {
    try {
        switch (bits) {
        case 8: QSREF_OCC_CHECKED(uint8_t)
        case 16: QSREF_OCC_CHECKED(uint16_t)
        default: QSREF_OCC_CHECKED(uint32_t)
        }
    } catch (std::runtime_error const &e) {
        if (msg && msg_len) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
        return 1;
    }
}

} // extern "C"
