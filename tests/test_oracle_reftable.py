"""Pin the oracle's rank/slot restatement against the UNMODIFIED reference header
quartet_lookup_table.hpp (compiled into oracle/_ref/libqs_reftable.so by oracle/Makefile).
CPU only. Skipped when the prebuilt _ref library is absent and /root/reference is too.
"""
import ctypes as C
import itertools

import numpy as np
import pytest

from oracle_api import lib, reflib


@pytest.fixture(scope="module")
def ref():
    R = reflib()
    if R is None:
        pytest.skip("oracle/_ref/libqs_reftable.so not available")
    return R


def test_rank_and_slot_exhaustive_small(ref):
    L = lib()
    n = 14
    h = ref.qsref_table_create(n, 32)
    seen = set()
    for q in itertools.combinations(range(n), 4):
        for p in itertools.permutations(q):
            r_ref = ref.qsref_lookup_index(h, 32, *p)
            assert L.qso_rank(*p) == r_ref
            assert L.qso_slot(*p) == ref.qsref_tuple_index(*p)
        seen.add(L.qso_rank(*q))
    assert seen == set(range(n * (n - 1) * (n - 2) * (n - 3) // 24))
    ref.qsref_table_destroy(h, 32)


def test_rank_large_ids(ref):
    L = lib()
    n = 1024
    h = ref.qsref_table_create(40, 32)  # index arithmetic does not depend on the table size
    rng = np.random.default_rng(5)
    for _ in range(2000):
        q = sorted(rng.choice(n, size=4, replace=False).tolist())
        p = rng.permutation(q).tolist()
        a, b, c, d = q
        expect = d * (d - 1) * (d - 2) * (d - 3) // 24 + c * (c - 1) * (c - 2) // 6 + b * (b - 1) // 2 + a
        assert L.qso_rank(*p) == expect
    ref.qsref_table_destroy(h, 32)


def test_table_size_formula(ref):
    for n, bits in [(8, 8), (32, 16), (40, 32)]:
        h = ref.qsref_table_create(n, bits)
        nq = n * (n - 1) * (n - 2) * (n - 3) // 24
        assert ref.qsref_table_size(h, bits) == nq * 3 * (bits // 8) + 8  # QSC:725
        ref.qsref_table_destroy(h, bits)


def test_savemem_double_count_semantics(ref):
    """A resolved quartet is enumerated at both ends of its middle path; both hits land in
    the same cell of the reference's compact table (SURVEY 3.2 iii / quirk Q1)."""
    h = ref.qsref_table_create(8, 8)
    # tree displays 0 1 | 2 3 : junction of {2,3} sees pair (0,1); junction of {0,1} sees pair (2,3)
    ref.qsref_table_increment(h, 8, 0, 1, 2, 3)
    ref.qsref_table_increment(h, 8, 2, 3, 0, 1)
    out = (C.c_uint64 * 3)()
    ref.qsref_table_occurrences(h, 8, 0, 1, 2, 3, out)
    assert list(out) == [2, 0, 0]
    ref.qsref_table_occurrences(h, 8, 0, 2, 1, 3, out)
    assert list(out) == [0, 2, 0]
    # u8 wrap at 2*150 = 300 -> 44 (Appendix D3)
    for _ in range(149):
        ref.qsref_table_increment(h, 8, 0, 1, 2, 3)
        ref.qsref_table_increment(h, 8, 2, 3, 0, 1)
    ref.qsref_table_occurrences(h, 8, 0, 1, 2, 3, out)
    assert list(out) == [44, 0, 0]
    ref.qsref_table_destroy(h, 8)
