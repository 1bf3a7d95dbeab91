"""ctypes binding of the CPU checker under oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. Nothing under quartetscores_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_REF = None


def build_oracle():
    """Compile oracle/ (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.qso_create.restype = C.c_void_p
        L.qso_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.qso_destroy.argtypes = [C.c_void_p]
        L.qso_last_error.restype = C.c_char_p
        L.qso_last_error.argtypes = [C.c_void_p]
        for f in ("qso_n_taxa", "qso_n_edges", "qso_n_nodes", "qso_is_bifurcating"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.qso_n_quartets.restype = C.c_uint64
        L.qso_n_quartets.argtypes = [C.c_void_p]
        L.qso_taxon_name.restype = C.c_char_p
        L.qso_taxon_name.argtypes = [C.c_void_p, C.c_int]
        L.qso_time_count.restype = C.c_double
        L.qso_time_count.argtypes = [C.c_void_p]
        L.qso_time_score.restype = C.c_double
        L.qso_time_score.argtypes = [C.c_void_p]
        L.qso_set_budget.restype = None
        L.qso_set_budget.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.qso_increments_done.restype = C.c_ulonglong
        L.qso_increments_done.argtypes = [C.c_void_p]
        L.qso_count.restype = C.c_int
        L.qso_count.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.qso_lookup.restype = C.c_int
        L.qso_lookup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.qso_get_counts.restype = C.c_int
        L.qso_get_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.qso_score.restype = C.c_int
        L.qso_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qso_get_scores.restype = C.c_int
        L.qso_get_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.qso_edge_side.restype = C.c_int
        L.qso_edge_side.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.qso_raw_qic.restype = C.c_int
        L.qso_raw_qic.argtypes = [C.c_void_p, C.c_char_p]
        L.qso_rank.restype = C.c_uint64
        L.qso_rank.argtypes = [C.c_uint64] * 4
        L.qso_slot.restype = C.c_int
        L.qso_slot.argtypes = [C.c_uint64] * 4
        L.qso_log_score.restype = C.c_double
        L.qso_log_score.argtypes = [C.c_uint64] * 3
        L.qso_cint_bits_for_m.restype = C.c_int
        L.qso_cint_bits_for_m.argtypes = [C.c_uint64]
        _LIB = L
    return _LIB


def reflib():
    """oracle/_ref/libqs_reftable.so: the unmodified reference quartet_lookup_table.hpp."""
    global _REF
    if _REF is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libqs_reftable.so")
        if not os.path.exists(path):
            build_oracle()
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.qsref_tuple_index.restype = C.c_int
        R.qsref_tuple_index.argtypes = [C.c_uint64] * 4
        R.qsref_table_create.restype = C.c_void_p
        R.qsref_table_create.argtypes = [C.c_uint64, C.c_int]
        R.qsref_table_destroy.argtypes = [C.c_void_p, C.c_int]
        R.qsref_table_size.restype = C.c_uint64
        R.qsref_table_size.argtypes = [C.c_void_p, C.c_int]
        R.qsref_lookup_index.restype = C.c_uint64
        R.qsref_lookup_index.argtypes = [C.c_void_p, C.c_int] + [C.c_uint64] * 4
        R.qsref_table_increment.argtypes = [C.c_void_p, C.c_int] + [C.c_uint64] * 4
        R.qsref_table_occurrences.argtypes = [C.c_void_p, C.c_int] + [C.c_uint64] * 4 + [C.c_void_p]
        R.qsref_table_occurrences_checked.restype = C.c_int
        R.qsref_table_occurrences_checked.argtypes = [C.c_void_p, C.c_int] + [C.c_uint64] * 4 + [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64]
        _REF = R
    return _REF


class OracleError(RuntimeError):
    pass


class Oracle:
    """One reference tree + its count table + scores, computed by the CPU restatement."""

    def __init__(self, ref_newick: str):
        L = lib()
        err = C.create_string_buffer(256)
        self._h = L.qso_create(ref_newick.encode(), err, 256)
        if not self._h:
            raise OracleError(err.value.decode())
        self.n = L.qso_n_taxa(self._h)
        self.n_edges = L.qso_n_edges(self._h)
        self.nq = int(L.qso_n_quartets(self._h))
        self.names = [L.qso_taxon_name(self._h, i).decode() for i in range(self.n)]
        self.bifurcating = bool(L.qso_is_bifurcating(self._h))

    def close(self):
        if self._h:
            lib().qso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, eval_text: str, savemem=False, cint_bits=0, nthreads=1, mult=None):
        b = eval_text.encode()
        mp, mn = None, 0
        if mult is not None:
            arr = np.ascontiguousarray(mult, dtype=np.uint64)
            mp, mn = arr.ctypes.data_as(C.c_void_p), len(arr)
        rc = lib().qso_count(self._h, b, len(b), int(savemem), int(cint_bits), int(nthreads), mp, mn)
        if rc != 0:
            raise OracleError(lib().qso_last_error(self._h).decode())
        return lib().qso_time_count(self._h)

    def set_budget(self, seconds: float, prefault: bool = True):
        """bench.py cpu_baseline only: bound the following count() calls to `seconds` (0 = off). A bounded count
        leaves an incomplete table; increments_done() says how much of the work was timed."""
        lib().qso_set_budget(self._h, float(seconds), int(prefault))

    def increments_done(self) -> int:
        return int(lib().qso_increments_done(self._h))

    def counts(self) -> np.ndarray:
        """(C(n,4), 3) uint64: what countQuartetOccurrences returns for every a<b<c<d."""
        out = np.zeros((max(self.nq, 1), 3), dtype=np.uint64)
        rc = lib().qso_get_counts(self._h, out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise OracleError("no counts")
        return out[: self.nq]

    def lookup(self, a, b, c, d):
        out = (C.c_uint64 * 3)()
        rc = lib().qso_lookup(self._h, a, b, c, d, out)
        if rc != 0:      # savemem + a repeated id whose sorted index lies behind the table: the reference throws
            raise OracleError(lib().qso_last_error(self._h).decode())
        return tuple(int(x) for x in out)

    def score(self, nthreads=1, qp_exact64=False):
        rc = lib().qso_score(self._h, int(nthreads), int(qp_exact64))
        if rc != 0:
            raise OracleError(lib().qso_last_error(self._h).decode())
        return lib().qso_time_score(self._h)

    def scores_by_edge(self):
        ne = self.n_edges
        lq = np.zeros(ne); qp = np.zeros(ne); eqp = np.zeros(ne)
        rc = lib().qso_get_scores(self._h, lq.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p),
                                  eqp.ctypes.data_as(C.c_void_p))
        if rc < 0:
            raise OracleError("no scores")
        return lq, (qp if rc == 0 else None), (eqp if rc == 0 else None)

    def edge_side(self, e):
        m = np.zeros(self.n, dtype=np.uint8)
        lib().qso_edge_side(self._h, e, m.ctypes.data_as(C.c_void_p))
        return m

    def scores_by_bipartition(self):
        """{frozenset(names on the side NOT containing lookup id 0... canonical): (lq, qp, eqp)}.

        Key = frozenset of taxon names on the smaller side (ties: the side without
        the lexicographically smallest name). Leaf edges (+inf everywhere) are dropped.
        Two edges with the same bipartition (degree-2 root) are kept under key and key+('#2',).
        """
        lq, qp, eqp = self.scores_by_edge()
        out = {}
        for e in range(self.n_edges):
            side = self.edge_side(e)
            key = canonical_split(self.names, side)
            val = (lq[e], None if qp is None else qp[e], None if eqp is None else eqp[e])
            if len(key) <= 1 or len(key) >= self.n - 1:
                continue
            while key in out:
                key = frozenset(list(key) + ["#dup"])
            out[key] = val
        return out

    def raw_qic(self, path):
        return lib().qso_raw_qic(self._h, path.encode())


def canonical_split(names, member):
    a = sorted(n for n, m in zip(names, member) if m)
    b = sorted(n for n, m in zip(names, member) if not m)
    if len(a) < len(b) or (len(a) == len(b) and min(names) not in a):
        return frozenset(a)
    return frozenset(b)
