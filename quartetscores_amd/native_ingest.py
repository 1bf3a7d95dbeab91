"""ctypes binding of quartetscores_amd/lib/libquartetscores_host.so: the C++ host's multi-threaded Newick ingest
(csrc/host/ingest.hpp) for Python callers. Gives the same arrays as flatten.flatten_eval_trees (tests/test_cli.py
checks the two hosts against each other), ~100x faster than the pure-Python parser."""
from __future__ import annotations

import ctypes as C
import os
from typing import Tuple

import numpy as np

from . import flatten

_LIB = None
ALL = (1 << 64) - 1


class IngestError(RuntimeError):
    pass


def available() -> bool:
    return os.path.exists(_path())


def _path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libquartetscores_host.so")


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_path())
        L.qsh_last_error.restype = C.c_char_p
        L.qsh_ingest.restype = C.c_int
        L.qsh_ingest.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.qsh_ingest_text.restype = C.c_int
        L.qsh_ingest_text.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.qsh_synth_trees.restype = C.c_int
        L.qsh_synth_trees.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_double, C.c_uint,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.qsh_free_text.argtypes = [C.c_void_p]
        L.qsh_batch_n_trees.restype = C.c_uint32
        L.qsh_batch_n_trees.argtypes = [C.c_void_p]
        L.qsh_batch_array.restype = C.c_void_p
        L.qsh_batch_array.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.qsh_batch_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def ingest(ref_path: str, eval_path: str, tree_lo: int = 0, tree_hi: int = ALL, threads: int = 0,
           want_ranges: bool = True) -> Tuple[flatten.TreeBatch, int]:
    """Flatten trees [tree_lo, tree_hi) of `eval_path` against the taxa of the reference tree in `ref_path`.
    Returns (TreeBatch, number of trees in the file). threads = 0: all hardware threads. want_ranges=False skips the
    per-link leaf ranges (node_off all zero, rng_off = [0], ranges empty): only the scatter kernel reads them.
    An unknown taxon or a syntax error raises IngestError (the reference dies with std::out_of_range,
    QuartetCounterLookup.hpp:218)."""
    L = _lib()
    h = C.c_void_p()
    total = C.c_uint64(0)
    rc = L.qsh_ingest(ref_path.encode(), eval_path.encode(), tree_lo, min(tree_hi, ALL), threads, 1 if want_ranges else 0,
                      C.byref(h), C.byref(total))
    if rc != 0:
        raise IngestError(L.qsh_last_error().decode())
    return _take(L, h), int(total.value)


def ingest_text(ref_text: str, eval_text, tree_lo: int = 0, tree_hi: int = ALL, threads: int = 0,
                want_ranges: bool = True) -> Tuple[flatten.TreeBatch, int]:
    """ingest() on Newick text in memory (eval_text: str or bytes holding ';'-terminated trees)."""
    L = _lib()
    h = C.c_void_p()
    total = C.c_uint64(0)
    rb = ref_text.encode() if isinstance(ref_text, str) else ref_text
    eb = eval_text.encode() if isinstance(eval_text, str) else eval_text
    rc = L.qsh_ingest_text(rb, len(rb), eb, len(eb), tree_lo, min(tree_hi, ALL), threads, 1 if want_ranges else 0,
                           C.byref(h), C.byref(total))
    if rc != 0:
        raise IngestError(L.qsh_last_error().decode())
    return _take(L, h), int(total.value)


def synth_trees(n: int, m: int, seed: int, kind: str = "random", ref_text: str = None, mean_nni: float = -1.0,
                threads: int = 0) -> bytes:
    """m seeded synthetic trees on taxa t0..t{n-1} as ';'-terminated Newick lines (csrc/host/synth.hpp):
    kind "random" = uniformly random pairwise joining; "nni" = ref_text + Poisson(mean_nni, default n/8) random NNIs.
    Tree t depends only on (n, seed, t)."""
    L = _lib()
    p = C.c_void_p()
    ln = C.c_uint64(0)
    rc = L.qsh_synth_trees(n, m, seed, {"random": 0, "nni": 1}[kind], ref_text.encode() if ref_text else None, mean_nni,
                           threads, C.byref(p), C.byref(ln))
    if rc != 0:
        raise IngestError(L.qsh_last_error().decode())
    try:
        return C.string_at(p, ln.value)
    finally:
        L.qsh_free_text(p)


def _take(L, h) -> flatten.TreeBatch:
    try:
        def arr(which, dtype):
            n = C.c_uint64(0)
            p = L.qsh_batch_array(h, which, C.byref(n))
            if n.value == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (n.value * np.dtype(dtype).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dtype).copy()
        batch = flatten.TreeBatch(int(L.qsh_batch_n_trees(h)), arr(0, np.uint32), arr(1, np.uint16), arr(2, np.uint16),
                                  arr(3, np.uint32), arr(4, np.uint32), arr(5, np.uint16))
    finally:
        L.qsh_batch_free(h)
    return batch
