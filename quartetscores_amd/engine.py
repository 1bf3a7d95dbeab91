"""Python host above the C-ABI, mirroring the reference's interface for the hot path.

    QuartetCounterLookup  <->  QuartetCounterLookup<CINT>   (QuartetCounterLookup.hpp:24-53)
    QuartetScoreComputer  <->  QuartetScoreComputer<CINT>   (QuartetScoreComputer.hpp:43-80)

Same names, argument meaning and error behaviour, so that the parity tests read like tests
of the reference. All computation happens in libquartetscores_hip.so (HIP, gfx950); this
file only marshals arrays. torch is optional here: it is used when the caller wants the
count table inside a torch tensor (multi-GPU all-reduce over RCCL, see distributed.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Tuple, Union

import numpy as np

from . import _lib, flatten, newick
from ._lib import (QS_ALGO_AUTO, QS_ALGO_GATHER, QS_ALGO_SCATTER, QS_COUNT_OVERWRITE, QS_COUNT_TIMED, QS_COUNT_WIRE16X2, QS_SCORE_QP_EXACT64,  # noqa: F401
                   QS_SCORE_QP_WRAP32, QS_SCORE_ROOT_AS_EDGE, QS_SCORE_SAVEMEM_LOOKUPS)


class QSError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[qs {code}] {msg}")
        self.code = code


def count_bits_for(m: int) -> int:
    """Counter width for m evaluation trees. The reference picks u8/u16/u32/u64 by m
    (QuartetScores.cpp:115-147); the GPU table has 16- and 32-bit cells."""
    return 16 if m < (1 << 16) else 32


# qs_set_tuning values applied to every new Context (tests and A/B runs set entries here; empty in production)
DEFAULT_TUNING = {}
# A/B runs of whole test files / tools without editing them: QS_PY_TUNING="14=1,10=2" (key=value pairs of qs_set_tuning). This is
# the PYTHON harness' switch; the library itself reads no environment variables.
for _kv in filter(None, os.environ.get("QS_PY_TUNING", "").split(",")):
    DEFAULT_TUNING[int(_kv.split("=")[0])] = int(_kv.split("=")[1])


class Context:
    """Thin RAII wrapper of qs_ctx."""

    def __init__(self, n_taxa: int, count_bits: int = 32, device: int = 0, stream: int = 0, d_lo: int = 0, d_hi: int = 0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.qs_create(C.byref(h), n_taxa, count_bits, 0, device, C.c_void_p(stream or None), d_lo, d_hi or n_taxa)
        if rc != 0:
            raise QSError(rc, self.L.qs_last_error(None).decode())
        self.h = h
        self.n = n_taxa
        self.count_bits = count_bits
        self.d_lo, self.d_hi = d_lo, d_hi or n_taxa
        self._attached = None  # keeps an attached torch tensor alive
        for key, value in DEFAULT_TUNING.items():
            self.set_tuning(key, value)

    def close(self):
        if getattr(self, "h", None):
            self.L.qs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise QSError(rc, self.L.qs_last_error(self.h).decode())

    # table
    @property
    def table_tuples(self) -> int:
        return int(self.L.qs_table_tuples(self.h))

    @property
    def table_bytes(self) -> int:
        return int(self.L.qs_table_bytes(self.h))

    def table_alloc(self):
        self._chk(self.L.qs_table_alloc(self.h))

    def table_attach(self, tensor):
        """tensor: a torch CUDA tensor with >= table_bytes bytes (kept alive by this object); None detaches the
        caller-owned table (waits for the context's stream first)."""
        if tensor is None:
            self._chk(self.L.qs_table_attach(self.h, None, 0))
            self._attached = None
            return
        nbytes = tensor.numel() * tensor.element_size()
        self._chk(self.L.qs_table_attach(self.h, C.c_void_p(tensor.data_ptr()), nbytes))
        self._attached = tensor

    def issue_probe(self, iterations: int = 100000) -> float:
        """qs_issue_probe: ns per wave instruction and SIMD of the count kernel's bare instruction slot on this device."""
        out = C.c_float(0)
        self._chk(self.L.qs_issue_probe(self.h, iterations, C.byref(out)))
        return float(out.value)

    def sum_words(self, dst, sources):
        """qs_sum_words: dst += sum of the source tensors, as 32-bit words (torch CUDA tensors of equal size; the sources may
        live on peer devices this process has enabled access to); asynchronous."""
        n = dst.numel() * dst.element_size() // 4
        arr = (C.c_void_p * len(sources))(*[s_.data_ptr() for s_ in sources])
        self._chk(self.L.qs_sum_words(self.h, C.c_void_p(dst.data_ptr()), arr, len(sources), n))

    def table_pack16(self, tensor):
        """u32 table -> u16 table in `tensor` (torch CUDA tensor, >= ceil(cells/2)*4 bytes); asynchronous."""
        self._chk(self.L.qs_table_pack16(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()))

    def wire_attach(self, tensor):
        """Destination of count_batch(..., QS_COUNT_WIRE16X2): table_tuples int32 words (torch CUDA tensor; None detaches)."""
        if tensor is None:
            self._chk(self.L.qs_wire_attach(self.h, None, 0))
        else:
            self._chk(self.L.qs_wire_attach(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()))
        self._wire = tensor

    def table_pack16x2(self, tensor):
        """u32 table -> one word n0 | n1 << 16 per tuple (batches of binary trees holding all taxa); asynchronous."""
        self._chk(self.L.qs_table_pack16x2(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()))

    def table_pack32x2(self, tensor):
        """qs_table_pack32x2: (n0, n1) per tuple as two u32 words into `tensor` (int32, >= 2 * table_tuples elements)."""
        self._chk(self.L.qs_table_pack32x2(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()))

    def unpack32x2(self, src, n_tuples: int, total_trees: int, dst):
        """qs_unpack32x2: n_tuples (n0, n1) pairs of `src` -> [rank][3] u32 tuples in `dst` (n2 = total_trees - n0 - n1)."""
        if src.numel() < 2 * n_tuples or dst.numel() * dst.element_size() < n_tuples * 12:
            raise ValueError("unpack32x2: buffer too small")
        self._chk(self.L.qs_unpack32x2(self.h, C.c_void_p(src.data_ptr()), n_tuples, total_trees, C.c_void_p(dst.data_ptr())))

    def unpack16x2(self, src, n_tuples: int, total_trees: int, dst):
        """n_tuples reduced words -> [tuple][3] u16 cells in `dst` (n2 = total_trees - n0 - n1); asynchronous."""
        if src.numel() * src.element_size() < n_tuples * 4 or dst.numel() * dst.element_size() < n_tuples * 6:
            raise ValueError("unpack16x2: buffer too small")
        self._chk(self.L.qs_unpack16x2(self.h, C.c_void_p(src.data_ptr()), n_tuples, total_trees, C.c_void_p(dst.data_ptr())))

    def table_clear(self):
        self._chk(self.L.qs_table_clear(self.h))

    def table_download(self) -> np.ndarray:
        dt = np.uint32 if self.count_bits == 32 else np.uint16
        out = np.zeros((max(self.table_tuples, 1), 3), dtype=dt)
        self._chk(self.L.qs_table_download(self.h, out.ctypes.data_as(C.c_void_p), self.table_bytes))
        return out[: self.table_tuples]

    def table_upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.qs_table_upload(self.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    # batches
    @staticmethod
    def _batch_struct(b: flatten.TreeBatch, with_nodes=True):
        s = _lib.TreeBatchC()
        s.n_trees = b.n_trees
        keep = [np.ascontiguousarray(x) for x in (b.leaf_off, b.leaf_ids, b.adj_depth, b.node_off, b.rng_off, b.ranges)]
        s.leaf_off, s.leaf_ids, s.adj_depth = (k.ctypes.data for k in keep[:3])
        if with_nodes:
            s.node_off, s.rng_off, s.ranges = (k.ctypes.data for k in keep[3:])
        return s, keep

    def batch_upload(self, b: flatten.TreeBatch, with_nodes=True):
        s, keep = self._batch_struct(b, with_nodes)
        h = C.c_void_p()
        self._chk(self.L.qs_batch_upload(self.h, C.byref(s), C.byref(h)))
        del keep
        return h

    def batch_flags(self, hb) -> int:
        """_lib.QS_BATCH_ALL_TAXA | _lib.QS_BATCH_BINARY as found by the upload's validation."""
        return int(self.L.qs_batch_flags(hb))

    def batch_free(self, hb):
        self.L.qs_batch_free(self.h, hb)

    def count_batch(self, hb, algo=QS_ALGO_AUTO):
        self._chk(self.L.qs_count_batch(self.h, hb, algo))

    def count_trees(self, b: flatten.TreeBatch, algo=QS_ALGO_AUTO):
        s, keep = self._batch_struct(b, True)
        self._chk(self.L.qs_count_trees(self.h, C.byref(s), algo))
        del keep

    def sync(self):
        self._chk(self.L.qs_sync(self.h))

    @property
    def trees_counted(self) -> int:
        return int(self.L.qs_trees_counted(self.h))

    def last_count_ms(self) -> Tuple[float, float, float]:
        out = (C.c_float * 3)()
        self._chk(self.L.qs_last_count_ms(self.h, C.byref(out)))
        return tuple(float(x) for x in out)

    def last_score_ms(self) -> dict:
        """qs_last_score_ms: phases of the most recent score() in ms."""
        out = (C.c_float * 6)()
        self._chk(self.L.qs_last_score_ms(self.h, C.byref(out)))
        keys = ("total", "setup", "pass1", "pass2", "wait_overflow_d2h", "host_finish")
        return {k: float(v) for k, v in zip(keys, out)}

    def last_score_log(self) -> int:
        """Records the last score() logged in single-read mode; 0 = the table was read twice."""
        return int(self.L.qs_last_score_log(self.h))

    def last_score_estimate(self) -> int:
        """Automatic scoring mode: the log size the last score() predicted from its sample; 0 = no estimate ran."""
        return int(self.L.qs_last_score_estimate(self.h))

    def last_count_launches(self) -> int:
        return int(self.L.qs_last_count_launches(self.h))

    def last_count_fix_ms(self) -> float:
        """Share of last_count_ms()[1] spent in the depth-clamp correction kernels (QS_TUNE_DEPTH_CLAMP)."""
        return float(self.L.qs_last_count_fix_ms(self.h))

    def last_count_events(self):
        """[(kind, ms)] of every kernel of the last timed count_batch in launch order; kind: "panel" | "count" | "fix"."""
        ms = (C.c_float * 256)()
        kind = (C.c_uint8 * 256)()
        k = int(self.L.qs_last_count_events(self.h, ms, kind, 256))
        return [(("panel", "count", "fix")[min(int(kind[i]), 2)], float(ms[i])) for i in range(k)]

    def batch_clamp_info(self, hb) -> Tuple[int, int, int]:
        """(trees counted in a class below their own depth bits, their (tree, quartet) corrections, correction workgroups)."""
        out = (C.c_uint64 * 3)()
        self._chk(self.L.qs_batch_clamp_info(hb, C.byref(out)))
        return tuple(int(x) for x in out)

    def set_tuning(self, key: int, value: int):
        """qs_set_tuning: _lib.QS_TUNE_PANEL_SLICE_BYTES / QS_TUNE_GATHER_IMPL / QS_TUNE_PANEL_KERNEL (A/B runs, tests)."""
        self._chk(self.L.qs_set_tuning(self.h, key, value))

    def last_count_variant(self) -> str:
        return self.L.qs_last_count_variant(self.h).decode()

    def lookup(self, abcd: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(abcd, dtype=np.uint16).reshape(-1, 4)
        out = np.zeros((len(q), 3), dtype=np.uint64)
        self._chk(self.L.qs_lookup(self.h, len(q), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    # scoring
    @staticmethod
    def _ref_struct(ref: flatten.RefTree):
        s = _lib.RefTreeC()
        par = np.ascontiguousarray(ref.parent, dtype=np.int32)
        ln = np.ascontiguousarray(ref.leaf_node, dtype=np.uint32)
        s.n_nodes, s.n_taxa, s.parent, s.leaf_node = ref.n_nodes, ref.n_taxa, par.ctypes.data, ln.ctypes.data
        return s, (par, ln)

    def score(self, ref: flatten.RefTree, flags=QS_SCORE_QP_WRAP32):
        s, keep = self._ref_struct(ref)
        lq = np.zeros(ref.n_nodes); qp = np.zeros(ref.n_nodes); eqp = np.zeros(ref.n_nodes)
        bif = C.c_int(0)
        self._chk(self.L.qs_score(self.h, C.byref(s), flags, lq.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p),
                                  eqp.ctypes.data_as(C.c_void_p), C.byref(bif)))
        del keep
        return lq, qp, eqp, bool(bif.value)

    def score_prepare(self, ref: flatten.RefTree, n_trees_total: int = 0):
        """qs_score_prepare: the set-up of a first score() ahead of time (allowed while count kernels are in flight)."""
        s, keep = self._ref_struct(ref)
        self._chk(self.L.qs_score_prepare(self.h, C.byref(s), int(n_trees_total)))
        del keep

    # scoring in steps (table shards / multi-GPU); buffers are torch int64 CUDA tensors owned by the caller
    def score_pair_slots(self, ref: flatten.RefTree) -> int:
        s, keep = self._ref_struct(ref)
        return int(self.L.qs_score_pair_slots(C.byref(s)))

    def score_set_view(self, tensor, count_bits: int, rank_lo: int, n_tuples: int):
        """score_pass1/2 read tuples [rank_lo, rank_lo + n_tuples) from `tensor` (torch CUDA tensor; None = own table)."""
        if tensor is None:
            self._chk(self.L.qs_score_set_view(self.h, None, 0, 0, 0))
            self._view = None
            return
        need = n_tuples * 3 * (count_bits // 8)
        if tensor.numel() * tensor.element_size() < need:
            raise ValueError("view tensor smaller than the tuple range")
        self._chk(self.L.qs_score_set_view(self.h, C.c_void_p(tensor.data_ptr()), count_bits, rank_lo, n_tuples))
        self._view = tensor

    def score_pass1(self, ref: flatten.RefTree, sums, mins):
        s, keep = self._ref_struct(ref)
        self._chk(self.L.qs_score_pass1(self.h, C.byref(s), C.c_void_p(sums.data_ptr()), C.c_void_p(mins.data_ptr())))

    def score_pass2(self, ref: flatten.RefTree, mins, cand):
        s, keep = self._ref_struct(ref)
        self._chk(self.L.qs_score_pass2(self.h, C.byref(s), C.c_void_p(mins.data_ptr()), C.c_void_p(cand.data_ptr())))

    def score_overflow(self, ref: flatten.RefTree, mins, cand) -> np.ndarray:
        """qs_score_overflow: (k, 4) int64 rows (key, q1, q2, q3) = every near-minimal quartet of the node pairs whose
        candidate slots did not suffice in score_pass2 (k = 0 almost always)."""
        s, keep = self._ref_struct(ref)
        p = C.c_void_p()
        k = C.c_uint64(0)
        self._chk(self.L.qs_score_overflow(self.h, C.byref(s), C.c_void_p(mins.data_ptr()), C.c_void_p(cand.data_ptr()), C.byref(p), C.byref(k)))
        if not k.value:
            return np.zeros((0, 4), dtype=np.int64)
        try:
            buf = (C.c_int64 * (4 * k.value)).from_address(p.value)
            return np.frombuffer(buf, dtype=np.int64).reshape(-1, 4).copy()
        finally:
            self.L.qs_free_host(p)

    def score_finish(self, ref: flatten.RefTree, sums_host: np.ndarray, cand_host: np.ndarray, flags=QS_SCORE_QP_WRAP32, extra=None):
        """sums_host: int64[3P]; cand_host: int64[parts, 8P] (gathered over the shards); extra: (k, 4) rows of
        score_overflow (concatenated over the shards) or None."""
        return score_finish_host(ref, sums_host, cand_host, flags, extra=extra, _ctx=self)

    def raw_qic(self, ref: flatten.RefTree, r0: int, nq: int, lex: bool = False):
        """topology + count triple of nq quartets from rank r0 on (lex=False) or from the r0-th 4-subset in lexicographic
        order of the sorted ids on (lex=True: the reference's -q line order)."""
        s, keep = self._ref_struct(ref)
        topo = np.zeros(nq, dtype=np.uint8)
        q = np.zeros((nq, 3), dtype=np.uint64)
        f = self.L.qs_raw_qic_lex if lex else self.L.qs_raw_qic
        self._chk(f(self.h, C.byref(s), r0, nq, topo.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p)))
        del keep
        return topo, q


def score_finish_host(ref: flatten.RefTree, sums_host: np.ndarray, cand_host: np.ndarray, flags=QS_SCORE_QP_WRAP32, extra=None, _ctx=None):
    """qs_score_finish: pure host arithmetic on the reduced per-node-pair sums and the gathered candidates (no device
    needed; `_ctx` only lends its cached reference tree). sums_host: int64[3P]; cand_host: int64[parts, 8P]."""
    L = _lib.load()
    s, keep = Context._ref_struct(ref)
    sums_host = np.ascontiguousarray(sums_host, dtype=np.int64)
    cand_host = np.ascontiguousarray(cand_host, dtype=np.int64)
    parts = cand_host.size // (_lib.QS_SCORE_CAND_SLOTS * (sums_host.size // 3))
    lq = np.zeros(ref.n_nodes); qp = np.zeros(ref.n_nodes); eqp = np.zeros(ref.n_nodes)
    bif = C.c_int(0)
    h = _ctx.h if _ctx is not None else None
    ex = np.ascontiguousarray(extra, dtype=np.int64).reshape(-1, 4) if extra is not None and len(extra) else None
    rc = L.qs_score_finish(h, C.byref(s), flags, sums_host.ctypes.data_as(C.c_void_p), cand_host.ctypes.data_as(C.c_void_p), parts,
                           ex.ctypes.data_as(C.c_void_p) if ex is not None else None, len(ex) if ex is not None else 0,
                           lq.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p), eqp.ctypes.data_as(C.c_void_p), C.byref(bif))
    if rc != 0:
        raise QSError(rc, L.qs_last_error(h).decode())
    del keep
    return lq, qp, eqp, bool(bif.value)


def _read_eval(eval_trees) -> List[str]:
    """evalTreesPath of the reference: a file of ';'-terminated Newick trees. A list of
    Newick strings is accepted as well (tests)."""
    if isinstance(eval_trees, (str, os.PathLike)) and os.path.exists(str(eval_trees)):
        with open(eval_trees) as f:
            return [f.read()]
    if isinstance(eval_trees, str):
        return [eval_trees]
    return list(eval_trees)


class QuartetCounterLookup:
    """QuartetCounterLookup<CINT>(refTree, evalTreesPath, m, savemem) -- counting happens in the
    constructor (QuartetCounterLookup.hpp:245-273). `savemem` is accepted and ignored: the GPU
    table is always the compact C(n,4)x3 layout and holds semantic (1x) counts."""

    def __init__(self, refTree: Union[str, flatten.RefTree], evalTreesPath, m: Optional[int] = None, savemem: bool = False,
                 *, count_bits: Optional[int] = None, device: int = 0, stream: int = 0, algo: int = QS_ALGO_AUTO,
                 batch_trees: int = 4096, table_tensor=None):
        self.ref = refTree if isinstance(refTree, flatten.RefTree) else flatten.flatten_reference(refTree)
        texts = _read_eval(evalTreesPath)
        batch = flatten.flatten_eval_trees(texts, self.ref.name_to_id)  # raises UnknownTaxonError like QCL:218
        self.m = batch.n_trees if m is None else m
        self.savemem = savemem
        bits = count_bits or count_bits_for(self.m)
        self.ctx = Context(self.ref.n_taxa, bits, device, stream)
        if table_tensor is not None:
            self.ctx.table_attach(table_tensor)
            self.ctx.table_clear()
        else:
            self.ctx.table_alloc()
        for lo in range(0, batch.n_trees, batch_trees):
            hi = min(batch.n_trees, lo + batch_trees)
            self.ctx.count_trees(batch.slice(lo, hi) if (lo, hi) != (0, batch.n_trees) else batch, algo)
        # "lookup table size in bytes: ..." (QuartetCounterLookup.hpp:268-272)
        self.lookup_table_bytes = self.ctx.table_bytes
        self._node_to_lookup = {int(v): i for i, v in enumerate(self.ref.leaf_node)}

    def countQuartetOccurrences(self, aIdx: int, bIdx: int, cIdx: int, dIdx: int) -> Tuple[int, int, int]:
        """Arguments are reference-tree NODE indices (QuartetCounterLookup.hpp:299-318)."""
        ids = [self._node_to_lookup[x] for x in (aIdx, bIdx, cIdx, dIdx)]
        r = self.ctx.lookup(np.array(ids, dtype=np.uint16))[0]
        return int(r[0]), int(r[1]), int(r[2])

    def table(self) -> np.ndarray:
        return self.ctx.table_download()


def score_check(ref: flatten.RefTree, flags: int) -> None:
    """qs_score_check without a context (host-only): raises the QSError qs_score would end with for reasons of the reference tree
    alone -- QS_ERR_REFERENCE_THROWS for QS_SCORE_SAVEMEM_LOOKUPS with a rooted reference tree."""
    L = _lib.load()
    parent = np.ascontiguousarray(ref.parent, dtype=np.int32)
    leaf_node = np.ascontiguousarray(ref.leaf_node, dtype=np.uint32)
    s = _lib.RefTreeC(ref.n_nodes, ref.n_taxa, parent.ctypes.data, leaf_node.ctypes.data)
    rc = L.qs_score_check(None, C.byref(s), flags)
    if rc != 0:
        raise QSError(rc, L.qs_last_error(None).decode())


class QuartetScoreComputer:
    """QuartetScoreComputer<CINT>(refTree, evalTreesPath, m, verboseOutput, enforceSmallMem): does
    all the work in its constructor (QuartetScoreComputer.hpp:698-785). Scores are indexed by
    edge; edge e is the edge above node e+1 of the reference tree in preorder."""

    def __init__(self, refTree: Union[str, flatten.RefTree], evalTreesPath, m: Optional[int] = None, verboseOutput: bool = False,
                 enforceSmallMem: bool = False, *, qp_exact64: bool = False, root_as_edge: bool = False, fail_fast: bool = False, log=None, **kw):
        self.ref = refTree if isinstance(refTree, flatten.RefTree) else flatten.flatten_reference(refTree)
        say = log or (lambda s: None)
        n = self.ref.n_taxa
        score_flags = ((QS_SCORE_QP_EXACT64 if qp_exact64 else QS_SCORE_QP_WRAP32) | (QS_SCORE_ROOT_AS_EDGE if root_as_edge else 0) |
                       (QS_SCORE_SAVEMEM_LOOKUPS if enforceSmallMem else 0))
        # what the scoring would refuse because of the reference tree alone (`-s` + a rooted reference: the reference program
        # throws there, after it has counted) is known now. Like the CLI (QuartetScores.cpp: a note, then the reference's order of
        # events): say so, count, and let the scoring raise; fail_fast=True (the CLI's --fail-fast) raises before the counting
        try:
            score_check(self.ref, score_flags)
        except QSError as e:
            if fail_fast or e.code != _lib.QS_ERR_REFERENCE_THROWS:
                raise
            say("note: the scoring of this reference tree will end with the reference's exception (-s with a rooted reference tree); counting first, like the reference")
        self.quartetCounterLookup = QuartetCounterLookup(self.ref, evalTreesPath, m, enforceSmallMem, **kw)
        say(f"There are {self.quartetCounterLookup.m} evaluation trees.")
        say(f"The reference tree has {n} taxa.")
        say(f"lookup table size in bytes: {self.quartetCounterLookup.lookup_table_bytes}")
        say("Finished counting quartets.")
        ctx = self.quartetCounterLookup.ctx
        # root_as_edge: a degree-2 root as a subdivision of one edge instead of the reference's handling (quirk Q5)
        # enforceSmallMem (`-s`): the reference's compact table behind the lookups of a degree-2 root's node pairs -- it throws
        # there (quartet_lookup_table.hpp:79-85), and so does this constructor (QSError, code QS_ERR_REFERENCE_THROWS)
        lq, qp, eqp, bif = ctx.score(self.ref, score_flags)
        self.bifurcating = bif
        say("The reference tree is bifurcating." if bif else "The reference tree is multifurcating.")
        self._lq, self._qp, self._eqp = lq[1:], (qp[1:] if bif else None), (eqp[1:] if bif else None)
        say("Finished computing scores.")

    def getLQICScores(self) -> List[float]:
        return list(self._lq)

    def getQPICScores(self) -> List[float]:
        return [] if self._qp is None else list(self._qp)

    def getEQPICScores(self) -> List[float]:
        return [] if self._eqp is None else list(self._eqp)

    def edge_leafset(self, e: int) -> frozenset:
        """Taxon names below edge e (the child side)."""
        node = self.ref.nodes[e + 1]
        return frozenset(x.name for x in newick.preorder(node) if x.is_leaf)

    def scores_by_bipartition(self):
        """{canonical side: (lq, qp, eqp)} for internal edges, keyed like tests/oracle_api.py."""
        names = self.ref.names
        out = {}
        for e in range(self.ref.n_nodes - 1):
            below = self.edge_leafset(e)
            if len(below) <= 1 or len(below) >= len(names) - 1:
                continue
            other = frozenset(names) - below
            key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
            val = (self._lq[e], None if self._qp is None else self._qp[e], None if self._eqp is None else self._eqp[e])
            while key in out:
                key = frozenset(list(key) + ["#dup"])
            out[key] = val
        return out

    def printRawQICScores(self, rawPath: str, chunk: int = 1 << 20):
        """-q file: "(a,b|c,d): qic" per quartet resolved in the reference, in the reference's own line order: four
        nested loops over its Euler-tour leaves = lexicographic in the sorted lookup ids
        (QuartetScoreComputer.hpp:626-690). QIC via the host libm; %g like operator<<(double)."""
        import itertools
        ctx = self.quartetCounterLookup.ctx
        names = self.ref.names
        total = ctx.table_tuples
        combos = itertools.combinations(range(len(names)), 4)     # lexicographic
        with open(rawPath, "w") as f:
            for r0 in range(0, total, chunk):
                nq = min(chunk, total - r0)
                topo, q = ctx.raw_qic(self.ref, r0, nq, lex=True)
                for i in range(nq):
                    a, b, c, d = next(combos)
                    t = topo[i]
                    if t == 255:
                        continue
                    if t == 0:
                        lab = (names[a], names[b], names[c], names[d])
                    else:
                        lab = (names[a], names[d], names[b], names[c])
                    f.write("(%s,%s|%s,%s): %g\n" % (lab + (log_score(int(q[i, 0]), int(q[i, 1]), int(q[i, 2])),)))


def log_score(q1: int, q2: int, q3: int) -> float:
    """QuartetScoreComputer.hpp:135-159 with the host libm (C doubles via math.log)."""
    if q1 == 0 and q2 == 0 and q3 == 0:
        return 0.0
    s = q1 + q2 + q3
    p1, p2, p3 = q1 / s, q2 / s, q3 / s
    qic = 1.0
    if p1 != 0:
        qic += p1 * math.log(p1) / math.log(3)
    if p2 != 0:
        qic += p2 * math.log(p2) / math.log(3)
    if p3 != 0:
        qic += p3 * math.log(p3) / math.log(3)
    return qic * -1 if (q1 < q2 or q1 < q3) else qic
