"""CPU-only tests of the host logic: Newick I/O, flattening, rank arithmetic, the SWAR
arithmetic (numpy transcription) and that the C-ABI library loads and exports every symbol
include/quartetscores_hip.h declares. No compute calls into the library (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bruteforce
import emulate
from helpers import quads_in_rank_order, rank4
from oracle_api import Oracle
from quartetscores_amd import flatten, newick, ranks, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newick_roundtrip_and_dialect():
    t = newick.parse_tree("((a:0.1,'b c':2e-3)x:1,[comment](c,d)[another],e);")
    names = [x.name for x in newick.preorder(t) if x.is_leaf]
    assert names == ["a", "b c", "c", "d", "e"]
    assert newick.write(t) == "((a:0.1,'b c':2e-3)x:1,(c,d),e);"
    many = list(newick.parse_trees("(a,b,c);\n(a,(b,c));\n\n"))
    assert len(many) == 2
    with pytest.raises(newick.NewickError):
        newick.parse_tree("((a,b),c")
    deep = "(" * 3000 + "x" + ",y)" * 3000 + ";"
    assert len(newick.preorder(newick.parse_tree(deep))) == 6001  # no recursion limit


def test_reference_flatten_ids_are_dfs_order(golden):
    ref = flatten.flatten_reference(golden["D1"]["ref"])
    assert ref.names == ["t7", "t2", "t0", "t1", "t5", "t6", "t3", "t4"]  # = oracle lookup order
    o = Oracle(golden["D1"]["ref"])
    assert o.names == ref.names
    assert ref.parent[0] == -1 and (ref.parent[1:] >= 0).all()


@pytest.mark.parametrize("n,m,dropout,collapse,rooted,seed",
                         [(8, 20, 0, 0, False, 1), (12, 40, 0.3, 0.3, False, 2), (10, 30, 0, 0.5, True, 3),
                          (16, 30, 0.2, 0, True, 4), (9, 10, 0, 0, True, 5)])
def test_flatten_plus_fourpoint_matches_oracle(n, m, dropout, collapse, rooted, seed):
    ref_nw = synth.reference_tree(n, seed)
    trees = synth.tree_set(n, m, 100 + seed, dropout=dropout, collapse=collapse, rooted=rooted)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    assert batch.n_trees == m and len(batch.leaf_ids) == len(batch.adj_depth)
    T = emulate.counts_from_batch(batch, n)
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    assert o.names == ref.names
    assert (T == o.counts()).all()


def test_flatten_ranges_match_oracle_enumeration():
    """The circular ranges handed to the scatter kernel enumerate exactly the oracle's hits."""
    n, m = 9, 12
    ref_nw = synth.reference_tree(n, 11)
    trees = synth.tree_set(n, m, 12, collapse=0.3, dropout=0.2)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    T = np.zeros((ranks.n_quartets(n), 3), dtype=np.uint64)
    for t in range(batch.n_trees):
        lo, hi = int(batch.leaf_off[t]), int(batch.leaf_off[t + 1])
        ids = batch.leaf_ids[lo:hi].astype(int)
        L = hi - lo
        for v in range(int(batch.node_off[t]), int(batch.node_off[t + 1])):
            links = [(int(batch.ranges[2 * k]), int(batch.ranges[2 * k + 1]))
                     for k in range(int(batch.rng_off[v]), int(batch.rng_off[v + 1]))]
            sets = [[ids[(s + i) % L] for i in range((e - s) % L)] for (s, e) in links]
            assert sum(len(s) for s in sets) == L  # the links of a node partition the leaves
            for i1 in range(len(sets)):
                for i2 in range(i1 + 1, len(sets)):
                    for i3 in range(i2 + 1, len(sets)):
                        tri = (sets[i1], sets[i2], sets[i3])
                        for o_ in range(3):
                            P, Q, R = tri[o_], tri[(o_ + 1) % 3], tri[(o_ + 2) % 3]
                            for x in range(len(P)):
                                for y in range(x + 1, len(P)):
                                    for b in Q:
                                        for c in R:
                                            a, a2 = P[x], P[y]
                                            if min(a, a2) > min(b, c):
                                                continue
                                            s = sorted((a, a2, b, c))
                                            partner = a2 if a == s[0] else a
                                            T[int(rank4(*s)), s.index(partner) - 1] += 1
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    assert (T == o.counts()).all()


def test_unknown_taxon_raises():
    ref = flatten.flatten_reference("((a,b),(c,d),e);")
    with pytest.raises(flatten.UnknownTaxonError):
        flatten.flatten_eval_trees(["((a,b),(c,zzz),e);"], ref.name_to_id)


def test_recentring_bounds_depth():
    n = 128
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    cat += ";"
    ref = flatten.flatten_reference(synth.reference_tree(n, 1))
    b = flatten.flatten_eval_trees([cat], ref.name_to_id)
    assert int(b.adj_depth.max()) <= n // 2 + 1
    b2 = flatten.flatten_eval_trees([cat], ref.name_to_id, recentre=False)
    assert int(b2.adj_depth.max()) >= n - 4
    # counts do not depend on the rooting
    n2 = 14
    cat2 = "(t0,t1)"
    for i in range(2, n2):
        cat2 = "(" + cat2 + f",t{i})"
    ref2 = flatten.flatten_reference(synth.reference_tree(n2, 2))
    c1 = emulate.counts_from_batch(flatten.flatten_eval_trees([cat2 + ";"], ref2.name_to_id), n2)
    c2 = emulate.counts_from_batch(flatten.flatten_eval_trees([cat2 + ";"], ref2.name_to_id, recentre=False), n2)
    assert (c1 == c2).all() and c1.sum() == ranks.n_quartets(n2)


def test_rank_unrank_roundtrip():
    for n in (8, 33, 128):
        q = quads_in_rank_order(n) if n <= 33 else None
        r = np.arange(ranks.n_quartets(n), dtype=np.int64) if n <= 33 else np.random.default_rng(0).integers(
            0, ranks.n_quartets(n), 5000)
        ids = ranks.unrank4_np(r)
        assert (ranks.rank4(ids[:, 0], ids[:, 1], ids[:, 2], ids[:, 3]) == r).all()
        assert (np.diff(ids, axis=1) > 0).all()
        if q is not None:
            assert (ids == q).all()


@pytest.mark.parametrize("bits,mode", [(8, 0), (8, 1), (8, 2), (16, 0), (16, 1), (16, 2)])
def test_swar_step_matches_scalar(bits, mode):
    """The packed comparison of qs_count.hip::swar_step equals the per-tree scalar rule, for
    every depth value the dispatcher admits in that mode (incl. the limits)."""
    rng = np.random.default_rng(bits * 10 + mode)
    fields = 4 if bits == 8 else 2
    lim = {(8, 0): 63, (8, 1): 63, (8, 2): 31, (16, 0): 16383, (16, 1): 16383, (16, 2): 8191}[(bits, mode)]
    flag = 0x20 if bits == 8 else 0x2000
    N = 4000
    # tree-metric-like inputs: draw depths, then force the two smaller sums equal as in a real tree
    vals = rng.integers(0, lim + 1, size=(N, fields, 6))
    vals[: N // 8] = rng.choice([0, lim], size=(N // 8, fields, 6))  # extremes
    if mode == 0:  # binary: S1 != S2 always in a resolved tree unless topology 3; keep generic
        pass
    present = np.ones((N, fields, 4), dtype=bool)
    if mode == 2:
        present = rng.random((N, fields, 4)) > 0.2
    # pairs: ab, cd, ac, bd, ad, bc  <- taxa a,b,c,d = 0..3
    pair_taxa = [(0, 1), (2, 3), (0, 2), (1, 3), (0, 3), (1, 2)]
    for k, (x, y) in enumerate(pair_taxa):
        miss = ~(present[:, :, x] & present[:, :, y])
        vals[:, :, k] = np.where(miss, flag, vals[:, :, k])
    if mode != 2:
        # emulate the tree-metric property "the two smaller sums are equal": make S3 = min(S1,S2) where possible
        pass
    words = [np.zeros(N, dtype=np.uint64) for _ in range(6)]
    for f in range(fields):
        for k in range(6):
            words[k] |= vals[:, f, k].astype(np.uint64) << np.uint64(bits * f)
    n0, n1, n2 = emulate.swar_step(bits, mode, *words)
    ab, cd, ac, bd, ad, bc = (vals[:, :, k] for k in range(6))
    ok = present.all(axis=2)
    s1, s2, s3 = ab + cd, ac + bd, ad + bc
    assert (n0 == (ok & (s1 > s2)).sum(axis=1)).all()
    assert (n1 == (ok & (s2 > s1)).sum(axis=1)).all()
    if mode != 0:
        assert (n2 == (ok & (s1 == s2) & (s3 > s1)).sum(axis=1)).all()


def test_library_exports_every_declared_symbol():
    from quartetscores_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "quartetscores_hip.h")).read()
    declared = set(re.findall(r"\b(qs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"qs_ctx", "qs_device_batch", "qs_tree_batch", "qs_ref_tree"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    L = C.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    L.qs_version.restype = C.c_char_p
    assert b"gfx950" in L.qs_version()


def test_qs_create_fails_loudly_without_gpu():
    """No GPU in the CPU test container: the library must refuse, not fall back."""
    from quartetscores_amd import _lib
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    h = C.c_void_p()
    rc = L.qs_create(C.byref(h), 8, 32, 0, 0, None, 0, 8)
    assert rc == _lib.QS_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.qs_last_error(None)


def test_header_is_plain_c(tmp_path):
    """include/quartetscores_hip.h is the drop-in boundary: it must compile as C99 (no C++, no torch types) and a C
    program must link against the library with nothing but the header."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "quartetscores_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { qs_ctx *c = 0; int rc = qs_create(&c, 8, 32, 0, 0, 0, 0, 0);\n'
                   '  printf("%s|%d|%s\\n", qs_version(), rc, qs_last_error(c)); if (c) qs_destroy(c); return 0; }\n')
    exe = tmp_path / "use"
    lib_dir = os.path.join(ROOT, "quartetscores_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-lquartetscores_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "quartetscores_amd" in p.stdout
    import torch
    if not torch.cuda.is_available():       # no device here: the C caller sees the loud failure too
        assert "|-6|" in p.stdout and "no HIP device" in p.stdout


def test_python_closed_form_scores_match_the_oracle():
    """tests/emulate.scores_from_table (used as the expected value for hand-made tables on the GPU) against the oracle."""
    import emulate
    from oracle_api import Oracle
    from quartetscores_amd import flatten, newick, synth
    n = 12
    ref_nw = synth.reference_tree(n, 51)
    trees = synth.tree_set(n, 30, 52, collapse=0.2)
    ref = flatten.flatten_reference(ref_nw)
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    o.score()
    want = o.scores_by_bipartition()
    lq, qp, eqp = emulate.scores_from_table(ref, o.counts())
    names = ref.names
    seen = 0
    for e in range(ref.n_nodes - 1):
        below = frozenset(x.name for x in newick.preorder(ref.nodes[e + 1]) if x.is_leaf)
        if len(below) <= 1 or len(below) >= n - 1:
            continue
        other = frozenset(names) - below
        key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
        assert (lq[e + 1], qp[e + 1], eqp[e + 1]) == want[key]
        seen += 1
    assert seen == len(want)


def test_score_plan_covers_every_rank_range_exactly_once():
    """qs_score_plan (the host planner of the score bundle kernel: pure arithmetic, no device): for whole tables, table
    shards (ranges of the largest id) and arbitrary rank ranges -- the views of a reduce-scattered table, which start and
    end inside rows -- the full rows of the plan plus its partial rows are pairwise disjoint and their union is exactly
    [rank_lo, rank_lo + n_tuples)."""
    from quartetscores_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)     # (no device: the planner is host code)
    L.qs_score_plan.restype = C.c_int
    L.qs_score_plan.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]

    def plan(n, r0, cnt):
        plo = np.zeros(n, dtype=np.uint32); pc = np.zeros(n, dtype=np.uint32); parts = np.zeros(4, dtype=np.uint64)
        assert L.qs_score_plan(n, r0, cnt, plo.ctypes.data, pc.ctypes.data, parts.ctypes.data) == 0
        return plo, pc, parts

    def covered(n, r0, cnt):
        plo, pc, parts = plan(n, r0, cnt)
        hit = np.zeros(ranks.n_quartets(n) + 1, dtype=np.int32)
        for b in range(n):
            assert pc[b] == 0 or 1 <= b <= n - 3
            for p in range(int(plo[b]), int(plo[b]) + int(pc[b])):
                dd = int((1 + (1 + 8 * p) ** 0.5) / 2)          # pair index -> (c', d'), c' < d': p = C(d',2) + c'
                while dd * (dd - 1) // 2 > p:
                    dd -= 1
                while (dd + 1) * dd // 2 <= p:
                    dd += 1
                cc = p - dd * (dd - 1) // 2
                c, d = b + 1 + cc, b + 1 + dd
                assert b < c < d < n
                start = ranks.rank4(0, b, c, d)
                hit[start:start + b] += 1
        for lo, k in ((int(parts[0]), int(parts[1])), (int(parts[2]), int(parts[3]))):
            hit[lo:lo + k] += 1
        want = np.zeros_like(hit)
        want[r0:r0 + cnt] = 1
        assert (hit == want).all(), (n, r0, cnt)

    rng = np.random.default_rng(77)
    for n in (4, 5, 6, 9, 17, 24):
        nq = ranks.n_quartets(n)
        covered(n, 0, nq)
        covered(n, 0, 0)
        for d_lo in range(3, n):                                   # table shards: d in [d_lo, d_hi)
            d_hi = int(rng.integers(d_lo + 1, n + 1))
            covered(n, ranks.n_quartets(d_lo), ranks.n_quartets(d_hi) - ranks.n_quartets(d_lo))
        for _ in range(60):                                        # arbitrary ranges, also inside one row
            r0 = int(rng.integers(0, nq))
            cnt = int(rng.integers(0, nq - r0 + 1)) if rng.random() < 0.5 else int(rng.integers(0, min(nq - r0, 30) + 1))
            covered(n, r0, cnt)
        for r0 in range(min(nq, 40)):                              # every short range near the start (tiny rows)
            for cnt in range(0, min(nq - r0, 12)):
                covered(n, r0, cnt)
    # argument checks
    buf = np.zeros(8, dtype=np.uint64)
    assert L.qs_score_plan(3, 0, 0, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data) != 0
    assert L.qs_score_plan(8, 0, ranks.n_quartets(8) + 1, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data) != 0


# ---- rooted reference tree + the reference's memory-efficient table ---------------------------------
ROOTED_CASES = [("D4", None, 0)] + [(f"random{n}", n, seed) for n, seed in ((4, 5), (7, 81), (12, 82), (23, 83), (40, 84))]


def rooted_case(golden, name, n, seed):
    if name == "D4":
        return golden["D4"]["ref"], golden["D1"]["eval"]
    return synth.random_tree(n, np.random.default_rng(seed), rooted=True), synth.tree_set(n, 12, seed + 100, collapse=0.1)


@pytest.mark.parametrize("name,n,seed", ROOTED_CASES)
def test_rooted_reference_compact_mode_host_logic_matches_oracle(golden, name, n, seed):
    """`-s` (QS_SCORE_SAVEMEM_LOOKUPS) with a degree-2 reference root. The reference's scoring loop then looks up ids that
    repeat (QuartetScoreComputer.hpp:393-396), its compact table sorts them and throws std::runtime_error for an index behind
    the table (quartet_lookup_table.hpp:79-85; pinned on the unmodified header in tests/test_oracle_reftable.py) -- for EVERY
    rooted reference tree, so the reference's run ends there. The oracle restates that (sequential order, -t 1), and the
    product's host logic (qs_score_finish, no device involved) must report the same exception text: the first throwing call
    in the reference's order. Without the flag the same call scores the root's pairs like the n^4 table does."""
    from oracle_api import OracleError
    from quartetscores_amd import _lib, engine
    ref_nw, trees = rooted_case(golden, name, n, seed)
    eval_text = trees if isinstance(trees, str) else "\n".join(trees)
    o = Oracle(ref_nw)
    o.count(eval_text, savemem=True, cint_bits=16)
    with pytest.raises(OracleError) as eo:
        o.score(nthreads=1)
    nq = o.nq
    m = re.fullmatch(r"id = (\d+), but quartet_lookup_\.size\(\) = (\d+)", str(eo.value))
    assert m and int(m.group(2)) == nq and int(m.group(1)) >= nq
    ref = flatten.flatten_reference(ref_nw)
    P = ((ref.n_nodes - ref.n_taxa) ** 2)
    sums = np.zeros(3 * P, dtype=np.int64)
    cand = np.full((1, _lib.QS_SCORE_CAND_SLOTS * P), -1, dtype=np.int64)
    with pytest.raises(engine.QSError) as eg:
        engine.score_finish_host(ref, sums, cand, _lib.QS_SCORE_SAVEMEM_LOOKUPS)
    assert eg.value.code == _lib.QS_ERR_REFERENCE_THROWS
    assert str(eg.value).endswith(str(eo.value)), (str(eg.value), str(eo.value))
    # round 5: the same answer from the reference tree alone, before anything is counted (qs_score_check without a context)
    with pytest.raises(engine.QSError) as ec:
        engine.score_check(ref, _lib.QS_SCORE_SAVEMEM_LOOKUPS)
    assert ec.value.code == _lib.QS_ERR_REFERENCE_THROWS and str(ec.value).endswith(str(eo.value))
    engine.score_check(ref, 0)
    engine.score_check(ref, _lib.QS_SCORE_SAVEMEM_LOOKUPS | _lib.QS_SCORE_ROOT_AS_EDGE)
    # no flag, or the root treated as a point on an edge: no exception (the n^4 table's behaviour / no repeated ids at all)
    engine.score_finish_host(ref, sums, cand, 0)
    engine.score_finish_host(ref, sums, cand, _lib.QS_SCORE_SAVEMEM_LOOKUPS | _lib.QS_SCORE_ROOT_AS_EDGE)
    o.close()


def test_compact_mode_flag_changes_nothing_for_an_unrooted_reference(golden):
    from quartetscores_amd import _lib, engine
    ref = flatten.flatten_reference(golden["D1"]["ref"])
    P = ((ref.n_nodes - ref.n_taxa) ** 2)
    sums = np.arange(3 * P, dtype=np.int64) % 7
    cand = np.full((1, _lib.QS_SCORE_CAND_SLOTS * P), -1, dtype=np.int64)
    a = engine.score_finish_host(ref, sums, cand, 0)
    b = engine.score_finish_host(ref, sums, cand, _lib.QS_SCORE_SAVEMEM_LOOKUPS)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y, equal_nan=True)
    # and the oracle's savemem mode scores an unrooted reference without throwing
    o = Oracle(golden["D1"]["ref"])
    ev = golden["D1"]["eval"]
    o.count(ev if isinstance(ev, str) else "\n".join(ev), savemem=True, cint_bits=16)
    o.score()
    o.close()


def test_finish_evaluates_a_flagged_candidate_in_both_orders(golden):
    """A candidate slot flagged kCandSwap (bit 63) marks a quartet the reference evaluates twice for a degree-2 root, the second
    time with q2 and q3 exchanged (QuartetScoreComputer.hpp:393-396,417-454): log_score sums p1 log p1 + p2 log p2 + p3 log p3
    in that order (:141-156), so the two values can differ in the last bits and std::min keeps the smaller. qs_score_finish
    (pure host) takes both; with QS_SCORE_ROOT_AS_EDGE, or without the flag, only the stored order."""
    from quartetscores_amd import _lib, engine
    ref = flatten.flatten_reference(golden["D4"]["ref"])
    ni = ref.n_nodes - ref.n_taxa
    P = ni * ni
    # a triple whose two orders give different doubles
    trip = None
    for q1 in range(1, 60):
        for q2 in range(1, 60):
            for q3 in range(q2 + 1, 60):
                if engine.log_score(q1, q2, q3) != engine.log_score(q1, q3, q2):
                    trip = (q1, q2, q3)
                    break
            if trip:
                break
        if trip:
            break
    assert trip, "no order-sensitive triple below 60?"
    a, b = engine.log_score(*trip), engine.log_score(trip[0], trip[2], trip[1])
    packed = (trip[0] << 42) | (trip[1] << 21) | trip[2]
    sums = np.zeros(3 * P, dtype=np.int64)
    key = 0 * ni + 1                                    # some pair of inner nodes (iu < iv)
    for flagged, flags, want in ((False, 0, a), (True, 0, min(a, b)), (True, _lib.QS_SCORE_ROOT_AS_EDGE, a)):
        cand = np.full((1, _lib.QS_SCORE_CAND_SLOTS * P), -1, dtype=np.int64)
        word = packed | ((1 << 63) if flagged else 0)
        cand[0, _lib.QS_SCORE_CAND_SLOTS * key] = np.array([word], dtype=np.uint64).view(np.int64)[0]
        lq, _, _, _ = engine.score_finish_host(ref, sums, cand, flags)
        got = lq[np.isfinite(lq)]
        assert len(got) and (got == want).all(), (flagged, flags, got, want)
    # the same through an overflow-list record (key word bit 32)
    cand = np.full((1, _lib.QS_SCORE_CAND_SLOTS * P), -1, dtype=np.int64)
    extra = np.array([[key | (1 << 32), trip[0], trip[1], trip[2]]], dtype=np.int64)
    lq, _, _, _ = engine.score_finish_host(ref, sums, cand, 0, extra=extra)
    assert (lq[np.isfinite(lq)] == min(a, b)).all()


@pytest.mark.parametrize("n,m,clamp,seed,kw", [(16, 20, 2, 1, {}), (16, 20, 3, 2, {}), (20, 15, 2, 3, {"collapse": 0.3}),
                                                (20, 15, 2, 4, {"dropout": 0.2}), (18, 15, 1, 5, {"dropout": 0.2, "collapse": 0.3}),
                                                (24, 10, 3, 6, {"rooted": True})])
def test_depth_clamp_arithmetic(n, m, clamp, seed, kw):
    """The depth clamp of the bit-sliced count classes (qs_count.hip clamp_fix_kernel), in numpy: the four-point test on LCA
    depths CUT at `clamp` answers like the uncut one or not at all (never a wrong topology), and adding one count of the true
    topology for every quartet with three leaves in one run of adjacent depths >= clamp (emulate.clamp_corrections = the
    kernel's enumeration) restores the oracle's table -- binary, multifurcating, partial and rooted evaluation trees."""
    import copy
    from helpers import rank4
    ref_nw = synth.reference_tree(n, seed)
    trees = synth.tree_set(n, m, 100 + seed, **kw)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    assert int(batch.adj_depth.max()) > clamp
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    want = o.counts().astype(np.int64)
    cut = copy.deepcopy(batch)
    cut.adj_depth = np.minimum(cut.adj_depth, clamp).astype(cut.adj_depth.dtype)
    T = emulate.counts_from_batch(cut, n).astype(np.int64)
    assert ((T <= want).all()) and (T != want).any()          # the cut only loses counts
    for t in range(batch.n_trees):
        lo, hi = int(batch.leaf_off[t]), int(batch.leaf_off[t + 1])
        for quad, slot in emulate.clamp_corrections(batch.leaf_ids[lo:hi], batch.adj_depth[lo:hi], clamp):
            T[int(rank4(*quad)), slot] += 1
    assert (T == want).all()


def test_depth_clamp_plan_is_host_only_and_matches_the_emulation():
    """qs_depth_clamp_plan (the class plan qs_batch_upload applies) runs without a GPU: own depth bits, the class the budget
    allows, and a correction count equal to the emulation's run enumeration; budget 0 = no clamp; a ladder is never cut at the
    default budget, and never where its run exceeds 128 leaves (the correction kernel's LDS table)."""
    import ctypes as C
    from math import comb
    from quartetscores_amd import _lib
    L = _lib.load()
    n = 200
    ref = flatten.flatten_reference(synth.reference_tree(n, 70))
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    trees = synth.tree_set(n, 60, 71) + [cat + ";"]
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)
    m = batch.n_trees

    def plan(ppm):
        hb = _lib.TreeBatchC(m, batch.leaf_off.ctypes.data, batch.leaf_ids.ctypes.data, batch.adj_depth.ctypes.data, None, None, None)
        own, cls, corr = np.zeros(m, np.uint8), np.zeros(m, np.uint8), np.zeros(m, np.uint64)
        assert L.qs_depth_clamp_plan(n, C.byref(hb), ppm, own.ctypes.data, cls.ctypes.data, corr.ctypes.data) == 0
        return own, cls, corr

    own, cls, corr = plan(0)
    assert (own == cls).all() and not corr.any() and own[-1] == 8 and own[:-1].max() >= 5
    own2, cls2, corr2 = plan(1000000)
    # ladder of 198 levels: cut at 127 its run is 73 leaves (allowed), at 63 it is 137 > 128: never, whatever the budget
    assert (own2 == own).all() and (cls2[:-1] == 4).all() and cls2[-1] == 7
    for t in range(m):
        lo, hi = int(batch.leaf_off[t]), int(batch.leaf_off[t + 1])
        runs = emulate.clamp_runs(batch.adj_depth[lo:hi], hi - lo, (1 << int(cls2[t])) - 1) if cls2[t] < own2[t] else []
        assert sum(comb(s, 3) * (hi - lo - s) + comb(s, 4) for _, s in runs) == corr2[t]
    own3, cls3, corr3 = plan(20)
    assert cls3[-1] == own3[-1] and (cls3 <= own3).all() and (corr3 <= 20e-6 * comb(n, 4) * (own3 - cls3)).all()


def test_class_plan_rules_are_host_only():
    """qs_class_plan = the classes qs_batch_upload forms, without a GPU: (1) in a batch whose classes are all below the 1024-tree floor a
    few deep trees go DOWN to the larger 4-bit class (395 tied quartets each) instead of dragging it up to 5 bits -- which is what
    happens with the clamp off; (2) the same next to a class that is large enough; (3) ladders, whose cut subtree is most of the tree,
    keep their own class; (4) a handful of full binary trees join the mode of 1100 incomplete ones; the slots are a permutation that
    lists the classes one after the other."""
    import ctypes as C
    from quartetscores_amd import _lib
    L = _lib.load()
    n = 44
    ref = flatten.flatten_reference(synth.reference_tree(n, 4500))
    rng = np.random.default_rng(45)

    def deep_tree():
        order = [int(x) for x in rng.permutation(n)]
        sub = f"((t{order[0]},t{order[1]}),t{order[2]})"
        for x in order[3:19]:
            sub = f"({sub},t{x})"
        rest = synth._to_newick(synth._join_random([f"t{x}" for x in order[19:]], rng, stop_at=2))
        return f"({sub},{rest[1:-1]});"

    def ladder():
        cat = "(t0,t1)"
        for i in range(2, n):
            cat = "(" + cat + f",t{i})"
        return cat + ";"

    def plan(batch, class_min=1024, pct=10, ppm=20):
        m = batch.n_trees
        hb = _lib.TreeBatchC(m, batch.leaf_off.ctypes.data, batch.leaf_ids.ctypes.data, batch.adj_depth.ctypes.data, None, None, None)
        mode, bits, slot = np.zeros(m, np.uint8), np.zeros(m, np.uint8), np.zeros(m, np.uint32)
        assert L.qs_class_plan(n, C.byref(hb), class_min, pct, ppm, mode.ctypes.data, bits.ctypes.data, slot.ctypes.data) == 0
        assert sorted(slot) == list(range(m))
        key = [(int(mode[t]), int(bits[t])) for t in np.argsort(slot)]
        assert all(key[i] == key[i + 1] or key[i] not in key[i + 1:] for i in range(m - 1))   # classes are contiguous in slot order
        return mode, bits

    def cat(*parts):
        from test_host_cpu_helpers import concat_batches
        out = parts[0]
        for p_ in parts[1:]:
            out = concat_batches(out, p_)
        return out

    shallow = lambda m, seed, **kw: flatten.flatten_eval_trees(synth.tree_set(n, m, seed, **kw), ref.name_to_id)
    deep = flatten.flatten_eval_trees([deep_tree() for _ in range(3)], ref.name_to_id, recentre=False)
    lad = flatten.flatten_eval_trees([ladder()] * 2, ref.name_to_id, recentre=False)
    assert int(deep.adj_depth.max()) >= 16 and int(lad.adj_depth.max()) >= 32
    # (1) all classes small: the three deep trees go down; with the clamp off the 50 go up
    mode, bits = plan(cat(shallow(50, 4501), deep))
    assert (bits == 4).all() and (mode == 0).all()
    mode, bits = plan(cat(shallow(50, 4501), deep), ppm=0)
    assert (bits == 5).all()
    # (2) beside a class that is large enough
    mode, bits = plan(cat(shallow(1100, 4502), deep))
    assert (bits == 4).all()
    # (3) ladders keep their class
    mode, bits = plan(cat(shallow(1100, 4502), lad))
    assert (bits[:1100] == 4).all() and (bits[1100:] == 6).all()
    # (4) 30 full binary trees join the mode of 1100 binary trees with missing taxa
    mode, bits = plan(cat(shallow(30, 4503), shallow(1100, 4504, dropout=0.2)))
    assert (mode == 3).all() and (bits == 4).all()
    # (5) QS_CLASS_PLAN_FUSED (QS_TUNE_FUSE_CLASSES = 1, the default since round 6): classes of equal depth bits share a launch, so no tree
    # joins a dearer mode any more -- the 30 full trees keep binary_full next to the 1100 incomplete ones -- and a deep tree of one mode
    # goes down to the depth bits at which OTHER modes have enough trees (the group is asked, not the class)
    fused = pct_fused = 10 | _lib.QS_CLASS_PLAN_FUSED
    mode, bits = plan(cat(shallow(30, 4503), shallow(1100, 4504, dropout=0.2)), pct=pct_fused)
    assert (mode[:30] == 0).all() and (mode[30:] == 3).all() and (bits == 4).all()
    mode, bits = plan(cat(shallow(1100, 4504, dropout=0.2), deep), pct=fused)
    assert (mode[:1100] == 3).all() and (mode[1100:] == 0).all() and (bits == 4).all()
    # four modes, all small: every tree keeps its mode, one depth-bits group
    mode, bits = plan(cat(shallow(40, 4505), shallow(40, 4506, dropout=0.2), shallow(40, 4507, collapse=0.2), shallow(40, 4508, collapse=0.2, dropout=0.1), deep), pct=fused)
    assert sorted(set(mode.tolist())) == [0, 1, 2, 3] and len(set(bits.tolist())) == 1
