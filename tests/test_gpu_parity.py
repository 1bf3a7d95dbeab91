"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full size --
through size-independent properties. Bit-exact for counts; scores must be identical to the
oracle's (same libm on the host), tolerance stated as <= 1 ulp.
Run on the GPU box: python -m pytest tests -m gpu
"""
import os

import numpy as np
import pytest

from helpers import d5_trees, key_of, ulp_diff
from oracle_api import Oracle
from quartetscores_amd import _lib, flatten, ranks, synth

pytestmark = pytest.mark.gpu
IMPLS = {"bitslice": _lib.QS_IMPL_BITSLICE, "swar": _lib.QS_IMPL_SWAR, "auto": _lib.QS_IMPL_AUTO}

SCORE_ULP_TOL = 1  # north_star: IC scores within 1 ulp (we expect 0)


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from quartetscores_amd import engine
    return engine


def make_case(n, m, seed, **kw):
    ref_nw = synth.reference_tree(n, seed)
    trees = synth.tree_set(n, m, 1000 + seed, **kw)
    return ref_nw, trees


def oracle_counts(ref_nw, trees, **kw):
    o = Oracle(ref_nw)
    o.count("\n".join(trees), nthreads=4, **kw)
    return o


def gpu_table(eng, ref, batch, count_bits=32, algo=None, split=None):
    ctx = eng.Context(ref.n_taxa, count_bits)
    ctx.table_alloc()
    algo = eng.QS_ALGO_GATHER if algo is None else algo
    if split:
        for lo in range(0, batch.n_trees, split):
            ctx.count_trees(batch.slice(lo, min(batch.n_trees, lo + split)), algo)
    else:
        ctx.count_trees(batch, algo)
    return ctx, ctx.table_download()


CASES = [
    # n, m, dropout, collapse, rooted, seed, expected variant substring
    (8, 20, 0.0, 0.0, False, 1, "binary_full"),
    (32, 200, 0.0, 0.0, False, 2, "binary_full"),     # BASELINE configs[0]
    (33, 37, 0.0, 0.0, True, 3, "binary_full"),       # rooted evaluation trees, ragged sizes
    (24, 50, 0.0, 0.4, False, 4, "general_full"),     # multifurcating evaluation trees
    (20, 60, 0.3, 0.3, False, 5, "partial"),          # taxon dropout + collapsed edges (fixture F2)
    (12, 17, 0.5, 0.0, True, 6, "partial"),
    (64, 40, 0.0, 0.0, False, 7, "binary_full"),
    (26, 45, 0.2, 0.0, False, 8, "binary_partial"),   # gene trees: binary, taxa missing (two comparisons + presence, two a-columns)
    (41, 70, 0.05, 0.0, True, 9, "binary_partial"),   # few taxa missing: the handful of full trees join the class
]


@pytest.mark.parametrize("n,m,dropout,collapse,rooted,seed,variant", CASES)
@pytest.mark.parametrize("count_bits", [32, 16])
@pytest.mark.parametrize("impl", ["bitslice", "bitslice_bigpanel", "swar"])
def test_gather_counts_bit_exact(eng, monkeypatch, n, m, dropout, collapse, rooted, seed, variant, count_bits, impl):
    """Both gather implementations: bit-sliced (default: count_bitslice3_kernel; "bitslice_bigpanel" forces the
    panel builder meant for n > ~256) and the byte-SWAR one (fallback for deep trees)."""
    if impl == "bitslice_bigpanel":
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_KERNEL, 1)
        impl = "bitslice"
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, IMPLS[impl])
    ref_nw, trees = make_case(n, m, seed, dropout=dropout, collapse=collapse, rooted=rooted)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx, T = gpu_table(eng, ref, batch, count_bits)
    # (the byte-SWAR kernel has no binary_partial instance: it counts such batches in its partial mode)
    assert (variant if impl == "bitslice" else variant.replace("binary_partial", "/partial/")) in ctx.last_count_variant(), ctx.last_count_variant()
    assert ("bitslice" in ctx.last_count_variant()) == (impl == "bitslice")
    o = oracle_counts(ref_nw, trees)
    assert o.names == ref.names
    assert (T.astype(np.uint64) == o.counts()).all()
    assert ctx.trees_counted == m


@pytest.mark.parametrize("n,m,dropout,collapse,rooted,seed,variant", CASES[:6] + CASES[7:8])
def test_scatter_counts_bit_exact(eng, n, m, dropout, collapse, rooted, seed, variant):
    ref_nw, trees = make_case(n, m, seed, dropout=dropout, collapse=collapse, rooted=rooted)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    for bits in (32, 16):
        ctx, T = gpu_table(eng, ref, batch, bits, algo=eng.QS_ALGO_SCATTER)
        assert "scatter" in ctx.last_count_variant()
        o = oracle_counts(ref_nw, trees)
        assert (T.astype(np.uint64) == o.counts()).all()


@pytest.mark.parametrize("n", [4, 5, 6, 7, 9, 15, 17])
@pytest.mark.parametrize("impl", ["bitslice", "swar", "scatter"])
def test_tiny_and_odd_taxon_counts(eng, monkeypatch, n, impl):
    """Edge geometry: fewer taxa than one tile, n not a multiple of the tile or d-block size, m = 1."""
    algo = eng.QS_ALGO_GATHER
    if impl == "scatter":
        algo = eng.QS_ALGO_SCATTER
    elif impl == "swar":
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    for m, kw in ((1, {}), (33, {}), (21, dict(collapse=0.3)), (19, dict(dropout=0.3))):
        ref_nw, trees = make_case(n, m, 60 + n, **kw)
        ref = flatten.flatten_reference(ref_nw)
        batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
        _, T = gpu_table(eng, ref, batch, 32, algo=algo)
        assert (T.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all(), (n, m, kw)


def test_degenerate_batches(eng):
    """Empty batch, trees with fewer than four leaves, a tree that is a single leaf."""
    ref_nw = synth.reference_tree(10, 77)
    ref = flatten.flatten_reference(ref_nw)
    ctx = eng.Context(10, 32)
    ctx.table_alloc()
    ctx.count_trees(flatten.flatten_eval_trees([], ref.name_to_id))
    assert ctx.trees_counted == 0
    trees = ["(t0,t1,t2);", "(t3,t4);", "t5;", "((t0,t1),(t2,t3),t4);"] + synth.tree_set(10, 3, 78)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx.count_trees(batch)
    T = ctx.table_download()
    o = oracle_counts(ref_nw, [t for t in trees if t.count(",") >= 1])  # the oracle's parser needs >= 2 leaves
    assert (T.astype(np.uint64) == o.counts()).all()
    assert ctx.trees_counted == len(trees)
    # overwrite mode with an empty batch still discards the previous contents
    ctx.count_trees(flatten.flatten_eval_trees([], ref.name_to_id), eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    assert ctx.trees_counted == 0 and not ctx.table_download().any()


def test_binary_batches_on_table_shards(eng):
    """The two-a-column kernel with d_lo > 0 / d_hi < n and 16-bit cells."""
    n, m = 45, 40
    ref_nw, trees = make_case(n, m, 71)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    full = oracle_counts(ref_nw, trees).counts()
    for d_lo, d_hi in ((0, 17), (17, 31), (31, 45), (3, 4), (44, 45)):
        ctx = eng.Context(n, 16, d_lo=d_lo, d_hi=d_hi)
        ctx.table_alloc()
        ctx.count_trees(batch)
        assert "x2" in ctx.last_count_variant()
        T = ctx.table_download()
        r0, r1 = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)
        assert T.shape[0] == r1 - r0 and (T.astype(np.uint64) == full[r0:r1]).all(), (d_lo, d_hi)


def test_two_column_kernel_tile_set(eng):
    """count_bitslice3_kernel enumerates every quartet exactly once at small n, odd sizes and on table shards
    (d-blocks are counted down from d_hi, the partial block sits at the bottom)."""
    for n, m, shard in ((4, 3, None), (9, 10, None), (17, 33, None), (40, 20, None), (64, 40, (20, 50)), (45, 12, (44, 45)),
                        (70, 9, (0, 11)), (70, 9, (5, 14)), (33, 70, (3, 33))):
        ref_nw, trees = make_case(n, m, 90 + n)
        ref = flatten.flatten_reference(ref_nw)
        batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
        full = oracle_counts(ref_nw, trees).counts()
        d_lo, d_hi = shard or (0, n)
        ctx = eng.Context(n, 32, d_lo=d_lo, d_hi=d_hi)
        ctx.table_alloc()
        ctx.count_trees(batch)
        assert "x2" in ctx.last_count_variant()
        r0, r1 = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)
        assert (ctx.table_download().astype(np.uint64) == full[r0:r1]).all(), (n, shard)


def test_deep_trees_take_the_u16_panel(eng, monkeypatch):
    n = 96
    ref_nw = synth.reference_tree(n, 9)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    trees = [cat + ";"] * 3 + synth.tree_set(n, 5, 10)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)  # depth up to n-2 > 63
    assert int(batch.adj_depth.max()) > 63
    o = oracle_counts(ref_nw, trees)
    ctx, T = gpu_table(eng, ref, batch)
    assert "bitslice_b7" in ctx.last_count_variant()  # 7 depth bits still fit the bit-sliced kernel
    assert (T.astype(np.uint64) == o.counts()).all()
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx, T = gpu_table(eng, ref, batch)
    assert "depth_u16" in ctx.last_count_variant()
    assert (T.astype(np.uint64) == o.counts()).all()
    monkeypatch.delitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL)
    # partial + deep
    trees2 = [cat + ";"] * 2 + synth.tree_set(n, 6, 11, dropout=0.2)
    batch2 = flatten.flatten_eval_trees(trees2, ref.name_to_id, recentre=False)
    ctx2, T2 = gpu_table(eng, ref, batch2)
    v2 = ctx2.last_count_variant()   # partial batches take the bit-sliced kernel up to 10 depth bits too (round 3)
    # (round 6: the two ladders keep the binary_full step next to the incomplete trees' binary_partial one, both at 7 bits in one fused launch)
    assert ("partial/" in v2 or "binary_partial.bitslice_b7" in v2) and "bitslice_b7" in v2 and "depth_u" not in v2, v2
    assert (T2.astype(np.uint64) == oracle_counts(ref_nw, trees2).counts()).all()


def _concat_batches(a, b):
    return flatten.TreeBatch(
        a.n_trees + b.n_trees,
        np.concatenate([a.leaf_off, b.leaf_off[1:] + a.leaf_off[-1]]).astype(np.uint32),
        np.concatenate([a.leaf_ids, b.leaf_ids]), np.concatenate([a.adj_depth, b.adj_depth]),
        np.concatenate([a.node_off, b.node_off[1:] + a.node_off[-1]]).astype(np.uint32),
        np.concatenate([a.rng_off, b.rng_off[1:] + a.rng_off[-1]]).astype(np.uint32),
        np.concatenate([a.ranges, b.ranges]))


@pytest.mark.parametrize("kind", ["binary_full", "general_full", "partial"])
def test_depth_classes_are_counted_separately(eng, kind):
    """Trees are counted class by class (bits of the deepest LCA): 1100 shallow trees run the B = 4 instance, the 40
    deep ones (interleaved in the batch) the B = 6 instance; the table equals the oracle's."""
    n = 40
    ref_nw = synth.reference_tree(n, 410)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    kw = {"binary_full": {}, "general_full": {"collapse": 0.2}, "partial": {"dropout": 0.1}}[kind]
    shallow = synth.tree_set(n, 1100, 411, **kw)
    parts, trees = [], []
    for k in range(4):   # deep, shallow, deep, shallow, ... in the batch
        parts.append(flatten.flatten_eval_trees([cat + ";"] * 10, ref.name_to_id, recentre=False))
        parts.append(flatten.flatten_eval_trees(shallow[k * 275:(k + 1) * 275], ref.name_to_id))
        trees += [cat + ";"] * 10 + shallow[k * 275:(k + 1) * 275]
    batch = parts[0]
    for p_ in parts[1:]:
        batch = _concat_batches(batch, p_)
    assert batch.n_trees == 1140
    for algo_split in (None, 500):
        ctx, T = gpu_table(eng, ref, batch, 32, split=algo_split)
        if algo_split is None:
            v = ctx.last_count_variant()
            # (round 6: the ladders keep the binary_full step whatever the shallow trees need -- no tree joins a dearer mode -- and
            # the handful of complete trees among the incomplete ones does too)
            assert "bitslice_b4" in v and "bitslice_b6x2:40" in v and v.count(":40") == 1 and (kind != "binary_full" or ":1100+" in v), v
        assert (T.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all(), (kind, algo_split)


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("count_bits", [32, 16])
def test_mixed_batches_are_counted_mode_by_mode(eng, monkeypatch, count_bits, fuse):
    """The kernel mode is a property of the TREE, not of the batch: a batch of one third full binary trees,
    one third binary trees with missing taxa and one third trees with collapsed edges (some of those with missing taxa too),
    interleaved, with two deep ladders among them, is counted class by class -- binary_full, binary_partial, general_full,
    partial, each with its own depth classes -- and the table equals the oracle's; with the default class floor (1024
    trees) the small classes join the most general mode present and the table is the same. The reference's loop is
    shape-independent (QuartetCounterLookup.hpp:65-106, partial trees :214-221).
    fuse = 1 (QS_TUNE_FUSE_CLASSES, the default since round 6): the four classes that share 4 depth bits run as segments of ONE
    launch of count_bitslice3_fused_kernel (the two ladders keep a launch of their own at 5 bits), and with the default floors no
    mode joins a dearer one any more -- the small 4-bit classes go up to the ladders' 5 bits instead and everything is one launch."""
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_FUSE_CLASSES, fuse)
    n = 30
    ref_nw = synth.reference_tree(n, 430)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    full, drop = synth.tree_set(n, 40, 431), synth.tree_set(n, 40, 432, dropout=0.12)
    coll, both = synth.tree_set(n, 30, 433, collapse=0.2), synth.tree_set(n, 12, 434, collapse=0.2, dropout=0.1)
    trees, parts = [], []
    for k in range(10):
        chunk = full[4 * k:4 * k + 4] + drop[4 * k:4 * k + 4] + coll[3 * k:3 * k + 3] + both[k:k + 1]
        trees += chunk
        parts.append(flatten.flatten_eval_trees(chunk, ref.name_to_id))
    trees += [cat + ";"] * 2
    parts.append(flatten.flatten_eval_trees([cat + ";"] * 2, ref.name_to_id, recentre=False))   # depth 28: the 5-bit class of binary_full
    batch = parts[0]
    for p_ in parts[1:]:
        batch = _concat_batches(batch, p_)
    want = oracle_counts(ref_nw, trees).counts()
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    ctx, T = gpu_table(eng, ref, batch, count_bits)
    v = ctx.last_count_variant()
    assert "gather/mixed/" in v, v
    for piece in ("binary_full.bitslice_b4x2:", "binary_full.bitslice_b5x2:2", "binary_partial.bitslice_b4x2:", "general_full.bitslice_b4x2:", "partial.bitslice_b4x2:"):
        assert piece in v, (piece, v)
    assert ("/fused:1" in v) == bool(fuse), v
    assert (T.astype(np.uint64) == want).all(), v
    # accumulation over two uploads of the same mixed batch, split at an odd place
    ctx2, T2 = gpu_table(eng, ref, batch, count_bits, split=37)
    assert (T2.astype(np.uint64) == want).all()
    # default floors: the small classes join a more general mode that is present -- same table
    monkeypatch.delitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES)
    monkeypatch.delitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT)
    ctx3, T3 = gpu_table(eng, ref, batch, count_bits)
    v3 = ctx3.last_count_variant()
    if fuse:   # every tree keeps its own mode; one depth-bits group, one launch
        assert "gather/mixed/" in v3 and "/fused:1" in v3 and v3.count("bitslice_b") == 4 and "binary_full.bitslice_b" in v3, v3
    else:
        assert "gather/partial/" in v3, v3
    assert (T3 == T).all()
    # the byte-SWAR implementation takes the batch as a whole in the mode it needs
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx4, T4 = gpu_table(eng, ref, batch, count_bits)
    assert "gather/partial/depth_u" in ctx4.last_count_variant()
    assert (T4 == T).all()


def test_a_few_incomplete_trees_do_not_slow_the_full_ones(eng, monkeypatch):
    """1200 full binary trees and 1100 binary trees with missing taxa in one batch (interleaved): two classes, the full
    trees keep the binary_full instance; tuples still equal the oracle's."""
    n = 20
    ref_nw = synth.reference_tree(n, 440)
    ref = flatten.flatten_reference(ref_nw)
    full, drop = synth.tree_set(n, 1200, 441), synth.tree_set(n, 1100, 442, dropout=0.2)
    trees = []
    for k in range(100):
        trees += full[12 * k:12 * k + 12] + drop[11 * k:11 * k + 11]
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx, T = gpu_table(eng, ref, batch, 32)
    v = ctx.last_count_variant()
    assert "gather/mixed/" in v and "binary_full.bitslice_b4x2:12" in v and "binary_partial.bitslice_b4x2:10" in v, v
    assert (T.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all()


def test_only_the_deep_trees_take_the_deep_instance(eng, monkeypatch):
    n = 140
    ref_nw = synth.reference_tree(n, 420)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    deep = flatten.flatten_eval_trees([cat + ";"] * 2, ref.name_to_id, recentre=False)
    batch = _concat_batches(flatten.flatten_eval_trees(synth.tree_set(n, 70, 421), ref.name_to_id), deep)
    assert int(batch.adj_depth.max()) > 127
    ctx, T = gpu_table(eng, ref, batch)
    v = ctx.last_count_variant()   # depth 138 < 256: the 8-bit instance; the 70 shallow trees (< 1024: no class of their own) join it
    assert "bitslice_b8x2" in v and "depth_u" not in v, v
    assert (T.sum(axis=1) == 72).all()
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx2, T2 = gpu_table(eng, ref, batch)
    assert "depth_u16" in ctx2.last_count_variant() and "bitslice" not in ctx2.last_count_variant()
    assert (T == T2).all()


@pytest.mark.parametrize("n,bits", [(20, 5), (40, 6), (80, 7), (150, 8)])
@pytest.mark.parametrize("kind", ["binary_full", "general_full", "partial", "binary_partial"])
@pytest.mark.parametrize("count_bits", [32, 16])
def test_every_depth_width_of_the_bitsliced_kernel(eng, monkeypatch, n, bits, kind, count_bits):
    """All (depth bits B, mode) instances of count_bitslice3_kernel up to 8 bits against the oracle: a caterpillar that is
    not re-rooted forces depth n-2; 40 trees span two 32-tree groups; both panel builders (9 and 10 bits:
    test_deep_ladders_9_and_10_bits, on table shards)."""
    ref_nw = synth.reference_tree(n, 200 + n)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    kw = {"binary_full": {}, "general_full": {"collapse": 0.25}, "partial": {"dropout": 0.15, "collapse": 0.1}, "binary_partial": {"dropout": 0.15}}[kind]
    trees = [cat + ";"] * 2 + synth.tree_set(n, 38, 300 + n, **kw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)
    assert (1 << (bits - 1)) <= int(batch.adj_depth.max()) < (1 << bits)
    want = oracle_counts(ref_nw, trees).counts()
    for builder in ("small", "big"):
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_KERNEL, 1 if builder == "big" else 0)
        ctx, T = gpu_table(eng, ref, batch, count_bits)
        v = ctx.last_count_variant()
        assert kind in v and f"bitslice_b{bits}" in v and "depth_u" not in v, v
        assert (T.astype(np.uint64) == want).all(), (builder, v)


# (132, 34, 8): a ladder of more than 128 LCA levels -- the cooperative kernel carries at most 7 planes, so classes of 8-10 depth
# bits must take count_bitslice3_kernel over the full launch order even with QS_TUNE_COOP = 1
@pytest.mark.parametrize("n,m,bits", [(40, 60, 4), (33, 70, 5), (58, 100, 6), (90, 40, 7), (132, 34, 8)])
@pytest.mark.parametrize("count_bits", [32, 16])
def test_cooperative_count_kernel_matches_oracle(eng, monkeypatch, n, m, bits, count_bits):
    """count_bitslice4_kernel (binary full batches; the four waves of a workgroup = four consecutive third ids of one
    (a,b,d) tile share their M[ab] / M[bd] loads through LDS, shadow tiles pad incomplete groups) forced on at small
    sizes: whole tables equal to the oracle's for every depth width, both cell widths, a table shard, accumulation over
    two batches, the two-cell wire output -- and equal to what count_bitslice3_kernel alone produces."""
    ref_nw = synth.reference_tree(n, 700 + n)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"        # a caterpillar that is not re-rooted has depth n - 2: forces the instance with `bits` depth bits
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    trees = ([cat + ";"] * 2 if bits > 4 else []) + synth.tree_set(n, m - (2 if bits > 4 else 0), 800 + n)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=(bits == 4))   # re-centred random trees: 4 bits
    assert ((1 << (bits - 1)) <= int(batch.adj_depth.max()) or bits == 4) and int(batch.adj_depth.max()) < (1 << bits)
    want = oracle_counts(ref_nw, trees).counts()
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_COOP, 2)
    _, T_plain = gpu_table(eng, ref, batch, count_bits)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_COOP, 1)
    ctx, T = gpu_table(eng, ref, batch, count_bits)
    v = ctx.last_count_variant()
    # (the cooperative kernel carries at most 7 planes: an 8-bit class runs count_bitslice3_kernel alone and the variant says so)
    assert "binary_full" in v and ("coop" in v) == (bits <= 7) and f"bitslice_b{bits}" in v, v
    assert (T.astype(np.uint64) == want).all(), v
    assert (T == T_plain).all()
    # two batches accumulate (the second launch reads-modifies-writes)
    ctx2 = eng.Context(ref.n_taxa, count_bits)
    ctx2.table_alloc()
    half = m // 2
    ctx2.count_trees(batch.slice(0, half), eng.QS_ALGO_GATHER)
    ctx2.count_trees(batch.slice(half, m), eng.QS_ALGO_GATHER)
    ctx2.sync()
    assert (ctx2.table_download().astype(np.uint64) == want).all()
    # a table shard: d in [d_lo, n)
    if n >= 30:
        d_lo = n - 9
        ctx3 = eng.Context(ref.n_taxa, count_bits, d_lo=d_lo, d_hi=n)
        ctx3.table_alloc()
        ctx3.count_trees(batch, eng.QS_ALGO_GATHER)
        ctx3.sync()
        lo = ranks.n_quartets(d_lo)
        assert (ctx3.table_download().astype(np.uint64) == want[lo:]).all()


def test_very_deep_trees_stay_bit_sliced(eng):
    n = 150
    ref_nw = synth.reference_tree(n, 19)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    trees = [cat + ";"] * 2 + synth.tree_set(n, 3, 20)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)
    assert int(batch.adj_depth.max()) > 127
    ctx, T = gpu_table(eng, ref, batch)
    assert "bitslice_b8" in ctx.last_count_variant() and "depth_u" not in ctx.last_count_variant()
    assert (T.sum(axis=1) == 5).all()
    _, T2 = gpu_table(eng, ref, flatten.flatten_eval_trees(trees, ref.name_to_id))  # re-centred: shallow again
    assert (T == T2).all()


@pytest.mark.parametrize("n,bits", [(300, 9), (600, 10), (1100, 11)])
@pytest.mark.parametrize("kind", ["binary_full", "general_full", "partial", "binary_partial"])
def test_deep_ladders_9_and_10_bits(eng, monkeypatch, n, bits, kind):
    """Ladder-like trees (LCA depths up to n - 2, not re-rooted) on the 9- and 10-bit instances of the bit-sliced kernel,
    11 bits = beyond them (byte-SWAR kernel with 16-bit depths). Tables of these sizes are too large for the oracle, so a
    table SHARD (the two largest ids) is compared with the byte-SWAR kernel's shard bit for bit and with the split-based
    brute force (tests/bruteforce.py) on random quartets. The reference's loop is shape-independent
    (QuartetCounterLookup.hpp:65-106): no cliff for deep trees."""
    import sys
    import bruteforce
    old_limit = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old_limit, 20 * n))                           # the brute force parses ladders recursively
    ref_nw = synth.reference_tree(n, 900 + n)
    ref = flatten.flatten_reference(ref_nw)
    cat = "(t0,t1)"
    for i in range(2, n):
        cat = "(" + cat + f",t{i})"
    kw = {"binary_full": {}, "general_full": {"collapse": 0.25}, "partial": {"dropout": 0.1, "collapse": 0.1}, "binary_partial": {"dropout": 0.1}}[kind]
    trees = [cat + ";"] * 2 + synth.tree_set(n, 34, 950 + n, **kw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)
    assert (1 << (bits - 1)) <= int(batch.adj_depth.max()) < (1 << bits)
    d_lo = n - 2

    def shard_table():
        ctx = eng.Context(n, 16, d_lo=d_lo, d_hi=n)
        ctx.table_alloc()
        ctx.count_trees(batch, eng.QS_ALGO_GATHER)
        ctx.sync()
        return ctx, ctx.table_download()
    ctx, T = shard_table()
    v = ctx.last_count_variant()
    # (beyond 10 bits the two ladders take the byte-SWAR kernel -- in its partial mode when they share a mode with incomplete trees)
    assert kind in v and ((f"bitslice_b{bits}" in v) if bits <= 10 else ("depth_u16" in v)), v
    rng = np.random.default_rng(n)
    qs_ = np.sort(np.stack([np.append(rng.choice(d_lo + 1, size=3, replace=False), rng.integers(d_lo, n)) for _ in range(1500)]), axis=1)
    qs_ = qs_[(qs_[:, 2] < qs_[:, 3])].astype(np.uint16)
    want = bruteforce.quartet_counts_for(trees, ref.names, qs_.astype(np.int64))
    assert (ctx.lookup(qs_) == want).all()
    if bits <= 10:
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
        ctx2, T2 = shard_table()
        assert "depth_u16" in ctx2.last_count_variant()
        assert (T == T2).all()
    sys.setrecursionlimit(old_limit)


def _ladder(n, order=None):
    order = list(range(n)) if order is None else order
    cat = f"(t{order[0]},t{order[1]})"
    for i in order[2:]:
        cat = "(" + cat + f",t{i})"
    return cat + ";"


@pytest.mark.parametrize("kind", ["binary_full", "binary_partial", "general_full", "partial"])
@pytest.mark.parametrize("count_bits", [32, 16])
def test_depth_clamp_counts_bit_exact(eng, monkeypatch, kind, count_bits):
    """Depth clamp (QS_TUNE_DEPTH_CLAMP): trees counted in a class BELOW their own depth bits -- their LCA depths cut at the class's
    largest value by the panel builders, the quartets the cut ties (three leaves below one node of that depth) added by
    clamp_fix_kernel -- give the oracle's table, in every kernel mode and both cell widths: ladders and ladder + NNI trees of
    44 taxa (up to 42 LCA levels, cut at 15: one run of 28 leaves, or two runs when recentred) among random trees that fit
    the class as they are. The reference's loop is shape-independent (QuartetCounterLookup.hpp:65-106)."""
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_DEPTH_CLAMP, 1000000)     # any price per bit
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    n = 44
    ref_nw = synth.reference_tree(n, 4400)
    ref = flatten.flatten_reference(ref_nw)
    rng = np.random.default_rng(44)
    kw = {"binary_full": {}, "general_full": {"collapse": 0.2}, "partial": {"dropout": 0.1, "collapse": 0.15}, "binary_partial": {"dropout": 0.1}}[kind]
    deep = [_ladder(n, list(rng.permutation(n))) for _ in range(3)]
    deep += synth.nni_tree_set(deep[0], 12, 4401, mean_nni=5)
    if kind in ("general_full", "partial"):       # multifurcations inside the cut subtrees as well
        deep += [t.replace("(((t", "((t", 1).replace("),", ",", 1) for t in deep[:3]]
    trees = deep + synth.tree_set(n, 20, 4402, **kw)
    for recentre in (False, True):
        batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=recentre)
        ctx = eng.Context(n, count_bits)
        ctx.table_alloc()
        hb = ctx.batch_upload(batch, with_nodes=False)
        clamped, quartets, units = ctx.batch_clamp_info(hb)
        assert clamped >= 10 and quartets > 0 and units >= clamped, (clamped, quartets, units)
        ctx.count_batch(hb, eng.QS_ALGO_GATHER)
        ctx.sync()
        v = ctx.last_count_variant()
        assert f"/clamp:{clamped}" in v and "bitslice_b4" in v and "bitslice_b5" not in v and "bitslice_b6" not in v, v
        T = ctx.table_download()
        ctx.batch_free(hb)
        o = oracle_counts(ref_nw, trees)
        assert (T.astype(np.uint64) == o.counts()).all(), (kind, count_bits, recentre)
    # the same batch without the clamp: the deep trees keep their own classes, same table
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_DEPTH_CLAMP, 0)
    ctx, T0 = gpu_table(eng, ref, batch, count_bits)
    assert "/clamp" not in ctx.last_count_variant() and (T0 == T).all()


def test_depth_clamp_budget_and_long_runs(eng, monkeypatch):
    """The clamp's decisions: the default budget (20 millionths of C(n,4) per bit saved) leaves a ladder alone (its cut subtree
    is most of the tree), a run of more than 128 leaves is never cut whatever the budget, accumulation over two uploads and
    several panel slices per class keep the corrections with their slice, and a table shard takes only its own quartets."""
    import ctypes as C
    n = 150
    ref_nw = synth.reference_tree(n, 9000)
    ref = flatten.flatten_reference(ref_nw)
    trees = [_ladder(n)] * 2 + synth.tree_set(n, 30, 9001)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)     # ladder: 148 levels (8 bits)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER)
    ctx.sync()
    assert "bitslice_b8" in ctx.last_count_variant()                             # default budget: the ladders keep their 8-bit class
    ctx.batch_free(hb)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_DEPTH_CLAMP, 1000000)
    # the host-only plan says what to expect: cut at 31 (5 bits) the ladder's run is 119 leaves, at 15 it is 135 > 128: never
    L = _lib.load()
    own, cls = np.zeros(batch.n_trees, np.uint8), np.zeros(batch.n_trees, np.uint8)
    hbs = _lib.TreeBatchC(batch.n_trees, batch.leaf_off.ctypes.data, batch.leaf_ids.ctypes.data, batch.adj_depth.ctypes.data, None, None, None)
    assert L.qs_depth_clamp_plan(n, C.byref(hbs), 1000000, own.ctypes.data, cls.ctypes.data, None) == 0
    assert own[0] == 8 and cls[0] == 5 and (cls[2:] == 4).all()
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER)
    ctx.sync()
    v = ctx.last_count_variant()
    assert "bitslice_b5" in v and "bitslice_b8" not in v and f"/clamp:{int((cls < own).sum())}" in v, v
    o = oracle_counts(ref_nw, trees)
    assert (ctx.table_download().astype(np.uint64) == o.counts()).all()
    ctx.batch_free(hb)
    n = 90
    ref_nw = synth.reference_tree(n, 9000)
    ref = flatten.flatten_reference(ref_nw)
    # recentred ladders + NNIs: accumulate two uploads, 3 panel slices per class, on two table shards
    trees2 = synth.nni_tree_set(_ladder(n), 70, 9002, mean_nni=4) + synth.tree_set(n, 40, 9003)
    batch2 = flatten.flatten_eval_trees(trees2, ref.name_to_id)
    assert int(batch2.adj_depth.max()) >= 32
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_SLICE_BYTES, int(n * (n - 1) / 2) * 16 * 2)   # two tree groups per slice
    want = oracle_counts(ref_nw, trees2).counts()
    lo_tuples = 0
    for d_lo, d_hi in ((0, 70), (70, n)):
        c2 = eng.Context(n, 16, d_lo=d_lo, d_hi=d_hi)
        c2.table_alloc()
        for lo, hi in ((0, 50), (50, len(trees2))):
            hb = c2.batch_upload(batch2.slice(lo, hi), with_nodes=False)
            assert c2.batch_clamp_info(hb)[0] > 0
            c2.count_batch(hb, eng.QS_ALGO_GATHER)
            c2.batch_free(hb)
        c2.sync()
        T = c2.table_download().astype(np.uint64)
        assert (T == want[lo_tuples:lo_tuples + len(T)]).all(), (d_lo, d_hi)
        lo_tuples += len(T)
    assert lo_tuples == len(want)


def test_a_few_deep_trees_join_the_larger_class_below_them(eng):
    """Default tuning, a batch whose classes are ALL below the 1024-tree floor: 50 trees that fit 4 depth bits and 3 whose one deep
    subtree (5 leaves below depth 15) needs 5. The old rule sent the larger class UP to the smaller one's depth bits; now the few
    deep trees go down (395 tied quartets each, far below the price of a table pass) and the batch is one 4-bit class. Table = oracle."""
    n = 44
    ref_nw = synth.reference_tree(n, 4500)
    ref = flatten.flatten_reference(ref_nw)
    rng = np.random.default_rng(45)
    deep = []
    for _ in range(3):
        order = [int(x) for x in rng.permutation(n)]
        sub = f"((t{order[0]},t{order[1]}),t{order[2]})"                        # inner nodes at depths 17 and 18
        for x in order[3:19]:                                                    # a spine of 16 single leaves above it
            sub = f"({sub},t{x})"
        rest = synth._to_newick(synth._join_random([f"t{x}" for x in order[19:]], rng, stop_at=2))
        deep.append(f"({sub},{rest[1:-1]});")
    trees = synth.tree_set(n, 50, 4501) + deep
    batch_a = flatten.flatten_eval_trees(trees[:50], ref.name_to_id)
    batch_b = flatten.flatten_eval_trees(deep, ref.name_to_id, recentre=False)
    assert int(batch_a.adj_depth.max()) <= 15 and 16 <= int(batch_b.adj_depth.max()) <= 31
    batch = _concat_batches(batch_a, batch_b)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    assert ctx.batch_clamp_info(hb)[0] == 3 and ctx.batch_clamp_info(hb)[1] == 3 * (10 * 39 + 5)   # C(5,3) (44 - 5) + C(5,4) per tree
    ctx.count_batch(hb, eng.QS_ALGO_GATHER)
    ctx.sync()
    v = ctx.last_count_variant()
    assert "bitslice_b4x2" in v and "bitslice_b5" not in v and "/clamp:3" in v and ":" not in v.split("/clamp")[0], v
    assert (ctx.table_download().astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all()
    ctx.batch_free(hb)


def test_depth_clamp_in_the_wire_format(eng, monkeypatch):
    """QS_COUNT_WIRE16X2 with clamped trees: the corrections go to the wire words (n0 | n1 << 16; n2 is implied)."""
    import torch
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_DEPTH_CLAMP, 1000000)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    n = 40
    ref_nw = synth.reference_tree(n, 4000)
    ref = flatten.flatten_reference(ref_nw)
    trees = synth.nni_tree_set(_ladder(n), 40, 4001, mean_nni=3) + synth.tree_set(n, 25, 4002)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=False)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(batch)
    assert "/clamp:" in ctx.last_count_variant()
    want = torch.zeros(ctx.table_tuples, dtype=torch.int32, device="cuda")
    ctx.table_pack16x2(want)
    ctx.sync()
    c2 = eng.Context(n, 32)
    words = torch.zeros(ctx.table_tuples, dtype=torch.int32, device="cuda")
    c2.wire_attach(words)
    hb = c2.batch_upload(batch, with_nodes=False)
    c2.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_WIRE16X2)
    c2.sync()
    assert "/clamp:" in c2.last_count_variant() and "wire_u16x2" in c2.last_count_variant()
    assert torch.equal(words, want)
    o = oracle_counts(ref_nw, trees)
    assert (ctx.table_download().astype(np.uint64) == o.counts()).all()


def test_batches_accumulate_and_are_deterministic(eng):
    ref_nw, trees = make_case(28, 90, 12)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    _, T1 = gpu_table(eng, ref, batch)
    _, T2 = gpu_table(eng, ref, batch, split=16)  # ragged last batch
    _, T3 = gpu_table(eng, ref, batch, algo=eng.QS_ALGO_SCATTER, split=32)
    assert (T1 == T2).all() and (T1 == T3).all()
    assert (T1.sum(axis=1) == 90).all()  # every tree resolves every quartet


def test_overwrite_mode_equals_clear_plus_count(eng, monkeypatch):
    ref_nw, trees = make_case(26, 70, 16, collapse=0.1)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    for impl in ("bitslice", "swar"):
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_GATHER_IMPL, IMPLS[impl])
        ctx = eng.Context(26, 32)
        ctx.table_alloc()
        ctx.table_upload(np.full((ranks.n_quartets(26), 3), 12345, dtype=np.uint32))  # stale contents
        hb = ctx.batch_upload(batch)
        ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
        ctx.sync()
        T = ctx.table_download()
        assert ctx.trees_counted == 70
        assert (T.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all()
        ctx.count_batch(hb, eng.QS_ALGO_GATHER)  # accumulates on top
        ctx.sync()
        assert (ctx.table_download() == 2 * T).all() and ctx.trees_counted == 140
        ctx.batch_free(hb)


def test_panel_slicing_gives_the_same_table(eng, monkeypatch):
    """Large batches are counted in sub-batches whose panel fits the Infinity Cache; force tiny slices."""
    ref_nw, trees = make_case(20, 150, 15, collapse=0.2, dropout=0.1)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    _, T1 = gpu_table(eng, ref, batch)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_SLICE_BYTES, 190 * 16 * 2)  # two 16-tree chunks per slice
    _, T2 = gpu_table(eng, ref, batch)
    assert (T1 == T2).all()
    assert (T1.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all()


def test_lookup_matches_oracle(eng):
    ref_nw, trees = make_case(16, 30, 13, collapse=0.3)
    ref = flatten.flatten_reference(ref_nw)
    q = eng.QuartetCounterLookup(ref, trees)
    o = oracle_counts(ref_nw, trees)
    rng = np.random.default_rng(0)
    for _ in range(200):
        ids = rng.choice(16, size=4, replace=False)
        nodes = [int(ref.leaf_node[i]) for i in ids]
        assert q.countQuartetOccurrences(*nodes) == o.lookup(*[int(i) for i in ids])


def test_unknown_taxon_is_an_error(eng, golden):
    with pytest.raises(flatten.UnknownTaxonError):
        eng.QuartetCounterLookup(golden["D1"]["ref"], [golden["D6"]["bad_tree"]])


def test_u16_table_overflow_is_reported(eng):
    ref_nw, trees = make_case(8, 4, 14)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(8, 16)
    ctx.table_alloc()
    ctx.table_upload(np.full((70, 3), 65534, dtype=np.uint16))
    with pytest.raises(eng.QSError) as ei:
        ctx.count_trees(batch)
    assert ei.value.code == -5


# ---- scores ---------------------------------------------------------------------------------

def assert_scores_equal(gpu_sc, ora_sc):
    assert set(gpu_sc) == set(ora_sc)
    worst = 0
    for k, ov in ora_sc.items():
        gv = gpu_sc[k]
        for g, o in zip(gv, ov):
            if o is None:
                assert g is None
                continue
            d = int(ulp_diff(g, o))
            worst = max(worst, d)
            assert d <= SCORE_ULP_TOL, (sorted(k), g, o)
    return worst


@pytest.mark.parametrize("n,m,dropout,collapse,seed", [(8, 20, 0, 0, 21), (32, 200, 0, 0, 22), (20, 60, 0.3, 0.3, 23),
                                                       (48, 30, 0, 0.2, 24)])
def test_scores_bifurcating_reference(eng, n, m, dropout, collapse, seed):
    ref_nw, trees = make_case(n, m, seed, dropout=dropout, collapse=collapse)
    qsc = eng.QuartetScoreComputer(ref_nw, trees)
    assert qsc.bifurcating
    o = oracle_counts(ref_nw, trees)
    o.score(nthreads=4)
    worst = assert_scores_equal(qsc.scores_by_bipartition(), o.scores_by_bipartition())
    assert worst == 0  # same libm, same triples -> identical doubles


def test_scores_golden_D1_D2_D3(eng, golden):
    g = golden["D1"]
    qsc = eng.QuartetScoreComputer(g["ref"], g["eval"])
    sc = qsc.scores_by_bipartition()
    for k, v in g["scores"].items():
        kk = key_of(k)
        got = sc.get(kk) or sc[frozenset(set(qsc.ref.names) - kk)]
        assert list(got) == v
    # D2: multifurcating reference -> LQ-IC only
    qsc2 = eng.QuartetScoreComputer(golden["D2"]["ref"], g["eval"])
    assert not qsc2.bifurcating and qsc2.getQPICScores() == [] and qsc2.getEQPICScores() == []
    sc2 = qsc2.scores_by_bipartition()
    assert len(sc2) == len(golden["D2"]["lq"])
    for k, v in golden["D2"]["lq"].items():
        kk = key_of(k)
        got = sc2.get(kk) or sc2[frozenset(set(qsc2.ref.names) - kk)]
        assert got[0] == v
    # D3: 150/50 mix; semantic counts, so no savemem-u8 overflow (reference defect Q1 not reproduced)
    d3 = golden["D3"]
    trees = [d3["eval"][0]] * 150 + [d3["eval"][1]] * 50
    qsc3 = eng.QuartetScoreComputer(d3["ref"], trees, enforceSmallMem=True)
    ids = [qsc3.ref.leaf_node[qsc3.ref.name_to_id[x]] for x in "abcd"]
    assert qsc3.quartetCounterLookup.countQuartetOccurrences(*[int(i) for i in ids]) == tuple(d3["occ_abcd"])
    sc3 = qsc3.scores_by_bipartition()
    for k, v in d3["scores"].items():
        assert list(sc3[key_of(k)]) == v


def test_scores_multifurcating_reference_random(eng):
    n = 18
    ref_nw = synth.random_tree(n, np.random.default_rng(31), collapse=0.4)
    trees = synth.tree_set(n, 40, 32, collapse=0.2)
    qsc = eng.QuartetScoreComputer(ref_nw, trees)
    o = oracle_counts(ref_nw, trees)
    o.score()
    assert qsc.bifurcating == o.bifurcating
    assert_scores_equal(qsc.scores_by_bipartition(), o.scores_by_bipartition())


def test_D5_u32_wrap_of_qp_sums(eng, golden):
    """70 000 trees (two topologies) on 64 taxa: QP sums exceed 2^32."""
    g = golden["D5"]
    ref_nw, alt_nw = d5_trees(g["n"], g["block"])
    ref = flatten.flatten_reference(ref_nw)
    two = flatten.flatten_eval_trees([ref_nw, alt_nw], ref.name_to_id)
    L = g["n"]

    def tile(b, t, k):
        s = b.slice(t, t + 1)
        return s.leaf_ids, s.adj_depth, k
    ids = np.concatenate([np.tile(two.slice(t, t + 1).leaf_ids, k) for t, k in ((0, g["mult"][0]), (1, g["mult"][1]))])
    dep = np.concatenate([np.tile(two.slice(t, t + 1).adj_depth, k) for t, k in ((0, g["mult"][0]), (1, g["mult"][1]))])
    m = sum(g["mult"])
    big = flatten.TreeBatch(m, (np.arange(m + 1, dtype=np.uint32) * L), ids, dep, np.zeros(m + 1, dtype=np.uint32),
                            np.zeros(1, dtype=np.uint32), np.zeros(0, dtype=np.uint16))
    del tile
    ctx = eng.Context(g["n"], 32)
    ctx.table_alloc()
    ctx.count_trees(big)
    T = ctx.table_download()
    assert (T.sum(axis=1) == m).all()
    central = frozenset(f"t{i}" for i in range(32, 64))
    for flag, want in ((eng.QS_SCORE_QP_WRAP32, g["qp_wrap32"]), (eng.QS_SCORE_QP_EXACT64, g["qp_exact64"])):
        lq, qp, eqp, bif = ctx.score(ref, flag)
        assert bif
        hit = 0
        for v in range(1, ref.n_nodes):
            below = frozenset(x.name for x in __import__("quartetscores_amd").newick.preorder(ref.nodes[v]) if x.is_leaf)
            if below == central or frozenset(ref.names) - below == central:
                assert lq[v] == g["lq"] and qp[v] == want and eqp[v] == want
                hit += 1
            elif 1 < len(below) < g["n"] - 1:
                assert lq[v] == 1.0 and qp[v] == 1.0 and eqp[v] == 1.0
        assert hit == 1


@pytest.mark.parametrize("logged", [False, True])
@pytest.mark.parametrize("count_bits", [32, 16])
def test_table_sharded_counting_and_scoring(eng, count_bits, logged):
    """BASELINE configs[4] in miniature: the table is split by the largest taxon id into shards
    (here 3 contexts on one GPU stand in for 3 GPUs); every shard sees all trees; the per-node-pair
    accumulators are combined exactly as distributed.score_sharded does with collectives.
    logged: every shard's pass 1 keeps its candidate log (forced: the shards are far below 1 GB) and its pass 2 filters
    that log against the minima over ALL shards instead of reading the shard again -- same scores."""
    import torch
    from quartetscores_amd import distributed
    n, m, G = 40, 50, 3
    ref_nw, trees = make_case(n, m, 51, collapse=0.15, dropout=0.1)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    o = oracle_counts(ref_nw, trees)
    full = o.counts()
    ctxs, sums, mins = [], None, None
    for g in range(G):
        d_lo, d_hi = distributed.shard_of_largest_id(n, G, g)
        ctx = eng.Context(n, count_bits, d_lo=d_lo, d_hi=d_hi)
        ctx.table_alloc()
        ctx.count_trees(batch)
        T = ctx.table_download()
        ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, 2 if logged else 1)
        r0, r1 = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)
        assert T.shape[0] == r1 - r0 and (T.astype(np.uint64) == full[r0:r1]).all()
        # lookups outside the shard read as zero, inside as the oracle
        P = ctx.score_pair_slots(ref)
        s_ = torch.empty(3 * P, dtype=torch.int64, device="cuda"); m_ = torch.empty(P, dtype=torch.int64, device="cuda")
        ctx.score_pass1(ref, s_, m_)
        sums = s_ if sums is None else sums + s_            # all_reduce(SUM)
        mins = m_ if mins is None else torch.minimum(mins, m_)  # all_reduce(MIN)
        ctxs.append(ctx)
    cands = []
    for ctx in ctxs:
        c_ = torch.empty(8 * ctx.score_pair_slots(ref), dtype=torch.int64, device="cuda")
        ctx.score_pass2(ref, mins, c_)
        assert (ctx.last_score_log() > 0) == logged
        cands.append(c_.cpu().numpy())                      # all_gather
        if logged:      # a second pass 2 without a new pass 1 reads the table (the log is spent): same candidates
            c2 = torch.empty_like(c_)
            ctx.score_pass2(ref, mins, c2)
            assert ctx.last_score_log() == 0
            assert np.array_equal(np.sort(c2.cpu().numpy().reshape(-1, 8), axis=1), np.sort(cands[-1].reshape(-1, 8), axis=1))
    lq, qp, eqp, bif = ctxs[0].score_finish(ref, sums.cpu().numpy(), np.stack(cands))
    # the same numbers as the unsharded path and as the oracle
    whole = eng.Context(n, 32)
    whole.table_alloc()
    whole.count_trees(batch)
    lq2, qp2, eqp2, bif2 = whole.score(ref)
    assert bif == bif2 and (lq == lq2).all() and (qp == qp2).all() and (eqp == eqp2).all()
    lq3, qp3, eqp3, _ = distributed.score_sharded(whole, ref)  # single-rank path of the same helper
    assert (lq == lq3).all() and (qp == qp3).all() and (eqp == eqp3).all()
    o.score()
    osc = o.scores_by_bipartition()
    from quartetscores_amd import newick
    got = {}
    for v in range(1, ref.n_nodes):
        below = frozenset(x.name for x in newick.preorder(ref.nodes[v]) if x.is_leaf)
        if 1 < len(below) < n - 1:
            other = frozenset(ref.names) - below
            key = below if (len(below) < len(other) or (len(below) == len(other) and min(ref.names) not in below)) else other
            got[key] = (lq[v], qp[v], eqp[v])
    assert assert_scores_equal(got, osc) == 0


def test_rooted_reference_matches_the_reference_quirk_D4(eng, golden):
    """Default handling of a degree-2 root = the reference's (quirk Q5, QuartetScoreComputer.hpp:393-396): Appendix D4."""
    g4, g1 = golden["D4"], golden["D1"]
    qsc = eng.QuartetScoreComputer(g4["ref"], g1["eval"])
    found = {}
    names = frozenset(qsc.ref.names)
    for e in range(qsc.ref.n_nodes - 1):
        below = qsc.edge_leafset(e)
        if 1 < len(below) < len(names) - 1:
            found[below] = (qsc.getLQICScores()[e], qsc.getQPICScores()[e], qsc.getEQPICScores()[e])
    for k, v in g4["scores_changed"].items():
        kk = key_of(k)
        got = found.get(kk) or found.get(names - kk)
        assert list(got) == v, (k, got, v)
    for k, v in g1["scores"].items():
        if k in g4["scores_changed"]:
            continue
        kk = key_of(k)
        got = found.get(kk) or found.get(names - kk)
        assert list(got) == v, k


@pytest.mark.parametrize("case", ["D4", "rooted9", "rooted24", "rooted41"])
def test_rooted_reference_compact_mode_matches_oracle(eng, case):
    """A rooted reference tree in the reference's memory-efficient table mode (`-s` / enforceSmallMem): the reference's
    lookups for the root's node pairs repeat an id (QuartetScoreComputer.hpp:393-396), its compact table throws
    std::runtime_error for an index behind the table (quartet_lookup_table.hpp:79-85) and the run ends. The engine reports
    the same: QS_ERR_REFERENCE_THROWS carrying the what() of the first throwing call -- equal to the committed fixture and
    to the oracle run live (whose repeated-id arithmetic is pinned on the unmodified header, tests/test_oracle_reftable.py).
    Without enforceSmallMem the scores are the n^4 table's (test_rooted_reference_random_matches_oracle)."""
    import json
    import os
    from oracle_api import OracleError
    with open(os.path.join(os.path.dirname(__file__), "golden", "rooted_compact.json")) as f:
        fx = json.load(f)[case]
    o = Oracle(fx["ref"])
    o.count("\n".join(fx["eval"]), savemem=True, cint_bits=16)
    with pytest.raises(OracleError) as eo:
        o.score(nthreads=1)
    assert str(eo.value) == fx["reference_throws"]
    with pytest.raises(eng.QSError) as eg:
        eng.QuartetScoreComputer(fx["ref"], fx["eval"], None, False, True)          # (..., verboseOutput, enforceSmallMem)
    assert eg.value.code == _lib.QS_ERR_REFERENCE_THROWS and str(eg.value).endswith(fx["reference_throws"])
    # order of events = the CLI's = the reference's: a note, the counting and its log lines, then the exception from the scoring;
    # fail_fast=True (the CLI's --fail-fast) raises before anything is counted
    said = []
    with pytest.raises(eng.QSError):
        eng.QuartetScoreComputer(fx["ref"], fx["eval"], None, False, True, log=said.append)
    assert said[0].startswith("note:") and "Finished counting quartets." in said and "Finished computing scores." not in said
    said = []
    with pytest.raises(eng.QSError) as ef:
        eng.QuartetScoreComputer(fx["ref"], fx["eval"], None, False, True, fail_fast=True, log=said.append)
    assert ef.value.code == _lib.QS_ERR_REFERENCE_THROWS and said == []
    # the same inputs without -s: the runtime-efficient table's scores, equal to the oracle's
    qsc = eng.QuartetScoreComputer(fx["ref"], fx["eval"])
    o2 = oracle_counts(fx["ref"], fx["eval"])
    o2.score()
    want, got = o2.scores_by_bipartition(), qsc.scores_by_bipartition()
    assert set(got) == set(want)
    for k in got:
        assert got[k] == want[k], (sorted(k), got[k], want[k])
    # -s with --root-as-edge: no repeated ids are looked up, no exception
    eng.QuartetScoreComputer(fx["ref"], fx["eval"], None, False, True, root_as_edge=True)


@pytest.mark.parametrize("route", ["two_pass_bundle", "single_read_log", "scan_kernel", "overflow_lists"])
@pytest.mark.parametrize("case", ["rooted24", "rooted41"])
def test_rooted_reference_second_evaluation_order(eng, monkeypatch, case, route):
    """For a degree-2 root the reference's processNodePair(root, v) also walks quartets that another node pair owns and takes
    std::min of their log_score into the same edges (QuartetScoreComputer.hpp:393-396,417-454) -- for a leaf alone on one side
    of the root and an outsider that FOLLOWS v's subtree, with q2 and q3 exchanged, which changes the last bits of the sum
    (:141-156). The engine flags those quartets (root_swapped in qs_score.hip) through every route a candidate can take --
    the bundle kernel's pass 2, the candidate log of the single-read pass, the scan kernel, the overflow lists -- and the host
    evaluates both orders: LQ-IC identical to the oracle's, where an unflagged run is off by up to 2 ulp (this fixture)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "rooted_compact.json")) as f:
        fx = json.load(f)[case]
    tuning = {"two_pass_bundle": {_lib.QS_TUNE_SCORE_PASSES: 1}, "single_read_log": {_lib.QS_TUNE_SCORE_PASSES: 2},
              "scan_kernel": {_lib.QS_TUNE_SCORE_KERNEL: 1}, "overflow_lists": {_lib.QS_TUNE_SCORE_CAND_SLOTS: 1}}[route]
    for k_, v_ in tuning.items():
        monkeypatch.setitem(eng.DEFAULT_TUNING, k_, v_)
    o = oracle_counts(fx["ref"], fx["eval"])
    for exact in (False, True):
        o.score(qp_exact64=exact)
        want = o.scores_by_bipartition()
        qsc = eng.QuartetScoreComputer(fx["ref"], fx["eval"], qp_exact64=exact)
        got = qsc.scores_by_bipartition()
        assert set(got) == set(want)
        for k in got:
            assert got[k] == want[k], (route, exact, sorted(k), got[k], want[k])
    if route == "single_read_log":
        assert qsc.quartetCounterLookup.ctx.last_score_log() > 0


@pytest.mark.parametrize("n,seed", [(9, 71), (24, 72), (41, 73)])
def test_rooted_reference_random_matches_oracle(eng, n, seed):
    """Rooted random references (degree-2 root) against the oracle, which follows the reference's link arithmetic;
    also through table shards (every shard adds its part of the (root, v) sums) with wrap32 and 64-bit sums."""
    import torch
    rng = np.random.default_rng(seed)
    ref_nw = synth.random_tree(n, rng, rooted=True)
    trees = synth.tree_set(n, 60, seed + 100, collapse=0.1)
    o = oracle_counts(ref_nw, trees)
    for exact in (False, True):
        o.score(qp_exact64=exact)
        want = o.scores_by_bipartition()
        qsc = eng.QuartetScoreComputer(ref_nw, trees, qp_exact64=exact)
        got = qsc.scores_by_bipartition()
        assert set(got) == set(want)
        for k in got:
            assert got[k] == want[k], (exact, sorted(k), got[k], want[k])
    # three table shards
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctxs = []
    for d_lo, d_hi in ((0, n // 2), (n // 2, n - 2), (n - 2, n)):
        c = eng.Context(n, 32, d_lo=d_lo, d_hi=d_hi)
        c.table_alloc()
        c.count_trees(batch)
        ctxs.append(c)
    P = ctxs[0].score_pair_slots(ref)
    sums = torch.zeros(3 * P, dtype=torch.int64, device="cuda")
    mins = torch.full((P,), 2 ** 62, dtype=torch.int64, device="cuda")
    parts = []
    for c in ctxs:
        s_ = torch.empty_like(sums); m_ = torch.empty_like(mins)
        c.score_pass1(ref, s_, m_)
        torch.cuda.synchronize()
        sums += s_
        mins = torch.minimum(mins, m_)
    for c in ctxs:
        cd = torch.empty(8 * P, dtype=torch.int64, device="cuda")
        c.score_pass2(ref, mins, cd)
        torch.cuda.synchronize()
        parts.append(cd.cpu().numpy())
    lq, qp, eqp, _ = ctxs[0].score_finish(ref, sums.cpu().numpy(), np.stack(parts), eng.QS_SCORE_QP_EXACT64)
    whole = eng.QuartetScoreComputer(ref_nw, trees, qp_exact64=True)
    assert (lq[1:] == np.array(whole.getLQICScores())).all()
    assert (qp[1:] == np.array(whole.getQPICScores())).all()
    assert (eqp[1:] == np.array(whole.getEQPICScores())).all()


def test_rooted_reference_as_edge_subdivision(eng, golden):
    """QS_SCORE_ROOT_AS_EDGE: a degree-2 root is an edge subdivision; both root edges carry the scores the unrooted
    tree gives that internode (not what the reference prints: quirk Q5)."""
    g4, g1 = golden["D4"], golden["D1"]
    qsc = eng.QuartetScoreComputer(g4["ref"], g1["eval"], root_as_edge=True)
    found = {qsc.edge_leafset(e): (qsc.getLQICScores()[e], qsc.getQPICScores()[e], qsc.getEQPICScores()[e])
             for e in range(qsc.ref.n_nodes - 1)}
    want = g1["scores"]["t1,t3,t4,t5,t6"]
    assert list(found[key_of("t1,t3,t4,t5,t6")]) == want
    assert list(found[key_of("t0,t2,t7")]) == want
    for k, v in g1["scores"].items():
        if k != "t1,t3,t4,t5,t6":
            assert list(found[key_of(k)]) == v


def test_raw_qic_lines_match_oracle(eng, tmp_path):
    n = 10
    ref_nw = synth.random_tree(n, np.random.default_rng(41), collapse=0.2)
    trees = synth.tree_set(n, 25, 42, dropout=0.1)
    qsc = eng.QuartetScoreComputer(ref_nw, trees)
    p1, p2 = str(tmp_path / "gpu.txt"), str(tmp_path / "ora.txt")
    qsc.printRawQICScores(p1)
    o = oracle_counts(ref_nw, trees)
    o.raw_qic(p2)

    def canon(path):
        out = {}
        for line in open(path):
            lab, val = line.strip().split("): ")
            l, r = lab[1:].split("|")
            key = frozenset([frozenset(l.split(",")), frozenset(r.split(","))])
            out[key] = val
        return out
    a, b = canon(p1), canon(p2)
    assert set(a) == set(b)
    # same quartet, same topology; the printed QIC may differ only in the p2/p3 argument order
    same = sum(a[k] == b[k] for k in a)
    assert same == len(a)
    # and the FILE is the reference's: same line order (its four nested loops over the Euler-tour leaves), same order of
    # the labels inside a line
    assert open(p1).read() == open(p2).read()


# ---- BASELINE configs[1] at full size: size-independent properties -----------------------------

def test_config2_full_size_properties(eng):
    n, m = 128, 1000
    ref_nw = synth.reference_tree(n, 2000)
    trees = synth.tree_set(n, m, 2001)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx, T = gpu_table(eng, ref, batch, 32)
    assert "binary_full/bitslice" in ctx.last_count_variant()
    assert T.shape == (ranks.n_quartets(n), 3)
    assert (T.sum(axis=1, dtype=np.uint64) == m).all()                 # every tree resolves every quartet once
    # linearity: counting two halves separately and together gives the same table
    _, Th = gpu_table(eng, ref, batch, 32, split=500)
    assert (T == Th).all()
    # independent formulation (tree-major atomics) agrees on a 100-tree prefix, and so does the oracle
    sub = batch.slice(0, 100)
    _, Tg = gpu_table(eng, ref, sub, 32)
    _, Ts = gpu_table(eng, ref, sub, 32, algo=eng.QS_ALGO_SCATTER)
    assert (Tg == Ts).all()
    o = oracle_counts(ref_nw, trees[:100])
    assert (Tg.astype(np.uint64) == o.counts()).all()
    # ... and the scores of that 100-tree table are identical to the oracle's at n = 128
    qsc = eng.QuartetScoreComputer(ref, trees[:100])
    o.score(nthreads=8)
    assert assert_scores_equal(qsc.scores_by_bipartition(), o.scores_by_bipartition()) == 0
    # scores on the full table are finite on every internal edge and idempotent
    lq, qp, eqp, bif = ctx.score(ref)
    lq2, qp2, eqp2, _ = ctx.score(ref)
    assert bif and (lq == lq2).all() and (qp == qp2).all() and (eqp == eqp2).all()
    internal = [v for v in range(1, ref.n_nodes) if ref.nodes[v].children]
    assert np.isfinite(lq[internal]).all() and np.isfinite(qp[internal]).all() and np.isfinite(eqp[internal]).all()
    assert (eqp[internal] <= qp[internal]).all()                       # EQP-IC is a min that includes the edge's own pair


@pytest.mark.parametrize("kw", [{}, {"collapse": 0.15, "dropout": 0.1}])
def test_scores_above_128_inner_nodes_take_the_threaded_finalisation(eng, kw):
    """From 128 inner nodes on (n >= 130) qs_score_finish folds the node pairs on several host threads and the
    large-table variant of pass 1 is close: scores must still equal the oracle's, bit for bit."""
    n, m = 134, 8        # 132 inner nodes; the oracle's compact table costs ~m * n^4 / 12 increments
    ref_nw, trees = make_case(n, m, 131, **kw)
    o = Oracle(ref_nw)
    o.count("\n".join(trees), savemem=True, cint_bits=16)      # compact oracle table (the n^4 one would be 1 GB)
    o.score(nthreads=8)
    qsc = eng.QuartetScoreComputer(ref_nw, trees)
    assert assert_scores_equal(qsc.scores_by_bipartition(), o.scores_by_bipartition()) == 0


def test_pack16_wire_format(eng):
    """qs_table_pack16: the u32 table as u16 cells is a valid count_bits=16 table (same counts, same scores);
    packed words add without carries (what the multi-GPU all-reduce relies on); a cell >= 2^16 is reported."""
    import torch
    from quartetscores_amd import distributed
    n, m = 19, 300   # odd number of cells -> the padded last word is exercised too
    ref_nw = synth.reference_tree(n, 71)
    trees = synth.tree_set(n, m, 72, collapse=0.1, dropout=0.05)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(batch)
    want = ctx.table_download()
    packed = torch.full((distributed.table_words(ctx.table_tuples, 16),), -1, dtype=torch.int32, device="cuda")
    ctx.table_pack16(packed)
    ctx.sync()
    c16 = eng.Context(n, 16)
    c16.table_attach(packed)
    assert np.array_equal(c16.table_download().astype(np.uint32), want)
    for a, b in zip(ctx.score(ref)[:3], c16.score(ref)[:3]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    # "all-reduce" of three ranks' packed tables == packed sum
    total = packed * 3
    c16.table_attach(total)
    assert np.array_equal(c16.table_download().astype(np.uint32), want * 3)
    # too small a destination / 16-bit source are argument errors
    with pytest.raises(eng.QSError):
        ctx.table_pack16(packed[:-1])
    with pytest.raises(eng.QSError):
        c16.table_pack16(packed)
    # overflow
    big = want.copy()
    big[5, 1] = 70000
    ctx.table_upload(big)
    ctx.table_pack16(packed)
    with pytest.raises(eng.QSError) as ei:
        ctx.sync()
    assert ei.value.code == -5


def test_count_trees_multi_gpu_u16_wire_single_rank(eng):
    """distributed.count_trees_multi_gpu(wire='u16') on one rank: same table and scores as the plain path."""
    from quartetscores_amd import distributed
    n, m = 16, 120
    ref_nw = synth.reference_tree(n, 81)
    trees = synth.tree_set(n, m, 82)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    c_a, _ = distributed.count_trees_multi_gpu(ref, batch, wire="u32")
    c_b, _ = distributed.count_trees_multi_gpu(ref, batch, wire="u16")
    assert c_b.count_bits == 16
    assert np.array_equal(c_a.table_download(), c_b.table_download().astype(np.uint32))
    for a, b in zip(c_a.score(ref)[:3], c_b.score(ref)[:3]):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("bits", [16, 32])
def test_scoring_views_of_a_reduce_scattered_table(eng, bits):
    """The shard a rank holds after distributed.reduce_scatter_table (tuples [r*T, (r+1)*T) in its own buffer) is
    scored through qs_score_set_view; three views on one GPU stand in for three ranks, combined like
    distributed.score_sharded does with collectives. Same scores as the unsharded path, bit for bit."""
    import torch
    from quartetscores_amd import distributed
    n, m, W = 23, 60, 3
    ref_nw, trees = make_case(n, m, 91, collapse=0.1, dropout=0.1)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    whole = eng.Context(n, 32)
    whole.table_alloc()
    whole.count_trees(batch)
    want = whole.score(ref)
    full = whole.table_download()
    nq = full.shape[0]
    t_chunk, words = distributed.scatter_layout(nq, W, bits)
    dt = np.uint16 if bits == 16 else np.uint32
    ctx = eng.Context(n, 32)            # no table of its own: views only
    P = ctx.score_pair_slots(ref)
    sums, mins, shards = None, None, []
    for r in range(W):
        r_lo, n_own = distributed.scatter_owned(nq, W, r, bits)
        buf = np.zeros(words * 4 // dt().itemsize, dtype=dt)
        buf[: n_own * 3] = full[r_lo:r_lo + n_own].astype(dt).ravel()
        shard = torch.from_numpy(buf.view(np.int32)).cuda()
        shards.append((shard, r_lo, n_own))
        ctx.score_set_view(shard, bits, r_lo, n_own)
        s_ = torch.empty(3 * P, dtype=torch.int64, device="cuda"); m_ = torch.empty(P, dtype=torch.int64, device="cuda")
        ctx.score_pass1(ref, s_, m_)
        sums = s_ if sums is None else sums + s_
        mins = m_ if mins is None else torch.minimum(mins, m_)
    cands = []
    for shard, r_lo, n_own in shards:
        ctx.score_set_view(shard, bits, r_lo, n_own)
        c_ = torch.empty(8 * P, dtype=torch.int64, device="cuda")
        ctx.score_pass2(ref, mins, c_)
        cands.append(c_.cpu().numpy())
    got = ctx.score_finish(ref, sums.cpu().numpy(), np.stack(cands))
    for a, b in zip(got[:3], want[:3]):
        assert np.array_equal(a, b, equal_nan=True)
    # argument checks; clearing the view without an own table is a state error again
    with pytest.raises(eng.QSError):
        ctx.score_set_view(shards[0][0], bits, ranks.n_quartets(n), 1)
    ctx.score_set_view(None, 0, 0, 0)
    with pytest.raises(eng.QSError):
        ctx.score_pass1(ref, sums, mins)


def test_count_trees_reduce_scatter_single_rank(eng):
    from quartetscores_amd import distributed
    n, m = 18, 90
    ref_nw, trees = make_case(n, m, 93)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    plain, _ = distributed.count_trees_multi_gpu(ref, batch, wire="u32")
    want = plain.score(ref)
    for wire in ("u16x2", "u16", "u32"):
        ctx, shard, bits, r_lo, n_own = distributed.count_trees_reduce_scatter(ref, batch, wire=wire)
        assert (bits, r_lo, n_own) == (32 if wire == "u32" else 16, 0, ranks.n_quartets(n))
        got = distributed.score_sharded(ctx, ref)
        for a, b in zip(got[:3], want[:3]):
            assert np.array_equal(a, b, equal_nan=True)


def test_config1_fixture_on_the_gpu(eng):
    """BASELINE configs[0] against the committed fixture (tests/golden/config1.json): table hash in canonical
    taxon order and every internal edge's LQ/QP/EQP-IC as hex doubles."""
    import hashlib
    import json
    import os
    from helpers import remap_table
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.json")))
    n, m = fx["n"], fx["m"]
    trees = synth.tree_set(n, m, fx["eval_seed"])
    assert synth.reference_tree(n, fx["ref_seed"]) == fx["ref"] and trees[0] == fx["first_tree"] and trees[-1] == fx["last_tree"]
    qsc = eng.QuartetScoreComputer(fx["ref"], trees, device=0)
    names = list(qsc.ref.names)
    perm = [names.index(f"t{i}") for i in range(n)]
    table = remap_table(qsc.quartetCounterLookup.table(), perm).astype("<u4")
    assert hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest() == fx["table_sha256"]
    got = {",".join(sorted(k, key=lambda s: int(s[1:]))): [float(x).hex() for x in v]
           for k, v in qsc.scores_by_bipartition().items()}
    assert got == fx["scores_lq_qp_eqp_hex"]


@pytest.mark.parametrize("kind", ["random", "collapsed", "identical", "rooted_ref", "multifurcating_ref"])
def test_single_read_scoring_equals_two_passes(eng, monkeypatch, kind):
    """qs_score's single-read mode (QS_TUNE_SCORE_PASSES = 2: pass 1 logs every quartet that is near-minimal for its node
    pair at that moment, a filter over the log replaces pass 2; QuartetScoreComputer.hpp:417-469 visits every quartet
    once, too): the scores are identical, bit for bit, to the default two passes and to the oracle -- also when the log
    overflows and the call falls back to a second pass by itself (forced with a 16-record log; identical trees make
    every quartet a minimiser)."""
    n, m = 37, 90
    if kind == "multifurcating_ref":
        ref_nw = synth.tree_set(n, 1, 511, collapse=0.3)[0]
    else:
        ref_nw = synth.random_tree(n, np.random.default_rng(510), rooted=(kind == "rooted_ref"))
    ref = flatten.flatten_reference(ref_nw)
    if kind == "identical":
        trees = [synth.reference_tree(n, 512)] * m
    else:
        trees = synth.tree_set(n, m, 513, collapse=0.2 if kind == "collapsed" else 0.0)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    o = oracle_counts(ref_nw, trees)
    o.score()
    want = o.scores_by_bipartition()
    results = {}
    # "single": with the default pre-pass (minima of one chunk in 16 of every row); without one; one chunk in 2; one round in 4
    for mode, tuning in (("single", {_lib.QS_TUNE_SCORE_PASSES: 2}), ("two", {_lib.QS_TUNE_SCORE_PASSES: 1}), ("auto", {}),
                         ("single_nopre", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_SAMPLE: 0}),
                         ("single_s2", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_SAMPLE: 2}),
                         ("single_r4", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_SAMPLE: 4 | 65536}),
                         ("single_noties", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_DEDUPE: 0}),
                         ("overflow", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_LOG_CAP: 16}),
                         # capacities that are multiples of the 64-record chunk waves reserve: the log's counter can stop exactly
                         # AT the capacity while waves that found it full dropped their hits -- a full log must count as an overflow
                         ("cap64", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_LOG_CAP: 64}),
                         ("cap128", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_LOG_CAP: 128}),
                         ("cap256", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_LOG_CAP: 256}),
                         ("cap1024", {_lib.QS_TUNE_SCORE_PASSES: 2, _lib.QS_TUNE_SCORE_LOG_CAP: 1024})):
        for k_, v_ in tuning.items():
            monkeypatch.setitem(eng.DEFAULT_TUNING, k_, v_)
        ctx = eng.Context(ref.n_taxa, 32)
        ctx.table_alloc()
        ctx.count_trees(batch)
        lq, qp, eqp, bif = ctx.score(ref)
        results[mode] = (lq.copy(), qp.copy(), eqp.copy(), bif, ctx.last_score_log())
        for k_ in tuning:
            monkeypatch.delitem(eng.DEFAULT_TUNING, k_)
    assert results["single"][4] > 0 and results["two"][4] == 0 and results["overflow"][4] == 0
    for mode in ("single_nopre", "single_s2", "single_r4"):
        assert results[mode][4] > 0, mode
    assert results["auto"][4] == 0          # (a table below 1 GB: the automatic mode reads it twice)
    for mode in ("cap64", "cap128", "cap256", "cap1024"):     # a log that is used must have had room left
        assert results[mode][4] < int(mode[3:]), (mode, results[mode][4])
    for mode in ("two", "auto", "overflow", "single_nopre", "single_s2", "single_r4", "single_noties", "cap64", "cap128", "cap256", "cap1024"):
        for i in range(3):
            assert np.array_equal(results["single"][i], results[mode][i], equal_nan=True), (mode, i)
    qsc_like = results["single"]
    got = {}
    names = ref.names
    from quartetscores_amd import newick
    for e in range(ref.n_nodes - 1):
        below = frozenset(x.name for x in newick.preorder(ref.nodes[e + 1]) if x.is_leaf)
        if len(below) <= 1 or len(below) >= n - 1:
            continue
        other = frozenset(names) - below
        key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
        while key in got:
            key = frozenset(list(key) + ["#dup"])
        got[key] = (qsc_like[0][e + 1], qsc_like[1][e + 1] if qsc_like[3] else None, qsc_like[2][e + 1] if qsc_like[3] else None)
    if kind != "rooted_ref":   # (the two root edges share one bipartition: their order in the two dictionaries is not defined)
        assert set(got) == set(want)
        for k_ in got:
            assert got[k_] == want[k_], (sorted(k_), got[k_], want[k_])


@pytest.mark.parametrize("kind", ["random", "nni", "identical"])
def test_automatic_scoring_mode_on_a_gigabyte_table(eng, kind):
    """qs_score's default on a table of 1.2 GB (224 taxa): a minima-only pre-pass over one round in 64, a second sample
    that predicts the candidate log, then ONE read of the table -- or, when the prediction says the log would not hold
    (identical trees, where every quartet ties its node pair's minimum, and a log of 8192 records), two plain passes. Scores bit for bit
    those of the forced two-pass mode in every case."""
    from quartetscores_amd import native_ingest
    n, m = 224, 96
    ref_nw = native_ingest.synth_trees(n, 1, 610).decode().strip()
    if kind == "identical":
        text = (ref_nw + "\n").encode() * m
    else:
        text = native_ingest.synth_trees(n, m, 611, kind="nni" if kind == "nni" else "random", ref_text=ref_nw if kind == "nni" else None)
    ref = flatten.flatten_reference(ref_nw)
    batch, _ = native_ingest.ingest_text(ref_nw, text, want_ranges=False)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    assert ctx.table_bytes >= 1 << 30
    ctx.count_trees(batch)
    got = {}
    if kind == "identical":     # a small log: the sample must predict that it will not hold
        ctx.set_tuning(_lib.QS_TUNE_SCORE_LOG_CAP, 1 << 13)
    for mode in (1, 0, 2):
        ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, mode)
        lq, qp, eqp, bif = ctx.score(ref)
        got[mode] = (lq.copy(), qp.copy(), eqp.copy(), ctx.last_score_log(), ctx.last_score_estimate())
    ctx.close()
    assert got[1][3] == 0 and got[1][4] == 0
    assert got[0][4] > 0 or kind == "identical" or got[0][3] > 0        # the automatic mode made an estimate
    if kind == "random":
        assert got[0][3] > 0 and got[2][3] > 0                           # ... and read the table once
    if kind == "identical":
        assert got[0][3] == 0 and got[0][4] > 6 * (1 << 13)              # predicted overflow: two passes
        assert got[2][3] == 0                                            # forced single read: the log overflows, pass 2 follows
    for mode in (0, 2):
        for i in range(3):
            assert np.array_equal(got[1][i], got[mode][i], equal_nan=True), (mode, i)


def test_score_prepare_behind_enqueued_counts(eng):
    """qs_score_prepare (the CLI calls it behind its last qs_count_batch, while the device still counts): same scores as a
    context that never heard of it, for a bifurcating, a rooted and a multifurcating reference; calling it twice, or after
    the count has finished, changes nothing."""
    from quartetscores_amd import native_ingest
    n, m = 150, 600
    text = native_ingest.synth_trees(n, m, 621)
    for kind in ("binary", "rooted", "multifurcating"):
        if kind == "multifurcating":
            ref_nw = synth.tree_set(n, 1, 622, collapse=0.3)[0]
        else:
            ref_nw = synth.random_tree(n, np.random.default_rng(623), names=[f"t{i}" for i in range(n)], rooted=(kind == "rooted"))
        ref = flatten.flatten_reference(ref_nw)
        batch, _ = native_ingest.ingest_text(ref_nw, text, want_ranges=False)
        res = []
        for prepare in (False, True):
            ctx = eng.Context(n, 32)
            ctx.table_alloc()
            hb = ctx.batch_upload(batch, with_nodes=False)
            ctx.count_batch(hb)                      # asynchronous
            if prepare:
                ctx.score_prepare(ref, m)            # ... the set-up of the scoring behind it
                ctx.score_prepare(ref, m)
            sc = ctx.score(ref)
            if prepare:
                ctx.score_prepare(ref, m)
                sc2 = ctx.score(ref)
                assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(sc[:3], sc2[:3]))
            res.append(sc)
            ctx.close()
        assert res[0][3] == res[1][3]
        for i in range(3):
            assert np.array_equal(res[0][i], res[1][i], equal_nan=True), (kind, i)


def test_two_cell_wire_format(eng):
    """qs_table_pack16x2 / qs_unpack16x2: one word per tuple for batches of binary trees holding all taxa; the words
    of several ranks add without carries; the unpacked table equals the three-cell one; anything else is refused."""
    import torch
    n, m = 21, 77
    ref_nw, trees = make_case(n, m, 95)
    ref = flatten.flatten_reference(ref_nw)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(flatten.flatten_eval_trees(trees, ref.name_to_id))
    want = ctx.table_download()
    nq = want.shape[0]
    words = torch.full((nq,), -1, dtype=torch.int32, device="cuda")
    ctx.table_pack16x2(words)
    ctx.sync()
    w = words.cpu().numpy().view(np.uint32)
    assert np.array_equal(w & 0xFFFF, want[:, 0]) and np.array_equal(w >> 16, want[:, 1])
    out = torch.zeros((nq * 3 + 1) // 2, dtype=torch.int32, device="cuda")
    ctx.unpack16x2(words * 3, nq, 3 * m, out)          # "three ranks" with the same trees
    ctx.sync()
    got = out.cpu().numpy().view(np.uint16)[: nq * 3].reshape(nq, 3)
    assert np.array_equal(got.astype(np.uint32), want * 3)
    # a batch with an unresolved quartet does not fit the format
    ctx2 = eng.Context(n, 32)
    ctx2.table_alloc()
    ctx2.count_trees(flatten.flatten_eval_trees(synth.tree_set(n, 30, 96, collapse=0.3), ref.name_to_id))
    ctx2.table_pack16x2(words)
    with pytest.raises(eng.QSError) as ei:
        ctx2.sync()
    assert ei.value.code == -4 and "two-cell" in str(ei.value)
    with pytest.raises(eng.QSError):
        ctx.table_pack16x2(words[:-1])


def test_two_cell_wire_format_u32(eng):
    """qs_table_pack32x2 / qs_unpack32x2 (totals of 65536 trees and more: BASELINE configs[3] moves 8 instead of 12 bytes per
    quartet): (n0, n1) per tuple; the pairs of several ranks add; the unpacked table equals the three-cell one with
    n2 = total - n0 - n1, also for totals beyond 2^16; a batch with an unresolved quartet is refused; and the tree-sharded
    driver picks the format by itself (wire = "auto") once the total reaches 65536 trees."""
    import torch
    from quartetscores_amd import distributed
    n, m = 21, 77
    ref_nw, trees = make_case(n, m, 95)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(batch)
    want = ctx.table_download()
    nq = want.shape[0]
    words = torch.full((2 * nq,), -1, dtype=torch.int32, device="cuda")
    ctx.table_pack32x2(words)
    ctx.sync()
    w = words.cpu().numpy().view(np.uint32).reshape(nq, 2)
    assert np.array_equal(w[:, 0], want[:, 0]) and np.array_equal(w[:, 1], want[:, 1])
    ranks_ = 1000                                        # "1000 ranks" with the same trees: totals beyond 2^16
    out = torch.zeros(nq * 3, dtype=torch.int32, device="cuda")
    ctx.unpack32x2(words * ranks_, nq, ranks_ * m, out)
    ctx.sync()
    got = out.cpu().numpy().view(np.uint32).reshape(nq, 3)
    assert ranks_ * m > 65535 and np.array_equal(got, want * ranks_)
    ctx2 = eng.Context(n, 32)
    ctx2.table_alloc()
    ctx2.count_trees(flatten.flatten_eval_trees(synth.tree_set(n, 30, 96, collapse=0.3), ref.name_to_id))
    ctx2.table_pack32x2(words)
    with pytest.raises(eng.QSError) as ei:
        ctx2.sync()
    assert ei.value.code == -4 and "two-cell" in str(ei.value)
    with pytest.raises(eng.QSError):
        ctx.table_pack32x2(words[:-1])
    # the driver: explicit "u32x2" on one rank gives the table back; "auto" below 65536 trees stays with 16-bit cells
    c3, shard, bits, r_lo, n_own = distributed.reduce_scatter_counts(ref, batch, m, wire="u32x2")
    assert (bits, r_lo, n_own) == (32, 0, nq)
    assert np.array_equal(shard.cpu().numpy().view(np.uint32)[: nq * 3].reshape(nq, 3), want)
    _, _, bits16, _, _ = distributed.reduce_scatter_counts(ref, batch, m, wire="auto")
    assert bits16 == 16


def test_large_batch_validation_runs_on_several_host_threads(eng):
    """qs_batch_upload checks batches of >= 2048 trees with a pool of host threads: same table as many small
    batches, and the error reported is the one of the FIRST bad tree in tree order."""
    n, m = 9, 5000
    ref_nw, trees = make_case(n, m, 97, collapse=0.1)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    _, T1 = gpu_table(eng, ref, batch)
    _, T2 = gpu_table(eng, ref, batch, split=700)
    assert np.array_equal(T1, T2)
    assert (T1.astype(np.uint64) == oracle_counts(ref_nw, trees).counts()).all()
    bad = flatten.TreeBatch(batch.n_trees, batch.leaf_off.copy(), batch.leaf_ids.copy(), batch.adj_depth.copy(),
                            batch.node_off.copy(), batch.rng_off.copy(), batch.ranges.copy())
    for t in (4100, 2500):     # the later error is planted first
        bad.leaf_ids[bad.leaf_off[t]] = n + 3
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    with pytest.raises(eng.QSError) as ei:
        ctx.count_trees(bad)
    assert "tree 2500" in str(ei.value)


def test_counting_straight_into_the_wire_format(eng, monkeypatch):
    """QS_COUNT_WIRE16X2: the count kernel writes one word n0 | n1 << 16 per tuple into the attached buffer: equal to
    qs_table_pack16x2 of the table the normal path builds, also over several panel slices, accumulated over two
    calls, and with overwrite; refused for batches that are not binary with all taxa."""
    import torch
    n, m = 37, 150
    ref_nw, trees = make_case(n, m, 141)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(batch)
    nq = ctx.table_tuples
    want = torch.zeros(nq, dtype=torch.int32, device="cuda")
    ctx.table_pack16x2(want)
    ctx.sync()
    W = eng.QS_ALGO_GATHER | eng.QS_COUNT_WIRE16X2

    def wire_count(parts, overwrite_first=False, slice_bytes=None):
        if slice_bytes:
            monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_SLICE_BYTES, slice_bytes)
        c2 = eng.Context(n, 32)                      # no table at all
        words = torch.full((nq,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") if overwrite_first else torch.zeros(nq, dtype=torch.int32, device="cuda")
        c2.wire_attach(words)
        for k, (lo, hi) in enumerate(parts):
            hb = c2.batch_upload(batch.slice(lo, hi), with_nodes=False)
            c2.count_batch(hb, W | (eng.QS_COUNT_OVERWRITE if (overwrite_first and k == 0) else 0))
            c2.sync()
            c2.batch_free(hb)
        if slice_bytes:
            monkeypatch.delitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_SLICE_BYTES)
        assert "wire_u16x2" in c2.last_count_variant()
        return words

    assert torch.equal(wire_count([(0, m)]), want)
    assert torch.equal(wire_count([(0, 64), (64, m)]), want)                       # accumulation over calls
    assert torch.equal(wire_count([(0, m)], overwrite_first=True), want)           # stale contents discarded
    assert torch.equal(wire_count([(0, m)], slice_bytes=40000), want)              # several panel slices per call
    # refused: multifurcating trees; missing wire buffer
    c3 = eng.Context(n, 32)
    c3.wire_attach(torch.zeros(nq, dtype=torch.int32, device="cuda"))
    hb = c3.batch_upload(flatten.flatten_eval_trees(synth.tree_set(n, 20, 142, collapse=0.3), ref.name_to_id), with_nodes=False)
    with pytest.raises(eng.QSError) as ei:
        c3.count_batch(hb, W)
    assert ei.value.code == -4
    c3.wire_attach(None)
    hb2 = c3.batch_upload(batch.slice(0, 10), with_nodes=False)
    with pytest.raises(eng.QSError):
        c3.count_batch(hb2, W)


@pytest.mark.parametrize("slots,tol_exp", [(1, 12), (1, 2), (8, 1)])
def test_candidate_overflow_falls_back_to_the_full_list(eng, monkeypatch, slots, tol_exp):
    """Node pairs with more near-minimal count triples than candidate slots are marked by pass 2 and finished from
    the list of qs_score_overflow: scores stay identical to the oracle. Forced here with 1 slot / a tolerance of 1e-2."""
    import torch
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_SCORE_CAND_SLOTS, slots)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_SCORE_TOL_EXP, tol_exp)
    ref_nw, trees = make_case(22, 90, 501, collapse=0.1)
    qsc = eng.QuartetScoreComputer(ref_nw, trees)
    o = oracle_counts(ref_nw, trees)
    o.score()
    a, b = qsc.scores_by_bipartition(), o.scores_by_bipartition()
    assert set(a) == set(b)
    for k in a:
        assert a[k] == b[k], (sorted(k), a[k], b[k])
    # the overflow really happened, and the stepwise API reports it
    ctx, ref = qsc.quartetCounterLookup.ctx, qsc.ref
    P = ctx.score_pair_slots(ref)
    sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
    cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
    ctx.score_pass1(ref, sums, mins)
    ctx.score_pass2(ref, mins, cand)
    extra = ctx.score_overflow(ref, mins, cand)
    assert len(extra) > 0 and (cand.view(P, 8)[:, 7] == -2).any()
    lq, qp, eqp, _ = ctx.score_finish(ref, sums.cpu().numpy(), cand.cpu().numpy()[None, :], extra=extra)
    from quartetscores_amd import distributed
    lq2, qp2, eqp2, _ = distributed.score_sharded(ctx, ref)
    assert (lq == lq2).all() and (qp == qp2).all() and (eqp == eqp2).all()
    assert (lq[1:] == np.array(qsc.getLQICScores())).all()


def test_wide_reduced_counts_are_finished_from_the_list(eng):
    """Counts whose gcd-reduced triple needs more than 21 bits per component (2.1 million trees and more) do not fit a
    packed candidate slot; such node pairs take the same fallback. Expected values: tests/emulate.scores_from_table."""
    import emulate
    n = 13
    ref_nw = synth.reference_tree(n, 601)
    ref = flatten.flatten_reference(ref_nw)
    rng = np.random.default_rng(602)
    nq = ranks.n_quartets(n)
    T = rng.integers(2_200_000, 9_000_000, size=(nq, 3)).astype(np.uint32)
    T[rng.random(nq) < 0.3] //= 1000          # a mix of narrow and wide tuples
    T |= 1                                    # odd: the gcd rarely helps
    ctx = eng.Context(n, 32)
    ctx.table_alloc()
    ctx.table_upload(T)
    for flags, exact in ((eng.QS_SCORE_QP_WRAP32, False), (eng.QS_SCORE_QP_EXACT64, True)):
        lq, qp, eqp, bif = ctx.score(ref, flags)
        assert bif
        wlq, wqp, weqp = emulate.scores_from_table(ref, T.astype(np.int64), qp_exact64=exact)
        inner = np.isfinite(wlq)
        assert inner.sum() == n - 3
        assert (lq[inner] == wlq[inner]).all() and (qp[inner] == wqp[inner]).all() and (eqp[inner] == weqp[inner]).all()


@pytest.mark.parametrize("n,bits,kind", [(23, 32, "binary"), (70, 16, "multif"), (134, 32, "binary"), (41, 16, "ties"),
                                         (9, 32, "binary"), (160, 16, "ties")])
def test_score_bundle_kernel_equals_scan_kernel(eng, n, bits, kind):
    """Both kernels of score passes 1 and 2 (QS_TUNE_SCORE_KERNEL: 0 = bundle kernel, a wave walks 64 table rows with
    the same second id in lockstep; 1 = scan kernel, a lane walks 8 consecutive ranks) give the same per-node-pair sums
    bit for bit, minima that agree to the rounding of the device QIC, and the same final scores -- on random tables
    (incl. zero tuples and long ties such as (m,0,0)), u16 and u32 cells, counts beyond the LDS copy of the log table,
    bifurcating and multifurcating references, the whole table and ragged views of it (a view that starts and ends inside
    a row, lies inside one row, is a few tuples long): the bundle kernel's host plan (full rows per b + partial rows)."""
    import torch
    import emulate
    ref_nw = synth.reference_tree(n, 700 + n) if kind != "multif" else synth.tree_set(n, 1, 700 + n, collapse=0.3)[0]
    ref = flatten.flatten_reference(ref_nw)
    rng = np.random.default_rng(800 + n)
    nq = ranks.n_quartets(n)
    m = 5000 if bits == 16 else 200000          # beyond the LDS copy of the log table for u32
    T = rng.multinomial(m, [0.6, 0.3, 0.1], size=nq).astype(np.uint32)
    T = np.take_along_axis(T, rng.permuted(np.tile(np.arange(3), (nq, 1)), axis=1), axis=1)
    T[rng.random(nq) < 0.05] = 0
    if kind == "ties":
        tie = rng.random(nq) < 0.7
        T[tie] = np.array([m, 0, 0], dtype=np.uint32)[rng.permuted(np.tile(np.arange(3), (int(tie.sum()), 1)), axis=1)]
    dt = np.uint16 if bits == 16 else np.uint32
    ctx = eng.Context(n, bits)
    ctx.table_alloc()
    ctx.table_upload(T.astype(dt))
    P = ctx.score_pair_slots(ref)

    def steps(kernel):
        ctx.set_tuning(_lib.QS_TUNE_SCORE_KERNEL, kernel)
        sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
        cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
        ctx.score_pass1(ref, sums, mins)
        ctx.score_pass2(ref, mins, cand)
        extra = ctx.score_overflow(ref, mins, cand)
        sh, ch = sums.cpu().numpy(), cand.cpu().numpy()
        return sh, mins.cpu().numpy(), ctx.score_finish(ref, sh, ch[None, :], extra=extra)

    def same(x, y, what):
        assert (x[0] == y[0]).all(), what                      # sums: exact
        def dec(v):   # qs_common.hpp sortable_to_f64
            v = v.astype(np.int64)
            bits = np.where(v < 0, (np.uint64(1 << 63) - v.astype(np.uint64)).astype(np.uint64), v.astype(np.uint64))
            return bits.view(np.float64)
        with np.errstate(over="ignore"):
            mx, my = dec(x[1]), dec(y[1])
        assert (np.isclose(mx, my, rtol=0, atol=1e-12) | (x[1] == y[1])).all(), what   # minima of the device QIC
        for u, v in zip(x[2][:3], y[2][:3]):
            assert np.array_equal(u, v, equal_nan=True), what

    a, b = steps(0), steps(1)
    same(a, b, "whole table")
    if kind != "multif" and n <= 41:    # (plain-Python closed form: small tables only)
        wlq, wqp, weqp = emulate.scores_from_table(ref, T.astype(np.int64), qp_exact64=False)
        inner = np.isfinite(wlq)
        got = a[2]
        assert (got[0][inner] == wlq[inner]).all() and (got[1][inner] == wqp[inner]).all() and (got[2][inner] == weqp[inner]).all()
    # ragged views of the same table: (first rank, tuples)
    full = torch.from_numpy(np.ascontiguousarray(T.astype(dt)).reshape(-1).view(np.uint8)).cuda()
    item = 3 * (bits // 8)
    views = [(0, nq), (nq // 3 + 1, nq // 2), (nq - 5, 5), (7, 8192 + 9), (nq // 2, 3), (nq // 2 + 1, 1), (0, 1), (nq // 5, nq // 7)]
    for r_lo, cnt in views:
        if r_lo + cnt > nq or (r_lo * item) % 4:
            continue
        pad = (-cnt * item) % 4
        shard = torch.zeros(cnt * item + pad, dtype=torch.uint8, device="cuda")
        shard[: cnt * item] = full[r_lo * item:(r_lo + cnt) * item]
        ctx.score_set_view(shard.view(torch.int32), bits, r_lo, cnt)
        same(steps(0), steps(1), (r_lo, cnt))
    ctx.score_set_view(None, 0, 0, 0)
    ctx.close()


@pytest.mark.parametrize("dropout,collapse,variant", [(0.0, 0.25, "general_full"), (0.12, 0.0, "partial"), (0.1, 0.2, "partial"),
                                                      (0.0, 0.0, "binary_full")])
def test_lockstep_launches_match_the_swar_kernel(eng, dropout, collapse, variant):
    """From 200 taxa on the four waves of a count workgroup take their 32-tree steps together (one barrier per step).
    Every mode of the bit-sliced kernel -- multifurcating trees, missing taxa, binary -- at such a size, several tree
    groups and a ragged last one, both cell widths: the table equals the byte-SWAR kernel's (no barrier) bit for bit,
    and a sample of it the split-based brute force."""
    import bruteforce
    n, m = 212, 150
    ref_nw, trees = make_case(n, m, 977, dropout=dropout, collapse=collapse)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    for bits in (32, 16):
        ctx = eng.Context(n, bits)
        ctx.table_alloc()
        hb = ctx.batch_upload(batch, with_nodes=False)
        ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
        ctx.sync()
        assert variant in ctx.last_count_variant() and "bitslice" in ctx.last_count_variant(), ctx.last_count_variant()
        mine = ctx.table_download()
        ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
        ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
        ctx.sync()
        assert "depth_u" in ctx.last_count_variant()
        assert (mine == ctx.table_download()).all()
        rng = np.random.default_rng(978)
        q = np.sort(np.stack([rng.choice(n, size=4, replace=False) for _ in range(3000)]), axis=1)
        got = ctx.lookup(q.astype(np.uint16))
        want = bruteforce.quartet_counts_for(trees, ref.names, q)
        assert (got == want).all()
        ctx.batch_free(hb)
        ctx.close()


@pytest.mark.parametrize("n_words", [1, 3, 4, 1023, 1 << 20, (1 << 20) + 7])
def test_sum_words_adds_every_source(eng, n_words):
    """qs_sum_words (the peer-access reduce of the multi-GPU host): dst += sum of up to 15 sources over 32-bit words, for sizes
    with and without a tail below 16 bytes; u16 cells add as packed words."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(n_words)
    ctx = eng.Context(8, 32)
    for k in (0, 1, 2, 7, 15):
        dst = torch.randint(0, 1 << 20, (n_words,), generator=g, dtype=torch.int32).to(dev)
        srcs = [torch.randint(0, 1 << 20, (n_words,), generator=g, dtype=torch.int32).to(dev) for _ in range(k)]
        want = dst.clone()
        for s_ in srcs:
            want += s_
        ctx.sum_words(dst, srcs)
        ctx.sync()
        assert torch.equal(dst, want), (n_words, k)
    with pytest.raises(eng.QSError):
        ctx.sum_words(torch.zeros(4, dtype=torch.int32, device=dev), [torch.zeros(4, dtype=torch.int32, device=dev)] * 16)
    ctx.close()


@pytest.mark.parametrize("seed", range(24))
def test_rooted_reference_soak_against_the_oracle(eng, monkeypatch, seed):
    """Random rooted (degree-2 root) reference trees of 6-45 taxa with random small sets of binary / incomplete / multifurcating
    evaluation trees: LQ-, QP- and EQP-IC of every edge identical to the oracle's (wrap32 and 64-bit QP sums), through a random
    candidate route (two passes, forced single read, scan kernel, one candidate slot). Small tree sets make near-minimal ties --
    and with them both evaluation orders of the root pairs' quartets -- common."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(6, 46))
    m = int(rng.choice([3, 8, 15, 40]))
    kw = [dict(), dict(dropout=0.15), dict(collapse=0.2), dict(collapse=0.15, dropout=0.1)][int(rng.integers(0, 4))]
    ref_nw = synth.random_tree(n, rng, rooted=True)
    trees = synth.tree_set(n, m, 7100 + seed, **kw)
    route = [{_lib.QS_TUNE_SCORE_PASSES: 1}, {_lib.QS_TUNE_SCORE_PASSES: 2}, {_lib.QS_TUNE_SCORE_KERNEL: 1}, {_lib.QS_TUNE_SCORE_CAND_SLOTS: 1}][int(rng.integers(0, 4))]
    for k_, v_ in route.items():
        monkeypatch.setitem(eng.DEFAULT_TUNING, k_, v_)
    o = oracle_counts(ref_nw, trees)
    for exact in (False, True):
        o.score(qp_exact64=exact)
        want = o.scores_by_bipartition()
        got = eng.QuartetScoreComputer(ref_nw, trees, qp_exact64=exact).scores_by_bipartition()
        assert set(got) == set(want)
        for k in got:
            assert got[k] == want[k], (n, m, kw, route, exact, sorted(k), got[k], want[k])


def test_issue_probe_reports_a_plausible_rate(eng):
    """qs_issue_probe: the bare instruction slot of the count kernel (24 v_bitop3 + 4 v_bcnt, registers only, 4 waves per SIMD):
    between 1.0 and 2.5 ns per wave instruction and SIMD on any MI355X (1.39-1.40 on the boxes of profiles/r03_valu_yardstick.txt);
    twice the iterations take twice the time."""
    ctx = eng.Context(8, 32)
    a = ctx.issue_probe(20000)
    b = ctx.issue_probe(40000)
    assert 1.0 < a < 2.5 and 1.0 < b < 2.5, (a, b)
    assert abs(a / b - 1.0) < 0.1, (a, b)
    ctx.close()


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("case", ["F2", "four_modes"])
@pytest.mark.parametrize("split_classes", [False, True])
def test_modes_fixture_on_the_gpu(eng, monkeypatch, case, split_classes, fuse):
    """The committed fixtures of tests/golden/modes.json (F2 of SURVEY 8(c): taxon dropout + collapsed edges; all four kernel
    modes interleaved in one batch): table hash in canonical taxon order and every internal edge's LQ/QP/EQP-IC as hex doubles --
    with the default class floors (the small classes join the most general mode) and with every mode in a class of its own."""
    import hashlib
    import importlib.util
    import json
    import os
    from helpers import remap_table
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = json.load(open(os.path.join(here, "modes.json")))[case]
    spec = importlib.util.spec_from_file_location("make_modes_fixture", os.path.join(here, "make_modes_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    trees = mod.trees_of(fx)
    assert synth.reference_tree(fx["n"], fx["ref_seed"]) == fx["ref"] and trees[0] == fx["first_tree"] and trees[-1] == fx["last_tree"]
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_FUSE_CLASSES, fuse)
    if split_classes:
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    qsc = eng.QuartetScoreComputer(fx["ref"], trees, device=0)
    v = qsc.quartetCounterLookup.ctx.last_count_variant()
    if fuse:     # no mode joins a dearer one: a batch of several modes is "mixed" with any floors, its classes of equal depth bits share a launch
        assert "bitslice_b" in v and (case != "four_modes" or "/fused:" in v), v
    else:
        assert ("gather/mixed/" in v) == split_classes, v
    names = list(qsc.ref.names)
    perm = [names.index(f"t{i}") for i in range(fx["n"])]
    table = remap_table(qsc.quartetCounterLookup.table(), perm).astype("<u4")
    assert int(table.astype(np.uint64).sum()) == fx["checksum"]
    assert hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest() == fx["table_sha256"]
    got = {",".join(sorted(k, key=lambda s: int(s[1:]))): [float(x).hex() for x in v_]
           for k, v_ in qsc.scores_by_bipartition().items()}
    assert got == fx["scores_lq_qp_eqp_hex"]


@pytest.mark.parametrize("count_bits", [32, 16])
@pytest.mark.parametrize("n,bits_wanted", [(24, 4), (40, 5), (68, 6), (132, 7)])
def test_fused_launch_equals_class_by_class_and_the_oracle(eng, monkeypatch, n, bits_wanted, count_bits):
    """count_bitslice3_fused_kernel (qs_count_fused.hip): the classes of a batch that share their depth bits as segments of ONE
    launch -- all four kernel modes interleaved, at 4, 5, 6 and 7 depth bits (ladder + NNI trees set the depth), both cell widths,
    cut into several panel slices (a slice boundary falls inside a class and between classes), on a table shard, accumulated over
    two uploads and with overwrite: the table equals the class-by-class table of round 5 (QS_TUNE_FUSE_CLASSES = 0) and the
    oracle's. The reference's loop is shape-independent (QuartetCounterLookup.hpp:65-106,166-188)."""
    import sys
    sys.setrecursionlimit(100000)
    ref_nw = synth.reference_tree(n, 900 + n)
    ref = flatten.flatten_reference(ref_nw)
    kws = [dict(), dict(dropout=0.15), dict(collapse=0.2), dict(collapse=0.2, dropout=0.1)]
    per = 70 if n <= 64 else 12
    sets = [synth.tree_set(n, per, 910 + n + i, **kw) for i, kw in enumerate(kws)]
    trees = [sets[i % 4][i // 4] for i in range(4 * per)]
    parts = [flatten.flatten_eval_trees(trees, ref.name_to_id)]
    if bits_wanted > 4:   # deep trees of every mode: a caterpillar cut to 2^bits - 2 levels, whole / with a taxon missing / with a collapsed edge
        depth = (1 << bits_wanted) - 2
        lad = f"(t{n - 2},t{n - 1})"
        for i in range(n - 3, n - 3 - (depth - 1), -1):
            lad = f"(t{i},{lad})"
        rest = ",".join(f"t{i}" for i in range(0, n - 3 - (depth - 1) + 1))
        deep_full = f"({rest},{lad});"
        deep = [deep_full, deep_full.replace("t0,", "", 1), deep_full.replace(f"(t{n - 2},t{n - 1})", f"t{n - 2},t{n - 1}", 1)]
        trees += deep
        parts.append(flatten.flatten_eval_trees(deep, ref.name_to_id, recentre=False))
    batch = parts[0]
    for p_ in parts[1:]:
        batch = _concat_batches(batch, p_)
    want = oracle_counts(ref_nw, trees).counts()
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_MIN_TREES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_CLASS_PCT, 0)
    if bits_wanted > 4:
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_DEPTH_CLAMP, 0)          # the deep trees keep their own depth bits
    tables = {}
    for fuse in (1, 0):
        monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_FUSE_CLASSES, fuse)
        ctx, T = gpu_table(eng, ref, batch, count_bits)
        v = ctx.last_count_variant()
        assert ("/fused:" in v) == bool(fuse) and f"bitslice_b{bits_wanted}x2" in v, v
        tables[fuse] = T
        assert (T.astype(np.uint64) == want).all(), (fuse, v)
    # several slices (the panel of a slice holds pieces of up to four classes), accumulate over two uploads, a shard, overwrite
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_FUSE_CLASSES, 1)
    monkeypatch.setitem(eng.DEFAULT_TUNING, _lib.QS_TUNE_PANEL_SLICE_BYTES, 1 << 12)
    ctx, T = gpu_table(eng, ref, batch, count_bits, split=batch.n_trees // 2 + 3)
    assert "/fused:" in ctx.last_count_variant() and (T == tables[1]).all()
    d_lo, d_hi = n // 3, n - 2
    c2 = eng.Context(n, count_bits, d_lo=d_lo, d_hi=d_hi)
    c2.table_alloc()
    c2.count_trees(flatten.flatten_eval_trees(sets[0][:5], ref.name_to_id))           # overwritten below
    c2.count_trees(batch, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    assert "/fused:" in c2.last_count_variant() and c2.trees_counted == batch.n_trees
    assert (c2.table_download().astype(np.uint64) == want[ranks.n_quartets(d_lo): ranks.n_quartets(d_hi)]).all()
    c2.close()
