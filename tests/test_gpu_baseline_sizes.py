"""The HIP path at BASELINE.json's FULL sizes (configs[2], configs[3]'s per-GPU share, one table shard of configs[4]).

At these sizes the oracle cannot produce the whole table in bounded time (one 512-taxon tree is 5.7e9 increments of
QuartetCounterLookup.hpp:73-105), so parity goes through
  * size-independent properties on the device: every tuple of a batch of binary trees holding all taxa sums to the
    number of trees; the bit-sliced kernel's table equals the byte-SWAR kernel's table bit for bit;
  * qs_lookup (= countQuartetOccurrences, QuartetCounterLookup.hpp:299-318) of 10^5 random quartets against the
    independent split-based brute force (tests/bruteforce.py) on a 64-tree prefix of the same trees;
  * at 256 taxa additionally the oracle itself (fast n^4 table, 8-tree prefix): counts bit-exact, LQ-/QP-/EQP-IC identical
    (QuartetScoreComputer.hpp:379-490) -- the one test that reaches the XCD-remapped tile order (n >= 200) and the
    64-iteration variant of score pass 1 with an exact answer.
Trees: csrc/host/synth.hpp (seed = 1000 * config + set id, as bench.py). Run on the GPU box: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

import bruteforce
from oracle_api import Oracle
from quartetscores_amd import _lib, distributed, flatten, native_ingest, ranks, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from quartetscores_amd import engine
    return engine


def workload(cfg_no, n, m):
    ref_nw = native_ingest.synth_trees(n, 1, 1000 * cfg_no).decode().strip()
    text = native_ingest.synth_trees(n, m, 1000 * cfg_no + 1)
    ref = flatten.flatten_reference(ref_nw)
    batch, total = native_ingest.ingest_text(ref_nw, text, want_ranges=False)
    assert total == m and batch.n_trees == m
    return ref_nw, ref, text, batch


def tuple_sums_equal(torch, table, nq, bits, m):
    t16 = table.view(torch.int16)
    for lo in range(0, nq, 1 << 26):
        hi = min(nq, lo + (1 << 26))
        cells = table[lo * 3: hi * 3] if bits == 32 else (t16[lo * 3: hi * 3].to(torch.int32) & 0xFFFF)
        if not bool((cells.view(hi - lo, 3).sum(dim=1) == m).all().item()):
            return False
    return True


def check_full_size(eng, cfg_no, n, m, bits, d_lo=0, d_hi=0, expect_slices=True, slice_bytes=0):
    """Property gates + brute-force lookups for one (shard of a) table at full size; returns (ctx, table, extras)."""
    import torch
    ref_nw, ref, text, batch = workload(cfg_no, n, m)
    d_hi = d_hi or n
    ctx = eng.Context(n, bits, d_lo=d_lo, d_hi=d_hi)
    nq = ctx.table_tuples
    assert nq == ranks.n_quartets(d_hi) - ranks.n_quartets(d_lo)
    table = torch.zeros((ctx.table_bytes + 3) // 4, dtype=torch.int32, device="cuda")
    ctx.table_attach(table)
    if slice_bytes:     # (a workload whose one depth class fits one default slice: keep the read-modify-write path in the test)
        ctx.set_tuning(_lib.QS_TUNE_PANEL_SLICE_BYTES, slice_bytes)
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE | eng.QS_COUNT_TIMED)
    ctx.sync()
    variant = ctx.last_count_variant()
    ctx.first_variant, ctx.first_launches = variant, ctx.last_count_launches()
    assert "binary_full/bitslice" in variant, variant
    if expect_slices:   # the default panel-slice size is in force: several slices, i.e. the table read-modify-write path
        assert ctx.last_count_launches() > 1
    assert ctx.trees_counted == m
    assert tuple_sums_equal(torch, table, nq, bits, m)
    # the independent byte-SWAR kernel gives the same table, bit for bit
    mine = table.clone()
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    assert "depth_u" in ctx.last_count_variant()
    assert torch.equal(mine, table)
    del mine
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_AUTO)
    # accumulate mode on top of the existing table: exactly twice the counts (checks the RMW of every slice)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER)
    ctx.sync()
    if 2 * m < (1 << bits):
        assert tuple_sums_equal(torch, table, nq, bits, 2 * m)
    # 10^5 random quartets of a 64-tree prefix against the split-based brute force
    k = 64
    hb_prefix = ctx.batch_upload(batch.slice(0, k), with_nodes=False)
    ctx.count_batch(hb_prefix, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    rng = np.random.default_rng(1000 * cfg_no + 7)
    # largest id uniform over the shard's range, the other three below it (every 4-set of the shard can occur)
    dd = rng.integers(max(d_lo, 3), d_hi, size=100000)
    q = np.sort(np.stack([np.append(rng.choice(d, size=3, replace=False), d) for d in dd]), axis=1)
    # shuffle the order inside each 4-tuple: countQuartetOccurrences takes ids in any order
    perm = np.stack([rng.permutation(4) for _ in range(len(q))])
    qp = np.take_along_axis(q, perm, axis=1)
    got = ctx.lookup(qp.astype(np.uint16))
    lines = [ln.decode() for ln in text.split(b"\n")[:k]]
    want = bruteforce.quartet_counts_for(lines, ref.names, qp)
    assert (got == want).all()
    assert (got.sum(axis=1) == k).all()
    ctx.batch_free(hb_prefix)
    ctx.batch_free(hb)
    return ctx, table, (ref_nw, ref, text, batch)


def score_steps(ctx, ref, kernel):
    """Score passes 1 and 2 with the bundle kernel (0) or the scan kernel (1): per-node-pair sums, candidate slots."""
    import torch
    ctx.set_tuning(_lib.QS_TUNE_SCORE_KERNEL, kernel)
    P = ctx.score_pair_slots(ref)
    sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
    cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
    ctx.score_pass1(ref, sums, mins)
    ctx.score_pass2(ref, mins, cand)
    extra = ctx.score_overflow(ref, mins, cand)
    sh, ch = sums.cpu().numpy(), cand.cpu().numpy()
    return sh, ctx.score_finish(ref, sh, ch[None, :], extra=extra)


def test_configs2_512_taxa_10000_trees_u32(eng):
    """BASELINE configs[2]: 34 GB table. Since round 5 the depth clamp puts all 10 000 trees into the 4-bit class (7572 of them need
    5 bits: clamp_fix_kernel adds 1.4e8 tied quartets) and the class is ONE slice = one launch; the accumulate run inside
    check_full_size keeps the read-modify-write path under the gates. Scoring at full size: the bundle kernel (43 000 planned rounds) and the scan
    kernel give the same per-node-pair sums bit for bit and the same LQ-/QP-/EQP-IC."""
    ctx, table, (ref_nw, ref, text, batch) = check_full_size(eng, 2, 512, 10000, 32, expect_slices=False)
    assert "/clamp:" in ctx.first_variant and "bitslice_b5" not in ctx.first_variant and ctx.first_launches == 1, ctx.first_variant
    assert ctx.table_bytes == 33958525440
    a, b = score_steps(ctx, ref, 0), score_steps(ctx, ref, 1)
    assert (a[0] == b[0]).all()
    for x, y in zip(a[1][:3], b[1][:3]):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.isfinite(a[1][0][1:]).sum() == 512 - 3          # every internode of the binary reference got its LQ-IC
    ctx.close()


def test_configs3_share_256_taxa_12500_trees_u32_with_oracle(eng):
    """One GPU's share of BASELINE configs[3] + the oracle on an 8-tree prefix: counts bit-exact, scores identical."""
    n, m = 256, 12500
    # (with the depth clamp all 12500 trees share the 4-bit class = one default slice: two slices keep the accumulate path in the test)
    ctx, table, (ref_nw, ref, text, batch) = check_full_size(eng, 3, n, m, 32, slice_bytes=120_000_000)
    assert ctx.table_bytes == 2097511680
    k = 8
    lines = b"\n".join(text.split(b"\n")[:k]).decode()
    hb = ctx.batch_upload(batch.slice(0, k), with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    o = Oracle(ref_nw)
    o.count(lines, savemem=False, cint_bits=16, nthreads=32)
    assert o.names == ref.names
    T = ctx.table_download()
    want = o.counts()
    assert T.shape == want.shape and (T == want).all()
    del T, want
    lq, qp, eqp, bif = ctx.score(ref)
    assert bif
    o.score(nthreads=1)
    # oracle edges are keyed by bipartition; compare through the canonical keys
    from quartetscores_amd import newick
    mine = {}
    names = ref.names
    for e in range(ref.n_nodes - 1):
        node = ref.nodes[e + 1]
        below = frozenset(x.name for x in newick.preorder(node) if x.is_leaf)
        if len(below) <= 1 or len(below) >= n - 1:
            continue
        other = frozenset(names) - below
        key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
        mine[key] = (lq[e + 1], qp[e + 1], eqp[e + 1])
    theirs = o.scores_by_bipartition()
    assert set(mine) == set(theirs)
    for key in mine:
        assert mine[key] == theirs[key], (sorted(key)[:4], mine[key], theirs[key])
    o.close()
    ctx.batch_free(hb)
    ctx.close()


def test_configs4_shard_1024_taxa_5000_trees_u16(eng):
    """One of the 8 table shards of BASELINE configs[4] (34 GB of u16 cells; every GPU counts all 5000 trees)."""
    n, m = 1024, 5000
    d_lo, d_hi = distributed.shard_of_largest_id(n, 8, 3)
    ctx, table, (ref_nw, ref, text, batch) = check_full_size(eng, 4, n, m, 16, d_lo, d_hi, slice_bytes=900_000_000)
    assert 33e9 < ctx.table_bytes < 35e9
    # the shard's part of the scoring (u16 cells, a plan restricted to d in [d_lo, d_hi)): both kernels, same sums
    a, b = score_steps(ctx, ref, 0), score_steps(ctx, ref, 1)
    assert (a[0] == b[0]).all() and a[0].any()
    ctx.close()


@pytest.mark.parametrize("kind", ["dropout", "collapse", "mixed"])
def test_512_taxa_gene_tree_like_batches(eng, kind):
    """The round-3 verdict's workloads at BASELINE configs[2]'s taxon count (34 GB table): 900 trees with 10 % of the taxa
    missing (binary_partial), with 20 % of the internal edges collapsed (general_full on the two-column tile), or a third each
    of full / incomplete / collapsed trees (classes of several modes in one batch). The table is too large for the oracle:
    the bit-sliced table must equal the byte-SWAR kernel's bit for bit (which takes the whole batch in ONE mode), every
    tuple sums to at most the number of trees, and 20 000 random quartets equal the split-based brute force."""
    import torch
    n, m = 512, 900
    ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
    ref = flatten.flatten_reference(ref_nw)
    if kind == "mixed":
        parts = [synth.tree_set(n, 300, 2101), synth.tree_set(n, 300, 2102, dropout=0.1), synth.tree_set(n, 300, 2103, collapse=0.2)]
        trees = [parts[i % 3][i // 3] for i in range(m)]
    else:
        trees = synth.tree_set(n, m, 2100, **({"dropout": 0.1} if kind == "dropout" else {"collapse": 0.2}))
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(n, 32)
    table = torch.zeros((ctx.table_bytes + 3) // 4, dtype=torch.int32, device="cuda")
    ctx.table_attach(table)
    if kind == "mixed":       # 300 trees per mode: let them form their own classes
        ctx.set_tuning(_lib.QS_TUNE_CLASS_MIN_TREES, 64)
        ctx.set_tuning(_lib.QS_TUNE_CLASS_PCT, 0)
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    v = ctx.last_count_variant()
    assert {"dropout": "gather/binary_partial/", "collapse": "gather/general_full/", "mixed": "gather/mixed/"}[kind] in v and "depth_u" not in v, v
    nq = ctx.table_tuples
    for lo in range(0, nq, 1 << 26):
        hi = min(nq, lo + (1 << 26))
        sums = table[lo * 3: hi * 3].view(hi - lo, 3).sum(dim=1)
        assert int(sums.max().item()) <= m
    mine = table.clone()
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    assert "depth_u" in ctx.last_count_variant()
    assert torch.equal(mine, table)
    del mine
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_AUTO)
    # random quartets of a 48-tree prefix against the brute force
    k = 48
    hb2 = ctx.batch_upload(batch.slice(0, k), with_nodes=False)
    ctx.count_batch(hb2, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    rng = np.random.default_rng(2107)
    q = np.sort(np.stack([rng.choice(n, size=4, replace=False) for _ in range(20000)]), axis=1)
    got = ctx.lookup(q.astype(np.uint16))
    want = bruteforce.quartet_counts_for(trees[:k], ref.names, q)
    assert (got == want).all()
    ctx.batch_free(hb2)
    ctx.batch_free(hb)
    ctx.close()


def test_fused_launch_at_the_size_of_configs2(eng):
    """count_bitslice3_fused_kernel on the 34 GB table of configs[2] (512 taxa; 3.1 M wave tiles, XCD-remapped launch order, the per-step
    barrier on for three segments and off for general_full): 360 trees of all four kernel modes interleaved -- one launch of four segments
    -- give the table of the class-by-class launches (QS_TUNE_FUSE_CLASSES = 0) and of the independent byte-SWAR kernel, bit for bit; 5000 random qs_lookups equal the split-based brute force.
    The reference's loop is shape-independent (QuartetCounterLookup.hpp:65-106,166-188)."""
    import torch
    n = 512
    ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
    ref = flatten.flatten_reference(ref_nw)
    kws = [dict(), dict(dropout=0.1), dict(collapse=0.2), dict(collapse=0.2, dropout=0.1)]
    sets = [synth.tree_set(n, 90, 2600 + i, **kw) for i, kw in enumerate(kws)]
    trees = [sets[i % 4][i // 4] for i in range(360)]
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = eng.Context(n, 32)
    table = torch.zeros((ctx.table_bytes + 3) // 4, dtype=torch.int32, device="cuda")
    ctx.table_attach(table)
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE | eng.QS_COUNT_TIMED)
    ctx.sync()
    v = ctx.last_count_variant()
    assert "/fused:1" in v and ctx.last_count_launches() == 1 and v.count("bitslice_b4x2") == 4, v
    fused = table.clone()
    ctx.set_tuning(_lib.QS_TUNE_FUSE_CLASSES, 0)
    hb0 = ctx.batch_upload(batch, with_nodes=False)                      # (the class plan is made at upload time)
    ctx.count_batch(hb0, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE | eng.QS_COUNT_TIMED)
    ctx.sync()
    assert "/fused" not in ctx.last_count_variant() and ctx.last_count_launches() >= 1
    assert torch.equal(fused, table)
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_SWAR)
    ctx.count_batch(hb0, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    assert "depth_u" in ctx.last_count_variant()
    assert torch.equal(fused, table)
    ctx.set_tuning(_lib.QS_TUNE_GATHER_IMPL, _lib.QS_IMPL_AUTO)
    del fused
    # random quartets against the split-based brute force
    rng = np.random.default_rng(26)
    qs_ = np.sort(np.stack([rng.choice(n, size=4, replace=False) for _ in range(5000)]), axis=1).astype(np.uint16)
    got = ctx.lookup(qs_)
    want = bruteforce.quartet_counts_for(trees, ref.names, qs_.astype(np.int64))
    assert (got == want).all()
    ctx.batch_free(hb)
    ctx.batch_free(hb0)
    ctx.close()
