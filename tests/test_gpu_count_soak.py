"""Seeded count soak inside the -m gpu suite: every seed draws a taxon count, a tree count, a mix of tree
shapes (binary / missing taxa / collapsed edges / both / all four kernel modes interleaved / rooted / deep, not re-centred /
ladder + NNIs),
the cell width, the launch plan (depth-clamp budget: off, default, forced so that most trees are cut; class floors; fused launches
on / off; panel slice
size: several slices; tile order), a table shard [d_lo, d_hi) and overwrite-vs-accumulate, and compares the WHOLE table (or
shard) the HIP path produces with the oracle's (QuartetCounterLookup.hpp:196-238 restated in oracle/qs_oracle.c). A second set
of seeds does the same for the one-word-per-tuple wire format (QS_COUNT_WIRE16X2) with clamped trees. Sized for <= 60 s in all:
the oracle's work is bounded at ~6e7 (tree, quartet) units per case.
"""
import sys

import numpy as np
import pytest

from oracle_api import Oracle
from quartetscores_amd import _lib, flatten, ranks, synth

pytestmark = pytest.mark.gpu

N_CHOICES = [8, 9, 13, 16, 17, 24, 31, 33, 40, 48, 57, 64, 65, 72, 90, 97, 128, 129, 160]
KINDS = ["binary", "partial", "collapsed", "both", "modes", "modes", "rooted", "deep", "ladder", "ladder"]
ORACLE_UNITS = 6e7


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from quartetscores_amd import engine
    return engine


def draw_trees(rng, n, m, kind, seed):
    if kind == "modes":   # trees of all four kernel modes interleaved in one batch
        kws = [dict(), dict(dropout=0.2), dict(collapse=0.25), dict(collapse=0.2, dropout=0.15)]
        sets = [synth.tree_set(n, (m + 3) // 4, seed + 10 + i, **kw) for i, kw in enumerate(kws)]
        return [sets[i % 4][i // 4] for i in range(m)]
    if kind == "ladder" and n >= 9:   # a caterpillar + NNIs: LCA depths up to n / 2 (re-centred) or n - 2 -> the 6- to 8-bit instances
        lad = f"(t{n - 2},t{n - 1})"
        for i in range(n - 3, -1, -1):
            lad = f"(t{i},{lad})"
        return [lad + ";"] + (list(synth.nni_tree_set(lad + ";", m - 1, seed + 1)) if m > 1 else [])
    kw = {"partial": dict(dropout=0.2), "collapsed": dict(collapse=0.25), "both": dict(collapse=0.2, dropout=0.15),
          "rooted": dict(rooted=True, dropout=float(rng.choice([0.0, 0.1])))}.get(kind, {})
    return synth.tree_set(n, m, seed + 2, **kw)


def draw_case(seed, binary_full_only=False):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice(N_CHOICES))
    kind = "deep" if binary_full_only and rng.random() < 0.6 else ("binary" if binary_full_only else str(rng.choice(KINDS)))
    m = int(rng.choice([1, 31, 32, 33, 64, 100, 200, 700]))
    m = max(2 if n > 100 else 1, min(m, int(ORACLE_UNITS // ranks.n_quartets(n))))
    sub = int(rng.integers(1, 1 << 30))
    ref_nw = synth.reference_tree(n, sub)
    trees = draw_trees(rng, n, m, kind, sub)
    # not re-centred: LCA depths grow with the tree's height -> several depth classes, the depth clamp, the deep instances
    recentre = kind != "deep" and bool(rng.random() < 0.6)
    if kind == "ladder":
        sys.setrecursionlimit(100000)
    tuning = {}
    clamp = int(rng.choice([-1, 0, 1000000, 1000000]))        # -1: the library's default budget
    if clamp >= 0:
        tuning[_lib.QS_TUNE_DEPTH_CLAMP] = clamp
    tuning[_lib.QS_TUNE_CLASS_MIN_TREES] = int(rng.choice([1, 8, 1024]))
    tuning[_lib.QS_TUNE_CLASS_PCT] = int(rng.choice([0, 10, 60]))
    sl = int(rng.choice([0, 1 << 12, 1 << 16, 1 << 20]))
    if sl:
        tuning[_lib.QS_TUNE_PANEL_SLICE_BYTES] = sl
    if rng.random() < 0.3:
        tuning[_lib.QS_TUNE_FUSE_CLASSES] = 0                 # one launch per class (round 5) instead of one per depth-bits group
    if rng.random() < 0.5:
        tuning[_lib.QS_TUNE_TILE_ORDER] = int(rng.choice([0, 1 | 4 << 16, 2 | 32 << 16, 4 | 16 << 16, 3 | 7 << 16]))
    d_lo, d_hi = 0, n
    if n >= 9 and rng.random() < 0.4:
        d_lo = int(rng.integers(0, n - 4))
        d_hi = int(rng.integers(max(d_lo + 1, 4), n + 1))
    return dict(n=n, m=m, kind=kind, ref_nw=ref_nw, trees=trees, recentre=recentre, tuning=tuning, d_lo=d_lo, d_hi=d_hi,
                bits=int(rng.choice([16, 32])), overwrite=bool(rng.random() < 0.5), rng=rng)


def oracle_rows(case):
    o = Oracle(case["ref_nw"])
    o.count("\n".join(case["trees"]), nthreads=8)
    want = o.counts()
    o.close()
    return want[ranks.n_quartets(case["d_lo"]): ranks.n_quartets(case["d_hi"])]


@pytest.mark.parametrize("seed", range(32))
def test_count_soak_whole_table_vs_oracle(eng, seed):
    case = draw_case(seed)
    n, m = case["n"], case["m"]
    ref = flatten.flatten_reference(case["ref_nw"])
    batch = flatten.flatten_eval_trees(case["trees"], ref.name_to_id, recentre=case["recentre"])
    ctx = eng.Context(n, case["bits"], d_lo=case["d_lo"], d_hi=case["d_hi"])
    for k_, v_ in case["tuning"].items():
        ctx.set_tuning(k_, v_)
    ctx.table_alloc()
    if case["overwrite"]:
        # whatever the table held before is discarded by the first slice's stores
        junk = flatten.flatten_eval_trees(synth.tree_set(n, 3, 99 + seed), ref.name_to_id)
        ctx.count_trees(junk)
        ctx.count_trees(batch, eng.QS_ALGO_GATHER | eng.QS_COUNT_OVERWRITE)
        assert ctx.trees_counted == m
    else:
        cut = max(1, m // 2)
        ctx.count_trees(batch.slice(0, cut))
        if cut < m:
            ctx.count_trees(batch.slice(cut, m))                 # accumulate
    got = ctx.table_download().astype(np.uint64)
    variant = ctx.last_count_variant()
    ctx.close()
    want = oracle_rows(case)
    if case["bits"] == 16:
        want = want % (1 << 16)
    assert got.shape == want.shape, (seed, case["kind"], variant)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, (seed, n, m, case["kind"], case["recentre"], case["tuning"], (case["d_lo"], case["d_hi"]), variant, bad[:5], got[bad[:3]], want[bad[:3]])


@pytest.mark.parametrize("seed", range(100, 108))
def test_count_soak_wire_format_with_clamped_trees(eng, seed):
    """QS_COUNT_WIRE16X2: binary trees holding all taxa counted straight into one word n0 | n1 << 16 per tuple; the depth clamp's
    corrections go to the wire words too. Unpacked (n2 = trees - n0 - n1) it must be the oracle's table."""
    import torch
    case = draw_case(seed, binary_full_only=True)
    n, m = case["n"], case["m"]
    ref = flatten.flatten_reference(case["ref_nw"])
    batch = flatten.flatten_eval_trees(case["trees"], ref.name_to_id, recentre=case["recentre"])
    dev = torch.device("cuda", 0)
    ctx = eng.Context(n, 32, d_lo=case["d_lo"], d_hi=case["d_hi"])
    for k_, v_ in case["tuning"].items():
        ctx.set_tuning(k_, v_)
    nt = ctx.table_tuples
    wire = torch.full((max(nt, 1),), 0x7FFFFFFF, dtype=torch.int32, device=dev)       # garbage: the first count overwrites
    ctx.wire_attach(wire)
    cut = max(1, m // 2) if not case["overwrite"] else m
    hb = ctx.batch_upload(batch.slice(0, cut), with_nodes=False)
    ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_WIRE16X2 | eng.QS_COUNT_OVERWRITE)
    ctx.sync()
    ctx.batch_free(hb)
    if cut < m:
        hb = ctx.batch_upload(batch.slice(cut, m), with_nodes=False)
        ctx.count_batch(hb, eng.QS_ALGO_GATHER | eng.QS_COUNT_WIRE16X2)               # accumulate into the wire words
        ctx.sync()
        ctx.batch_free(hb)
    variant = ctx.last_count_variant()
    out = torch.zeros(max((nt * 6 + 3) // 4, 1), dtype=torch.int32, device=dev)
    ctx.unpack16x2(wire, nt, m, out)
    ctx.sync()
    got = out.cpu().numpy().view(np.uint16)[: nt * 3].reshape(-1, 3).astype(np.uint64)
    ctx.close()
    want = oracle_rows(case)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, (seed, n, m, case["kind"], case["tuning"], (case["d_lo"], case["d_hi"]), variant, bad[:5])
