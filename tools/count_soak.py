"""Randomised soak of the count kernels: the bit-sliced kernel (default launch plan, and random slice sizes / tile orders /
depth-class thresholds) against the byte-SWAR kernel on the same trees -- whole tables and random shards [d_lo, d_hi), both cell
widths, binary / multifurcating / partial / ladder-like trees, re-centred or not, accumulate across two batches, the depth clamp forced
on every tree it can take (round 5) -- and against the split-based
brute force (tests/bruteforce.py) on small cases.      python tools/count_soak.py [cases] [seed]"""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.setrecursionlimit(100000)
import bruteforce
from quartetscores_amd import _lib, engine, flatten, ranks, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
clamped = 0
for case in range(cases):
    n = int(rng.choice([4, 5, 7, 9, 16, 17, 24, 33, 40, 64, 65, 97, 130, 200, 257]))
    kind = str(rng.choice(["binary", "binary", "collapsed", "partial", "mixed", "ladder", "modes", "modes", "rooted"]))
    m = int(rng.choice([1, 31, 33, 200, 1500])) if n <= 130 else int(rng.choice([40, 300]))
    bits = int(rng.choice([16, 32]))
    seed = int(rng.integers(1, 1 << 30))
    ref_nw = synth.reference_tree(n, seed)
    if kind == "ladder" and n >= 9:
        lad = f"(t{n - 2},t{n - 1})"
        for i in range(n - 3, -1, -1):
            lad = f"(t{i},{lad})"
        trees = [lad + ";"] + list(synth.nni_tree_set(lad + ";", m - 1, seed + 1)) if m > 1 else [lad + ";"]
    elif kind == "modes":      # trees of all four kernel modes interleaved in one batch (round 4: the mode is a property of the tree)
        kws = [dict(), dict(dropout=0.2), dict(collapse=0.25), dict(collapse=0.2, dropout=0.15)]
        sets = [synth.tree_set(n, (m + 3) // 4, seed + 10 + i, **kw) for i, kw in enumerate(kws)]
        trees = [sets[i % 4][i // 4] for i in range(m)]
    else:
        kw = {"collapsed": dict(collapse=0.25), "partial": dict(dropout=0.2), "mixed": dict(collapse=0.2, dropout=0.15),
              "rooted": dict(rooted=True, dropout=float(rng.choice([0.0, 0.1])))}.get(kind, {})   # degree-2 root in the evaluation trees
        trees = synth.tree_set(n, m, seed + 2, **kw)
    ref = flatten.flatten_reference(ref_nw)
    recentre = bool(rng.random() < 0.65)             # un-centred trees are deep at moderate sizes: depth classes, the depth clamp
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id, recentre=recentre)
    half = batch.slice(0, max(1, m // 2)), batch.slice(max(1, m // 2), m)
    d_lo, d_hi = 0, n
    if n >= 9 and rng.random() < 0.4:
        d_lo = int(rng.integers(0, n - 4)); d_hi = int(rng.integers(max(d_lo + 1, 4), n + 1))
    tables = {}
    for name, tuning in (("swar", {_lib.QS_TUNE_GATHER_IMPL: _lib.QS_IMPL_SWAR}), ("default", {}),
                         ("random plan", {_lib.QS_TUNE_PANEL_SLICE_BYTES: int(rng.choice([1 << 12, 1 << 16, 1 << 20, 1 << 24])),
                                          _lib.QS_TUNE_TILE_ORDER: int(rng.choice([0, 1 | 4 << 16, 2 | 32 << 16, 4 | 16 << 16, 3 | 7 << 16])),
                                          _lib.QS_TUNE_CLASS_PCT: int(rng.choice([0, 10, 60, 100])),
                                          _lib.QS_TUNE_DEPTH_CLAMP: int(rng.choice([0, 20, 5000, 1000000])),
                                          _lib.QS_TUNE_FUSE_CLASSES: int(rng.choice([0, 1])),      # round 6: one launch per class / per depth-bits group
                                          _lib.QS_TUNE_CLASS_MIN_TREES: int(rng.choice([1, 8, 64, 1024]))}),
                         # round 5: every tree in the lowest class any budget allows -- the correction kernel on every shape
                         ("clamp", {_lib.QS_TUNE_DEPTH_CLAMP: 1000000, _lib.QS_TUNE_CLASS_MIN_TREES: int(rng.choice([1, 1024])),
                                    _lib.QS_TUNE_PANEL_SLICE_BYTES: int(rng.choice([0, 1 << 14, 1 << 20]))})):
        ctx = engine.Context(n, bits, d_lo=d_lo, d_hi=d_hi)
        for k_, v_ in tuning.items():
            ctx.set_tuning(k_, v_)
        ctx.table_alloc()
        ctx.count_trees(half[0])
        if half[1].n_trees:
            ctx.count_trees(half[1])                 # accumulate
        tables[name] = ctx.table_download().astype(np.uint64)
        variant = ctx.last_count_variant()
        clamped += name == "clamp" and "/clamp:" in variant
        ctx.close()
    ok = all(np.array_equal(tables["swar"], tables[k]) for k in ("default", "random plan", "clamp"))
    why = "" if ok else " [bit-sliced != SWAR]"
    if n <= 33 and m <= 200:                         # the split-based brute force on every quartet of the table
        want = bruteforce.count_table(list(ref.names), trees)
        r0, r1 = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)
        got = tables["default"].reshape(-1, 3)
        if bits == 16:
            want = want % (1 << 16)
        if not np.array_equal(got, want[r0:r1]):
            ok = False; why += " [!= brute force]"
    bad += not ok
    print(f"case {case}: n={n} m={m} u{bits} {kind}{'' if recentre else ' (not re-centred)'} d[{d_lo},{d_hi}) seed={seed} {variant}: {'ok' if ok else 'MISMATCH' + why}", flush=True)
print("mismatches:", bad, "| cases with trees below their own depth bits:", clamped)
sys.exit(1 if bad else 0)
