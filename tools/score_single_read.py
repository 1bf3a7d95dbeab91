"""qs_score: two passes against the single-read mode with different pre-pass samples (QS_TUNE_SCORE_SAMPLE); run on a GPU box:
    python tools/score_single_read.py [taxa:trees[:nni] ...]      default 512:10000 512:10000:1 256:12500
Prints per variant the best of 4 calls: whole call, pass 1 (incl. the pre-pass), pass 2 / log filter, log records, and
whether the scores equal the two-pass ones bit for bit."""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import _lib, engine, flatten, native_ingest
if os.environ.get('QS_LIB'):   # kernel experiments: another build of the library (tools/Makefile exp)
    _lib.LIB_PATH = os.path.abspath(os.environ['QS_LIB'])

cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(512, 10000), (512, 10000, 1), (256, 12500)]
for case in cases:
    n, m = case[:2]
    nni = len(case) > 2 and case[2]
    ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
    text = native_ingest.synth_trees(n, m, 2001, kind="nni" if nni else "random", ref_text=ref_nw if nni else None)
    ref = flatten.flatten_reference(ref_nw)
    batch, _ = native_ingest.ingest_text(ref_nw, text, want_ranges=False)
    ctx = engine.Context(n, 32)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb)
    ctx.sync()
    base = None
    for name, passes, sample in (("two passes", 1, 0), ("automatic (default)", 0, 64 | 65536), ("automatic, no tie filter", 0, 64 | 65536 | (1 << 30)), ("single, no pre-pass", 2, 0), ("single, chunk 1/8", 2, 8), ("single, chunk 1/16", 2, 16),
                                 ("single, chunk 1/32", 2, 32), ("single, chunk 1/64", 2, 64), ("single, round 1/16", 2, 16 | 65536),
                                 ("single, round 1/32", 2, 32 | 65536), ("single, round 1/64", 2, 64 | 65536)):
        ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, passes)
        ctx.set_tuning(_lib.QS_TUNE_SCORE_DEDUPE, 0 if sample >> 30 else 1)
        sample &= (1 << 30) - 1
        ctx.set_tuning(_lib.QS_TUNE_SCORE_SAMPLE, sample)
        best = None
        for _ in range(4):
            sc = ctx.score(ref)
            ms = list(ctx.last_score_ms().values())
            if best is None or ms[0] < best[0]:
                best = list(ms)
        if base is None:
            base = sc
        same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(base[:3], sc[:3]))
        print(f"n={n} m={m}{' nni' if nni else ''} {name:26s}: qs_score {best[0]:7.3f} ms | pass 1 {best[2]:7.3f} then {best[3]:7.3f} | wait+d2h {best[4]:6.3f} finish {best[5]:6.3f} | log {ctx.last_score_log():9d} records (predicted {ctx.last_score_estimate():9d}) | {'same scores' if same else 'SCORES DIFFER'}", flush=True)
    ctx.close()
