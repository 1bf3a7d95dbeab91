"""Timing of score pass 1 on the configs[2] table (512 taxa x 10 000 trees) for probe builds of the score kernel (QS_PY_LIB=...):
what the three LDS k log k look-ups of the device QIC cost.   python tools/r06_score_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quartetscores_amd import _lib, engine, flatten, native_ingest

n, m = 512, 10000
ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
ref = flatten.flatten_reference(ref_nw)
batch, _ = native_ingest.ingest_text(ref_nw, native_ingest.synth_trees(n, m, 2001), 0, m, want_ranges=False)
stream = torch.cuda.current_stream()
ctx = engine.Context(n, 32, device=0, stream=stream.cuda_stream)
ctx.table_alloc()
ctx.count_trees(batch, engine.QS_ALGO_GATHER)
for passes, label in ((1, "two plain passes"), (0, "automatic (single read, logging pass 1)")):
    ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, passes)
    best = None
    for _ in range(5):
        try:
            ctx.score(ref)
        except engine.QSError as e:
            print(label, "score failed:", str(e)[:80])
            break
        ph = ctx.last_score_ms()
        if best is None or ph["pass1"] < best["pass1"]:
            best = ph
    if best:
        print(f"{os.path.basename(os.environ.get('QS_PY_LIB', 'product')):34s} {label:42s} pass1 {best['pass1']:.2f} ms  pass2 {best['pass2']:.2f} ms  total {best['total']:.2f} ms  log records {ctx.last_score_log()}")
