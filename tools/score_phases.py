"""Wall time of the phases of qs_score (pass 1, pass 2, copies, host finalisation); run on a GPU box:
    python tools/score_phases.py [taxa:trees[:count_bits[:d_lo:d_hi]] ...]      default 128:1000 256:2000 512:10000"""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import _lib, engine, flatten, native_ingest
if os.environ.get('QS_LIB'):   # kernel experiments: another build of the library (tools/Makefile exp)
    _lib.LIB_PATH = os.path.abspath(os.environ['QS_LIB'])
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(128, 1000), (256, 2000), (512, 10000)]
for case in cases:
    n, m = case[:2]
    bits = case[2] if len(case) > 2 else 32
    d_lo, d_hi = (case[3], case[4]) if len(case) > 4 else (0, n)
    ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
    ref = flatten.flatten_reference(ref_nw)
    batch, _ = native_ingest.ingest_text(ref_nw, native_ingest.synth_trees(n, m, 2001), want_ranges=False)
    ctx = engine.Context(n, bits, d_lo=d_lo, d_hi=d_hi)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb)
    ctx.sync()

    def t(f, reps=5):
        torch.cuda.synchronize(); best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3, r
    P = ctx.score_pair_slots(ref)
    seen = None
    for kernel in (1, 0):   # 1 = scan kernel, 0 = chunk kernel (default)
        ctx.set_tuning(_lib.QS_TUNE_SCORE_KERNEL, kernel)
        shard = (d_lo, d_hi) != (0, n)        # a table shard is scored in steps (qs_score refuses)
        ms_all, sc = (float('nan'), None) if shard else t(lambda: ctx.score(ref))
        sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
        cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
        ms1, _ = t(lambda: ctx.score_pass1(ref, sums, mins))
        ms2, _ = t(lambda: ctx.score_pass2(ref, mins, cand))
        sh, ch = sums.cpu().numpy(), cand.cpu().numpy()[None, :]
        ms3, _ = t(lambda: ctx.score_finish(ref, sh, ch))
        msd, _ = t(lambda: (sums.cpu(), cand.cpu()))
        gb = ctx.table_bytes / 1e9
        fin = ctx.score_finish(ref, sh, ch)
        same = "" if seen is None else ("  same scores" if all((x == y).all() for x, y in zip(seen[:3], fin[:3])) and (seen_s == sh).all() else "  SCORES DIFFER")
        sc = fin
        seen, seen_s = sc, sh
        print(f"n={n} m={m} u{bits} d[{d_lo},{d_hi}) kernel={'scan' if kernel else 'bundle'}: qs_score {ms_all:.3f} ms | pass1 {ms1:.3f} ({gb / ms1:.2f} TB/s) pass2 {ms2:.3f} ({gb / ms2:.2f} TB/s) d2h {msd:.3f} finish {ms3:.3f}  P={P} table {gb:.2f} GB{same}", flush=True)
    ctx.close()
