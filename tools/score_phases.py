"""Wall time of the phases of qs_score (pass 1, pass 2, copies, host finalisation); run on a GPU box:
    python tools/score_phases.py [taxa:trees ...]      default 128:1000 256:2000 512:10000"""
import os
import sys
import time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import engine, flatten, native_ingest
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(128, 1000), (256, 2000), (512, 10000)]
for n, m in cases:
    ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
    ref = flatten.flatten_reference(ref_nw)
    batch, _ = native_ingest.ingest_text(ref_nw, native_ingest.synth_trees(n, m, 2001), want_ranges=False)
    ctx = engine.Context(n, 32)
    ctx.table_alloc()
    hb = ctx.batch_upload(batch, with_nodes=False)
    ctx.count_batch(hb)
    ctx.sync()

    def t(f, reps=5):
        torch.cuda.synchronize(); best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3, r
    ms_all, _ = t(lambda: ctx.score(ref))
    P = ctx.score_pair_slots(ref)
    sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
    cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
    ms1, _ = t(lambda: ctx.score_pass1(ref, sums, mins))
    ms2, _ = t(lambda: ctx.score_pass2(ref, mins, cand))
    sh, ch = sums.cpu().numpy(), cand.cpu().numpy()[None, :]
    ms3, _ = t(lambda: ctx.score_finish(ref, sh, ch))
    msd, _ = t(lambda: (sums.cpu(), cand.cpu()))
    gb = ctx.table_bytes / 1e9
    print(f"n={n} m={m}: qs_score {ms_all:.3f} ms | pass1 {ms1:.3f} ({gb / ms1:.2f} TB/s) pass2 {ms2:.3f} ({gb / ms2:.2f} TB/s) d2h {msd:.3f} finish {ms3:.3f}  P={P} table {gb:.2f} GB")
    ctx.close()
