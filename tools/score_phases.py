"""Wall time of the phases of qs_score (pass 1, pass 2, copies, host finalisation) at 128 and 256 taxa; run on a GPU box."""
import sys
import time

import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import engine, flatten, synth
cases = [(int(a), 64) for a in sys.argv[1:]] or [(128, 1000), (256, 2000)]
for n, m in cases:
    ref_nw = synth.reference_tree(n, 2000)
    trees = synth.tree_set(n, m, 2001)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    ctx = engine.Context(n, 32)
    ctx.table_alloc()
    ctx.count_trees(batch)
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
    print(f"n={n}: qs_score {ms_all:.3f} ms | pass1 call {ms1:.3f} pass2 call {ms2:.3f} d2h {msd:.3f} finish {ms3:.3f}  P={P}")
