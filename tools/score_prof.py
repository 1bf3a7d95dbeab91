"""Workload for profiling the score kernels: count once, then run score pass 1 and pass 2 a few times.
    python tools/score_prof.py <taxa> <trees> <kernel 0|1> [reps] [QS_TUNE_SCORE_PASSES: 1 = two passes, 2 = single read (default)]
In the single-read mode every repetition launches score_bundle_kernel<.., 1, ..> twice (the minima-only pre-pass over one
round in 64, then the logging pass) and score_log_kernel once: per-dispatch averages of the profiler mix the two."""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import _lib, engine, flatten, native_ingest
if os.environ.get('QS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['QS_LIB'])
n, m, kernel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ref_nw = native_ingest.synth_trees(n, 1, 2000).decode().strip()
ref = flatten.flatten_reference(ref_nw)
batch, _ = native_ingest.ingest_text(ref_nw, native_ingest.synth_trees(n, m, 2001), want_ranges=False)
ctx = engine.Context(n, 32)
ctx.table_alloc()
ctx.count_batch(ctx.batch_upload(batch, with_nodes=False))
ctx.set_tuning(_lib.QS_TUNE_SCORE_KERNEL, kernel)
ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, passes)
P = ctx.score_pair_slots(ref)
sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
cand = torch.empty(8 * P, dtype=torch.int64, device="cuda")
for _ in range(reps):
    ctx.score_pass1(ref, sums, mins)
    ctx.score_pass2(ref, mins, cand)
torch.cuda.synchronize()
ctx.close()
