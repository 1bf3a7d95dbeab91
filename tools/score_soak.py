"""Randomised soak of the score kernels: bundle kernel vs scan kernel (sums bit for bit, final scores identical) on random
tables, taxon counts, cell widths, reference shapes and ragged views; single-read scoring (random pre-pass samples, tie
filter on / off, log sizes that overflow or not, automatic mode) against two plain passes -- whole calls and the
pass 1 / pass 2 steps with minima LOWERED between them (what a MIN over several shards does): same candidate sets.
    python tools/score_soak.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quartetscores_amd import _lib, engine, flatten, ranks, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.choice([5, 6, 7, 11, 16, 17, 33, 47, 64, 65, 97, 129, 131, 150, 190]))
    bits = int(rng.choice([16, 32]))
    kind = rng.choice(["binary", "multif", "rooted"])
    seed = int(rng.integers(1, 1 << 30))
    if kind == "multif":
        ref_nw = synth.tree_set(n, 1, seed, collapse=0.3)[0]
    elif kind == "rooted":
        ref_nw = synth.random_tree(n, np.random.default_rng(seed), rooted=True)
    else:
        ref_nw = synth.reference_tree(n, seed)
    ref = flatten.flatten_reference(ref_nw)
    nq = ranks.n_quartets(n)
    m = int(rng.choice([7, 300, 5000, 60000])) if bits == 16 else int(rng.choice([7, 5000, 200000, 3000000]))
    p = rng.dirichlet([1.0, 1.0, 1.0])
    T = rng.multinomial(m, p, size=nq).astype(np.uint32)
    T[rng.random(nq) < rng.choice([0.0, 0.05, 0.5])] = 0
    if rng.random() < 0.4:
        tie = rng.random(nq) < 0.6
        T[tie] = np.array([m, 0, 0], dtype=np.uint32)[rng.permuted(np.tile(np.arange(3), (int(tie.sum()), 1)), axis=1)]
    dt = np.uint16 if bits == 16 else np.uint32
    ctx = engine.Context(n, bits)
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
        return sh, ctx.score_finish(ref, sh, ch[None, :], extra=extra)

    def same(x, y):
        return (x[0] == y[0]).all() and all(np.array_equal(u, v, equal_nan=True) for u, v in zip(x[1][:3], y[1][:3]))

    ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, 1)
    ok = same(steps(0), steps(1))
    # ---- single-read modes against two passes (bundle kernel, whole table)
    ctx.set_tuning(_lib.QS_TUNE_SCORE_KERNEL, 0)
    base = ctx.score(ref)
    why = ""
    for _ in range(4):
        passes = int(rng.choice([0, 2, 2, 2]))
        sample = int(rng.choice([0, 2, 4, 16, 4 | 65536, 64 | 65536, 2 | 65536]))
        dedupe = int(rng.integers(0, 2))
        cap = int(rng.choice([0, 64, 4096, 1 << 16]))
        ctx.set_tuning(_lib.QS_TUNE_SCORE_PASSES, passes)
        ctx.set_tuning(_lib.QS_TUNE_SCORE_SAMPLE, sample)
        ctx.set_tuning(_lib.QS_TUNE_SCORE_DEDUPE, dedupe)
        ctx.set_tuning(_lib.QS_TUNE_SCORE_LOG_CAP, cap)
        got = ctx.score(ref)
        if not all(np.array_equal(u, v, equal_nan=True) for u, v in zip(base[:3], got[:3])):
            ok = False
            why += f" [score passes={passes} sample={sample} dedupe={dedupe} cap={cap} log={ctx.last_score_log()}]"
        # the steps, with the minima lowered for a random subset of the node pairs between pass 1 and pass 2
        sums = torch.empty(3 * P, dtype=torch.int64, device="cuda"); mins = torch.empty(P, dtype=torch.int64, device="cuda")
        c_log = torch.empty(8 * P, dtype=torch.int64, device="cuda"); c_plain = torch.empty_like(c_log)
        ctx.score_pass1(ref, sums, mins)
        lower = torch.from_numpy(rng.random(P) < 0.3).cuda()
        big = mins < (1 << 62)
        mins2 = torch.where(lower & big, mins - int(rng.choice([1, 1 << 20, 1 << 40])), mins)
        ctx.score_pass2(ref, mins2, c_log)
        logged = ctx.last_score_log()
        ctx.score_pass2(ref, mins2, c_plain)          # (the log is spent: this one reads the table)
        a_, b_ = np.sort(c_log.cpu().numpy().reshape(-1, 8), axis=1), np.sort(c_plain.cpu().numpy().reshape(-1, 8), axis=1)
        # a node pair whose 8 slots did not suffice carries the overflow marker (-2) in both; WHICH of its triples made it into
        # the slots depends on the order of arrival (qs_score_overflow lists them all): those rows are compared by the marker only
        ov_a, ov_b = (a_ == -2).any(axis=1), (b_ == -2).any(axis=1)
        diff = (ov_a != ov_b) | (~ov_a & (a_ != b_).any(axis=1))
        if ctx.last_score_log() != 0 or diff.any():
            ok = False
            why += f" [steps passes={passes} sample={sample} dedupe={dedupe} cap={cap} log={logged}: candidate sets differ in {int(diff.sum())} pairs]"
    for key, val in ((_lib.QS_TUNE_SCORE_PASSES, 1), (_lib.QS_TUNE_SCORE_SAMPLE, 64 | 65536), (_lib.QS_TUNE_SCORE_DEDUPE, 1), (_lib.QS_TUNE_SCORE_LOG_CAP, 0)):
        ctx.set_tuning(key, val)
    full = torch.from_numpy(np.ascontiguousarray(T.astype(dt)).reshape(-1).view(np.uint8)).cuda()
    item = 3 * (bits // 8)
    for _ in range(4):
        r_lo = int(rng.integers(0, nq))
        cnt = int(rng.integers(1, nq - r_lo + 1))
        if (r_lo * item) % 4:
            r_lo -= 1 if r_lo else -1
            cnt = min(cnt, nq - r_lo)
        if (r_lo * item) % 4 or cnt < 1:
            continue
        pad = (-cnt * item) % 4
        shard = torch.zeros(cnt * item + pad, dtype=torch.uint8, device="cuda")
        shard[: cnt * item] = full[r_lo * item:(r_lo + cnt) * item]
        ctx.score_set_view(shard.view(torch.int32), bits, r_lo, cnt)
        ok = ok and same(steps(0), steps(1))
    ctx.score_set_view(None, 0, 0, 0)
    ctx.close()
    bad += not ok
    print(f"case {case}: n={n} u{bits} {kind} m={m} seed={seed}: {'ok' if ok else 'MISMATCH' + why}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
