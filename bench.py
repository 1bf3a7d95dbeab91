#!/usr/bin/env python3
"""bench.py -- quartets counted per second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N = 1): BASELINE.json configs[1] = 128 taxa, 1000 random evaluation trees, uint32
C(128,4)x3 count table (~1.07e7 quartets), seeded synthetic trees (quartetscores_amd/synth.py).
A step = one pass of the hot path over the batch of 1000 trees that is already resident in
HBM: clear the table, build the pair-depth panel, run the count kernel; for N > 1 every rank
counts its own 1000 trees (weak scaling: trees shard across GPUs) and the step ends with the
RCCL all-reduce of the count table (BASELINE.json north_star). Exactly K steps are timed
between barrier + torch.cuda.synchronize() on both sides; value = quartets counted by all
ranks / max-over-ranks time.

roofline: the dominant kernel is the count kernel. achieved = algorithmic bytes per launch
(8 B per (tree, quartet) = one read + one write of a u32 counter) / its
average launch duration measured with HIP events on the launch stream (qs_last_count_ms).
cpu_baseline: the oracle (CPU restatement of the reference, kind "port") timed on this host
on a bounded prefix of the same trees; reported, never the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)   # 0.2 ms each; short runs measure a cold, down-clocked GPU (20 steps: 0.233 ms, 1000: 0.192)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="untimed pre-conditioning before the warm-up steps (GPU clock ramp); 0 = off")
    ap.add_argument("--taxa", type=int, default=128)
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--algo", choices=["gather", "scatter"], default="gather")
    ap.add_argument("--count-bits", type=int, default=32)
    ap.add_argument("--cpu-baseline-trees", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-score", action="store_true")
    ap.add_argument("--no-impl-check", action="store_true", help="skip the on-device comparison with the byte-SWAR implementation (slow at >= 1024 taxa; parameter sweeps)")
    ap.add_argument("--distinct-trees", type=int, default=0,
                    help="generate only this many distinct trees and tile them to --trees (large configs; same GPU work)")
    ap.add_argument("--nni", action="store_true", help="evaluation trees = reference tree + Poisson(n/8) random NNIs (concentrated counts)")
    ap.add_argument("--collapse", type=float, default=0.0, help="collapse each internal edge with this probability (multifurcating trees)")
    ap.add_argument("--dropout", type=float, default=0.0, help="drop each taxon from a tree with this probability (partial trees)")
    ap.add_argument("--reduce", choices=["scatter", "all"], default="scatter",
                    help="N>1: reduce-scatter (each rank keeps and scores a shard of the reduced table) or all-reduce")
    ap.add_argument("--wire", choices=["auto", "u16x2", "u16", "u32"], default="auto",
                    help="N>1: format of the table on the wire (auto: while world x trees < 65536, u16x2 for binary full trees, else u16)")
    ap.add_argument("--table-shards", type=int, default=1,
                    help="table-sharded mode (configs[4]): split the table by the largest taxon id into this many shards")
    ap.add_argument("--shard-index", type=int, default=0, help="which shard this single-GPU run owns")
    return ap.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in quartetscores_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # QS_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all-reduce) even with one rank
    use_dist = world > 1 or os.environ.get("QS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from quartetscores_amd import engine, flatten, ranks, synth

    n, m = args.taxa, args.trees
    nq = ranks.n_quartets(n)
    # seeded inputs: seed = 1000 * config + tree set id; rank r counts tree set r
    ref_nw = synth.reference_tree(n, 2000)
    distinct = min(args.distinct_trees or m, m)
    if args.nni:   # SURVEY 8(d), second distribution: the reference tree + Poisson(n/8) random NNIs -> concentrated counts
        assert not (args.collapse or args.dropout), "--nni trees are binary and full"
        trees = synth.nni_tree_set(ref_nw, distinct, 2001 + rank)
    else:
        trees = synth.tree_set(n, distinct, 2001 + rank, collapse=args.collapse, dropout=args.dropout)
    ref = flatten.flatten_reference(ref_nw)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    if distinct < m:  # tile the flattened trees (needs every tree to hold all n taxa -> fixed stride)
        assert args.dropout == 0.0, "--distinct-trees needs full trees"
        reps = -(-m // distinct)
        ids = np.tile(batch.leaf_ids, reps)[: m * n]
        dep = np.tile(batch.adj_depth, reps)[: m * n]
        batch = flatten.TreeBatch(m, np.arange(m + 1, dtype=np.uint32) * n, ids, dep, np.zeros(m + 1, dtype=np.uint32),
                                  np.zeros(1, dtype=np.uint32), np.zeros(0, dtype=np.uint16))

    stream = torch.cuda.current_stream(dev)
    d_lo, d_hi = 0, n
    if args.table_shards > 1:
        from quartetscores_amd import distributed
        d_lo, d_hi = distributed.shard_of_largest_id(n, args.table_shards, args.shard_index)
        nq = ranks.n_quartets(d_hi) - ranks.n_quartets(d_lo)  # quartets this GPU owns
    ctx = engine.Context(n, args.count_bits, device=local_rank, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)
    n_words = (ctx.table_bytes + 3) // 4
    table = torch.zeros(n_words, dtype=torch.int32, device=dev)  # u16 tables all-reduce as packed words
    ctx.table_attach(table)
    # N > 1: the per-rank tables are combined with ONE collective per step, asynchronous on RCCL's stream, so that
    # it overlaps the counting of the next step (two buffers in flight).
    #   --reduce scatter (default): reduce-scatter; rank r ends with tuples [r*T, (r+1)*T) of the reduced table and
    #       scores that shard (distributed.score_sharded): half the bytes per link of an all-reduce.
    #   --reduce all: all-reduce, every rank ends with the full table (the wording of BASELINE.json north_star).
    # Wire format: while the summed counts stay below 2^16 (world x m trees; the reference's own CINT rule,
    # QuartetScores.cpp:115-147) the u32 table is packed to u16 cells first (qs_table_pack16): half the bytes again.
    from quartetscores_amd import distributed
    tables = [table]
    binary_full_trees = not (args.collapse or args.dropout)
    wire_fmt = None                       # None: the table's own cells travel; "u16" / "u16x2": packed first
    if use_dist and args.count_bits == 32 and args.algo == "gather":
        small = world * m < 65536
        if args.wire == "u16x2" or (args.wire == "auto" and small and binary_full_trees):
            wire_fmt = "u16x2"            # one word n0 | n1 << 16 per tuple (n2 = world * m - n0 - n1)
        elif args.wire == "u16" or (args.wire == "auto" and small):
            wire_fmt = "u16"
    wire16 = wire_fmt is not None
    if wire16:
        assert world * m < 65536, "--wire u16 / u16x2 need world x trees < 65536"
    assert wire_fmt != "u16x2" or binary_full_trees, "--wire u16x2 needs binary trees that hold all taxa"
    reduce_mode = args.reduce if use_dist else None
    bits_wire = 16 if wire16 else args.count_bits
    layout_wire = wire_fmt or args.count_bits
    chunk_words = 0
    if reduce_mode == "scatter":
        _, chunk_words = distributed.scatter_layout(ctx.table_tuples, world, layout_wire)
        send_words = world * chunk_words
        recv = [torch.zeros(chunk_words, dtype=torch.int32, device=dev) for _ in range(2)]
    else:
        send_words = ctx.table_tuples if wire_fmt == "u16x2" else distributed.table_words(ctx.table_tuples, bits_wire)
    if wire16:
        wire = [torch.zeros(send_words, dtype=torch.int32, device=dev) for _ in range(2)]
    elif use_dist:
        table = torch.zeros(max(n_words, send_words), dtype=torch.int32, device=dev)  # padded to world chunks
        ctx.table_attach(table)
        tables = [table]
        if 2 * table.numel() * 4 < 64 * (1 << 30):
            tables.append(torch.zeros_like(table))
    pending = [None] * 2
    step_no = [0]
    last_buf = [0]
    hb = ctx.batch_upload(batch, with_nodes=(args.algo == "scatter"))  # inputs resident in HBM before the timed region
    algo = engine.QS_ALGO_GATHER if args.algo == "gather" else engine.QS_ALGO_SCATTER

    # gather: QS_COUNT_OVERWRITE = "clear + count" in one pass (the kernel stores instead of accumulating)
    step_algo = algo | engine.QS_COUNT_OVERWRITE if args.algo == "gather" else algo

    def step(timed=False):
        i = step_no[0] % (2 if wire16 else len(tables))
        step_no[0] += 1
        if pending[i] is not None:       # the all-reduce that last used this buffer must be done
            pending[i].wait()
            pending[i] = None
        if len(tables) > 1:
            ctx.table_attach(tables[i])
        if args.algo != "gather":
            ctx.table_clear()
        if use_dist and wire_fmt == "u16x2" and not timed:
            # counted straight into the wire words: no table write, no pack pass (QS_COUNT_WIRE16X2)
            ctx.wire_attach(wire[i])
            ctx.count_batch(hb, step_algo | engine.QS_COUNT_WIRE16X2)
        else:
            ctx.count_batch(hb, step_algo | (engine.QS_COUNT_TIMED if timed else 0))
        if use_dist:
            if wire_fmt == "u16x2" and not timed:
                src = wire[i]
            elif wire16:
                (ctx.table_pack16x2 if wire_fmt == "u16x2" else ctx.table_pack16)(wire[i])
                src = wire[i]
            else:
                src = tables[i]
            if reduce_mode == "scatter":
                pending[i] = dist.reduce_scatter_tensor(recv[i], src[:send_words], op=dist.ReduceOp.SUM, async_op=True)
            else:
                pending[i] = dist.all_reduce(src, op=dist.ReduceOp.SUM, async_op=True)
        last_buf[0] = i

    def drain():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Untimed pre-conditioning, then the W warm-up steps the contract asks for: a 0.2 ms step is far shorter than the
    # GPU's clock ramp, so a run of 20 steps from idle measures 0.233 ms per step where 1000 steps measure 0.192 ms.
    if args.prewarm_ms > 0:
        step()                     # first step: lazy initialisation (code objects, RCCL communicator), not representative
        drain()
        torch.cuda.synchronize(dev)
        t_pre = time.perf_counter()
        step()
        drain()
        torch.cuda.synchronize(dev)
        one_ms = max((time.perf_counter() - t_pre) * 1e3, 1e-3)
        n_pre = int(min(1024, max(0, args.prewarm_ms / one_ms - 1)))
        if use_dist:      # the same number of steps (= collectives) on every rank
            npt = torch.tensor([n_pre], device=dev)
            dist.all_reduce(npt, op=dist.ReduceOp.MAX)
            n_pre = int(npt.item())
        for _ in range(n_pre):
            step()
        drain()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    fence()
    kern_ms = []
    # HIP events on the launch stream: two bracket the whole timed region (GPU time per step); the kernels of
    # the LAST timed step are bracketed individually inside qs_count_batch (QS_COUNT_TIMED; bracketing every
    # launch costs ~5 % of a 0.25 ms step, and reading events inside the loop would synchronise).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k_ in range(args.steps):
        step(timed=(k_ == args.steps - 1))
    ev1.record(stream)
    fence()
    t1 = time.perf_counter()
    ctx.sync()
    elapsed = t1 - t0
    region_gpu_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)
    last_step_ms = ctx.last_count_ms() if args.steps > 0 else None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    drain()
    # gate on the REDUCED table of the last timed step: every tuple sums to world x m (binary, full trees)
    reduced_ok = None
    shard16 = None
    if use_dist and args.steps > 0:
        if reduce_mode == "scatter":
            own_lo, own_n = distributed.scatter_owned(ctx.table_tuples, world, rank, layout_wire)
            red, n_red = recv[last_buf[0]], own_n
        else:
            red, n_red = (wire if wire16 else tables)[last_buf[0]], nq
        if wire_fmt == "u16x2":        # restore the third cell of the reduced tuples (a u16 table again)
            shard16 = torch.zeros(distributed.table_words(max(n_red, 1), 16), dtype=torch.int32, device=dev)
            ctx.unpack16x2(red, n_red, world * m, shard16)
            ctx.sync()                 # raises if a reduced tuple exceeds world x m
            w_ = red[:n_red]
            ok_local = bool((((w_ & 0xFFFF) + ((w_ >> 16) & 0xFFFF)) <= world * m).all().item())
        elif binary_full_trees:
            cells = red[: n_red * 3] if bits_wire == 32 else (red.view(torch.int16)[: n_red * 3].to(torch.int32) & 0xFFFF)
            ok_local = bool((cells.view(n_red, 3).sum(dim=1) == world * m).all().item())
            del cells
        else:
            ok_local = None
        if ok_local is not None:
            ok = torch.tensor([int(ok_local)], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank's shard must pass
            reduced_ok = bool(ok.item())
    if len(tables) > 1:                  # measurements below run on one table without collectives
        ctx.table_attach(table)
        tables[:] = [table]
        pending[:] = [None]
    use_dist_saved, use_dist = use_dist, False
    # per-kernel durations: an extra, untimed pass that reads the HIP events after every launch
    for _ in range(max(3, min(args.steps, 10))):
        step(timed=True)
        kern_ms.append(ctx.last_count_ms())
    panel_ms = float(np.mean([k[0] for k in kern_ms]))
    count_ms = float(np.mean([k[1] for k in kern_ms]))
    variant = ctx.last_count_variant()

    # parity gate run with every measurement: table of this rank's trees, checked on rank 0
    step()
    ctx.sync()
    if args.collapse or args.dropout:
        parity = None                  # tuples sum to m only when every tree resolves every quartet
    elif args.count_bits == 32:        # checked on the device
        parity = bool((table[: nq * 3].view(nq, 3).sum(dim=1) == m).all().item())
    else:
        parity = bool((ctx.table_download().sum(axis=1, dtype=np.uint64) == m).all())
    # stronger gate (the tuple sums are trivially m in the binary_full variant): the table must equal the one
    # the independent byte-SWAR implementation of the same count produces, bit for bit, on the device
    impl_match = None
    if args.algo == "gather" and "bitslice" in variant and not args.no_impl_check:
        mine = table.clone()
        os.environ["QS_GATHER_IMPL"] = "swar"
        step()
        ctx.sync()
        del os.environ["QS_GATHER_IMPL"]
        impl_match = bool(torch.equal(mine, table))
        swar_variant = ctx.last_count_variant()
        step()  # restore the default implementation's table (and variant string)
        ctx.sync()
        assert "swar" not in ctx.last_count_variant() and "depth_u" in swar_variant
        del mine

    score_ms = None
    if not args.no_score and args.count_bits == 32 and args.table_shards == 1:
        torch.cuda.synchronize(dev)
        s0 = time.perf_counter()
        if reduce_mode == "scatter" and args.steps > 0:
            # every rank scores the shard it received (view), accumulators combined with small collectives
            own_lo, own_n = distributed.scatter_owned(ctx.table_tuples, world, rank, layout_wire)
            ctx.score_set_view(shard16 if wire_fmt == "u16x2" else recv[last_buf[0]], bits_wire, own_lo, own_n)
            distributed.score_sharded(ctx, ref)
            ctx.score_set_view(None, 0, 0, 0)
        else:
            ctx.score(ref)
        score_ms = (time.perf_counter() - s0) * 1e3

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    units_per_step = m * nq * world
    value = units_per_step * args.steps / elapsed
    bytes_per_unit = 2 * (args.count_bits // 8)
    achieved = (m * nq * bytes_per_unit) / (count_ms * 1e-3) / 1e9
    out = {
        "metric": "quartets counted/sec",
        "value": value,
        "unit": "quartets/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32" if args.count_bits == 32 else "u16",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: {n} taxa, {m} random eval trees per GPU, uint{args.count_bits} C(n,4)x3 table "
                        f"({nq} quartets), seeds 2000/2001+rank" + (", NNI-perturbed copies of the reference tree" if args.nni else ""),
            "distinct_trees": distinct,
            "table_shard": [d_lo, d_hi] if args.table_shards > 1 else None,
            "algo": variant,
            "step": ("pair-depth panel build + count kernel (overwrite mode: no separate table clear)" if args.algo == "gather" else "table clear + count kernel") + ((({"u16": " + pack to u16 cells", "u16x2": ", counted straight into one word n0|n1<<16 per tuple (binary full trees: n2 = trees - n0 - n1; no table write, no pack pass)", None: ""}[wire_fmt]) + (" + RCCL reduce-scatter of the table (rank r keeps and scores tuples [r*T,(r+1)*T))" if reduce_mode == "scatter" else " + RCCL all-reduce of the table") + ", asynchronous, overlapped with the next step (two buffers in flight)") if use_dist_saved else ""),
            "collective": reduce_mode,
            "collective_input_bytes_per_rank": send_words * 4 if use_dist_saved else None,
            "parity_reduced_tuple_sums_ok": reduced_ok,
            "parity_tuple_sums_ok": parity,
            "parity_bitslice_equals_swar_impl": impl_match,
            "panel_kernel_ms": panel_ms,
            "count_kernel_ms": count_ms,
            "prewarm_ms": args.prewarm_ms,
            "count_kernel_ms_last_timed_step": last_step_ms[1] if last_step_ms else None,
            "gpu_ms_per_step_events_over_timed_region": region_gpu_ms,
            "score_phase_ms": score_ms,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": ("count_bitslice3_kernel" if "bitslice" in variant else "count_gather_kernel") if args.algo == "gather" else "count_scatter_kernel",
            "algorithmic_bytes_per_launch": m * nq * bytes_per_unit,
            "avg_launch_ms": count_ms,
            "note": "achieved = algorithmic RMW bytes of the reference formulation (8 B per tree x quartet) / kernel time; "
                    "the gather kernel keeps counters in registers and writes each cell once, so real HBM traffic is far "
                    "below this and frac can exceed 1 (its own limit is VALU issue, see DESIGN.md)",
        },
    }

    # HBM traffic of the dominant kernel: bench.py cannot collect PMC counters itself; when the committed
    # rocprofv3 summary of THIS kernel variant on THIS default workload exists, report it (per launch;
    # FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is, KB -> bytes)
    try:
        if (n, m, args.count_bits, args.table_shards) == (128, 1000, 32, 1) and ("binary_full" in variant or args.algo != "gather"):
            with open(os.path.join(ROOT, "profiles", "r01_final_pmc_summary.json")) as f:
                pmc = json.load(f)["counters"]
            kname = out["roofline"]["kernel"]
            fk = [v for k_, v in pmc["FETCH_SIZE"].items() if kname in k_]
            wk = [v for k_, v in pmc["WRITE_SIZE"].items() if kname in k_]
            vi = [v for k_, v in pmc.get("SQ_INSTS_VALU", {}).items() if kname in k_]
            if vi and "bitslice" in variant:
                # the kernel's real bound: VALU issue. Time the launch would take if the 1024 SIMDs issued its VALU
                # instructions at the rate tools/valu_rates.hip measures for this mix at 4 waves/SIMD
                # (85 % v_bitop3-class at 3.13 cycles, 15 % v_bcnt at 5.6 cycles, of a 2.4 GHz clock)
                ns_per_inst = (0.85 * 3.13 + 0.15 * 5.6) / 2.4
                bound_ms = vi[0]["avg_per_dispatch"] / 1024.0 * ns_per_inst * 1e-6
                out["roofline"]["valu_issue"] = {"wave_insts_per_launch": vi[0]["avg_per_dispatch"], "bound_ms": bound_ms,
                                                 "frac": bound_ms / count_ms,
                                                 "source": "profiles/r01_final_pmc_summary.json (SQ_INSTS_VALU) x profiles/r01_valu_rates_pass2.txt"}
            if fk and wk:
                out["roofline"]["traffic"] = (2 * fk[0]["avg_per_dispatch"] + wk[0]["avg_per_dispatch"]) * 1024
                out["roofline"]["traffic_source"] = ("profiles/r01_final_pmc_summary.json (tools/pmc_collect.sh: rocprofv3 --pmc, "
                                                     "separate FETCH_SIZE / WRITE_SIZE passes; L2 miss traffic, most of it served by the Infinity Cache)")
    except Exception:
        pass

    if not args.no_cpu_baseline and world == 1:      # reported at N=1 only (rank 0 would stall the other ranks' exit)
        try:
            from oracle_api import Oracle
            ncpu = os.cpu_count() or 1
            o = Oracle(ref_nw)
            # The reference's OpenMP loop runs INSIDE one tree over very unequal items
            # (QuartetCounterLookup.hpp:223-228) and stops scaling early; calibrate the thread
            # count on a few trees, then time a bounded sample (~10-20 s) at the best setting.
            calib = max(2, min(8, m))
            best_t, best_rate = 1, 0.0
            for th in sorted({1, min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
                tcal = o.count("\n".join(trees[:calib]), savemem=False, cint_bits=16, nthreads=th)
                if calib * nq / tcal > best_rate:
                    best_t, best_rate = th, calib * nq / tcal
            cores = best_t
            mp = args.cpu_baseline_trees or max(8, int(best_rate * 12.0 / nq))
            mp = min(mp, distinct)
            tc = o.count("\n".join(trees[:mp]), savemem=False, nthreads=cores)
            cpu_val = mp * nq / tc
            # the same prefix counted on the GPU must give the same table (bit-exact gate)
            ctx.table_clear()
            ctx.count_trees(batch.slice(0, mp), algo)
            same = bool((ctx.table_download().astype(np.uint64) == o.counts()).all())
            out["cpu_baseline"] = {
                "value": cpu_val, "unit": "quartets/s", "cores": cores, "kind": "port",
                "sample": f"first {mp} of the {m} trees, n={n}, reference fast (n^4) table, OpenMP -t {cores}, {tc:.2f} s",
                "gpu_table_bit_exact_on_sample": same,
            }
            o.close()
        except Exception as e:  # the baseline is reported, never required for the metric
            out["cpu_baseline"] = {"value": None, "unit": "quartets/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    if use_dist_saved:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line must be the LAST line on stdout: RCCL prints a version banner through C stdio, which is
    # block-buffered on a pipe and would otherwise be flushed after Python's line at exit
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
