"""Multi-GPU command-line driver: the reference's CLI surface (-r -e -o [-v], QuartetScores.cpp:48-79) on N GPUs of
one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m quartetscores_amd.dist_cli -r ref.nwk -e eval.nwk -o out.nwk [-v] [--exact-qp] [--mode auto|tree|table] [--wire auto|u16x2|u16|u32]

Every rank parses the evaluation file, flattens and counts its own contiguous share of the trees
(distributed.shard_range), the tables are combined with one RCCL reduce-scatter, every rank scores the shard it
received and the per-node-pair accumulators are combined with small collectives (distributed.score_sharded); rank 0
writes the annotated Newick file in the format of the single-GPU CLI (quartetscores_amd/bin/QuartetScores).
Works unchanged with one process (no launcher). There is no CPU fallback.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time


def score_flags(args):
    from . import _lib
    return ((_lib.QS_SCORE_QP_EXACT64 if args.exact_qp else _lib.QS_SCORE_QP_WRAP32) |
            (_lib.QS_SCORE_ROOT_AS_EDGE if args.root_as_edge else 0) | (_lib.QS_SCORE_SAVEMEM_LOOKUPS if args.savemem else 0))


def _annotate(ref, lq, qp, eqp, bif):
    """Newick text with "qp-ic:X;lq-ic:Y;eqp-ic:Z" comments per edge, parts omitted when +inf; the qp-ic guard tests
    the LQ value like the reference (quartet_newick_writer.hpp:164-187, quirk Q6). %f = std::to_string(double)."""
    from . import newick
    index = {id(nd): i for i, nd in enumerate(ref.nodes)}

    def comment(nd):
        v = index[id(nd)]
        if v == 0:
            return None
        parts = []
        if bif and lq[v] != math.inf:
            parts.append("qp-ic:%f" % qp[v])
        if lq[v] != math.inf:
            parts.append("lq-ic:%f" % lq[v])
        if bif and eqp[v] != math.inf:
            parts.append("eqp-ic:%f" % eqp[v])
        return ";".join(parts) if parts else None
    return newick.write(ref.nodes[0], comment)


def _table_sharded(args, ref, world, rank, dev, say, t_begin):
    """--table-shards K: every rank ingests ALL trees, counts them into the shard(s) it owns (contexts created with
    [d_lo, d_hi) of the largest taxon id), and the scores come from distributed.score_table_shards."""
    import torch
    import torch.distributed as dist
    from . import _lib, distributed, engine, flatten, native_ingest, newick, ranks
    n_shards = args.table_shards or world
    if n_shards < world:
        raise ValueError(f"--table-shards {n_shards}: fewer shards than ranks ({world})")
    if native_ingest.available():
        batch, m = native_ingest.ingest(args.ref, args.eval, 0, native_ingest.ALL, args.threads, want_ranges=False)
    else:
        trees = list(newick.parse_trees(open(args.eval).read()))
        m = len(trees)
        batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    bits = 16 if m < (1 << 16) else 32           # counter width by m (QuartetScores.cpp:115-147; u8 widened to u16)
    say(f"There are {m} evaluation trees.")
    say(f"The reference tree has {ref.n_taxa} taxa.")
    say(f"Counting in {n_shards} table shard(s) by largest taxon id on {world} GPU(s): every GPU counts all trees into its "
        f"shard(s), no table collective.")
    t0 = time.perf_counter()
    stream = torch.cuda.current_stream(dev)
    by = "cost" if n_shards == world and world > 1 else "c4"
    bounds = distributed.shard_bounds(ref.n_taxa, n_shards, by)
    # (tiny taxon counts leave some of many shards empty: a rank without a shard still takes part in the collectives of the scoring)
    mine = [k for k in distributed.shards_of_rank(n_shards, world, rank) if ranks.n_quartets(bounds[k + 1]) > ranks.n_quartets(bounds[k])]
    counted = []

    def open_shard(k):
        if k is None:      # a rank without a shard: a context only for the host-side parts
            return engine.Context(ref.n_taxa, bits, device=dev.index or 0, stream=stream.cuda_stream)
        # one resident shard per rank: balanced by the count kernel's work; more shards than ranks: by the tuples held (memory decides)
        d_lo, d_hi = bounds[k], bounds[k + 1]
        ctx = engine.Context(ref.n_taxa, bits, device=dev.index or 0, stream=stream.cuda_stream, d_lo=d_lo, d_hi=d_hi)
        ctx.table_alloc()
        ctx.count_trees(batch, engine.QS_ALGO_GATHER)
        ctx.sync()
        counted.append(k)
        if args.verbose:
            print(f"[rank {rank}] shard {k}: largest id in [{d_lo},{d_hi}), {ctx.table_bytes} bytes ({ctx.last_count_variant()})")
        return ctx
    flags = score_flags(args)
    lq, qp, eqp, bif = distributed.score_table_shards(open_shard, mine, ref, flags, device=dev, close=lambda c: c.close())
    if world > 1:
        dist.barrier()
    say("Finished counting quartets and computing scores.")
    say(f"It took: {int((time.perf_counter() - t0) * 1e6)} microseconds.")
    say("The reference tree is bifurcating." if bif else "The reference tree is multifurcating.")
    if rank == 0:
        with open(args.output, "w") as f:
            f.write(_annotate(ref, lq, qp, eqp, bif) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    say(f"Elapsed time: {int((time.perf_counter() - t_begin) * 1e6)} microseconds.")
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="quartetscores_amd.dist_cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("-r", "--ref", required=True)
    ap.add_argument("-e", "--eval", required=True)
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("-t", "--threads", type=int, default=0, help="host threads per rank for parsing the evaluation trees (0 = all)")
    ap.add_argument("--exact-qp", action="store_true", help="64-bit QP sums instead of the reference's 32-bit wrap")
    ap.add_argument("-s", "--savemem", action="store_true",
                    help="the reference's memory-efficient table behind the lookups of a ROOTED reference tree: it throws there "
                         "(quartet_lookup_table.hpp:79-85) and so does this run (QS_SCORE_SAVEMEM_LOOKUPS)")
    ap.add_argument("--root-as-edge", action="store_true", help="a degree-2 reference root as a point on one edge (QS_SCORE_ROOT_AS_EDGE)")
    ap.add_argument("--wire", choices=["auto", "u16x2", "u16", "u32"], default="auto")
    ap.add_argument("--mode", choices=["auto", "tree", "table"], default="auto",
                    help="how N ranks share the job (DESIGN.md 5): tree = trees / N per rank + ONE reduce-scatter of the table; table = every rank "
                         "counts ALL trees into its shard of the table (by largest taxon id, cost-balanced), no table collective; auto = by the model "
                         "(distributed.auto_mode; a one-shot run also pays for RCCL's communicators on the tree route)")
    ap.add_argument("--table-shards", type=int, default=-1,
                    help="table-sharded mode (1024 taxa x u16 = 273 GB: 8 ranks, 8 shards): the count table in K shards by largest taxon "
                         "id, shard s on rank s mod N, every rank counts ALL trees into its shard(s), no table collective; 0 = one per rank")
    args = ap.parse_args(argv)
    t_begin = time.perf_counter()

    import torch
    import torch.distributed as dist
    from . import _lib, distributed, flatten, native_ingest, newick
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.path.exists(args.output):
        if rank == 0:
            print("ERROR: The specified output file already exists.")
        return 1
    if not torch.cuda.is_available():
        print("ERROR: no HIP device (quartetscores_amd has no CPU fallback)", file=sys.stderr)
        return 1
    # QS_DIST_BACKEND=gloo (rehearsal, tests): the ranks share the visible GPUs and talk over gloo, device tensors staged through the
    # host (collectives.py) -- RCCL refuses two ranks on one device
    backend = os.environ.get("QS_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    say = print if rank == 0 else (lambda *a, **k: None)
    try:
        ref = flatten.flatten_reference(open(args.ref).read())
        if args.table_shards >= 0:
            return _table_sharded(args, ref, world, rank, dev, say, t_begin)
        if world > 1 and args.mode != "tree":
            # one cheap scan for the number of trees decides (every rank arrives at the same answer)
            m_scan = native_ingest.ingest(args.ref, args.eval, 0, 0, args.threads, want_ranges=False)[1] if native_ingest.available() \
                else sum(1 for _ in newick.parse_trees(open(args.eval).read()))
            choice, est = distributed.auto_mode(ref.n_taxa, m_scan, world, rccl_init_ms=1700.0)
            if args.mode == "table" or choice == "table":
                say(f"Table-sharded counting on {world} GPUs ({'--mode table' if args.mode == 'table' else 'auto'}: table collective ~"
                    f"{est['tree_collective_ms_ring_bound']} ms against ~{est['table_extra_ms']} ms of replicated panel build and imbalance).")
                args.table_shards = world
                return _table_sharded(args, ref, world, rank, dev, say, t_begin)
        if native_ingest.available():
            # the C++ host's multi-threaded ingest: one scan for the tree spans (= m), then only this rank's share is
            # parsed and flattened (the reference parses the whole file twice on one thread, QuartetScores.cpp:23-32)
            _, m = native_ingest.ingest(args.ref, args.eval, 0, 0, args.threads, want_ranges=False)
            lo, hi = distributed.shard_range(m, world, rank)
            local, _ = native_ingest.ingest(args.ref, args.eval, lo, hi, args.threads, want_ranges=False)
        else:
            say("note: libquartetscores_host.so not built; using the (slow) Python Newick parser")
            trees = list(newick.parse_trees(open(args.eval).read()))
            m = len(trees)
            lo, hi = distributed.shard_range(m, world, rank)
            local = flatten.flatten_eval_trees(trees[lo:hi], ref.name_to_id)
        say(f"There are {m} evaluation trees.")
        say(f"The reference tree has {ref.n_taxa} taxa.")
        t0 = time.perf_counter()
        ctx, shard, bits, rank_lo, n_owned = distributed.reduce_scatter_counts(ref, local, m, device=dev, wire=args.wire)
        if args.verbose:
            print(f"[rank {rank}] trees [{lo},{hi}) counted ({ctx.last_count_variant() if hi > lo else 'none'}); owns tuples "
                  f"[{rank_lo},{rank_lo + n_owned}) as u{bits}")
        say("Finished counting quartets.")
        say(f"It took: {int((time.perf_counter() - t0) * 1e6)} microseconds.")
        t0 = time.perf_counter()
        flags = score_flags(args)
        lq, qp, eqp, bif = distributed.score_sharded(ctx, ref, flags, device=dev)
        say("The reference tree is bifurcating." if bif else "The reference tree is multifurcating.")
        say("Finished computing scores.")
        say(f"It took: {int((time.perf_counter() - t0) * 1e6)} microseconds.")
        if rank == 0:
            with open(args.output, "w") as f:
                f.write(_annotate(ref, lq, qp, eqp, bif) + "\n")
    except Exception as e:  # same exit behaviour as the single-GPU CLI: message + status 1
        print(f"ERROR: {e}", file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        return 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    say(f"Elapsed time: {int((time.perf_counter() - t_begin) * 1e6)} microseconds.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
