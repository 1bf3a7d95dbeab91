"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Two ways to combine the per-rank tables of the tree-sharded mode: all_reduce_table (every rank ends with the
full table; BASELINE.json north_star's wording) and reduce_scatter_table (every rank ends with 1/world of it and
scores that shard: half the link traffic, same scores).

Tree-sharded mode (BASELINE.json north_star): evaluation trees are independent
and counts add, so rank r counts trees [lo_r, hi_r) into a private full table and ONE in-place
all-reduce (sum) of the table follows. The table lives in a torch tensor that the C-ABI context
writes through qs_table_attach, so RCCL reduces it in place, no copy. u16 tables are reduced as
packed int32 words: every cell total stays < 2^16 (enforced: m < 65536), so no carry crosses a
half-word.

The reference has no counterpart (single process, OpenMP only).
"""
from __future__ import annotations

from typing import Callable, Tuple

from . import collectives as coll


def shard_range(m: int, world: int, rank: int) -> Tuple[int, int]:
    """Trees [lo, hi) of rank `rank`: contiguous, sizes differ by at most one, covers [0, m)."""
    base, extra = divmod(m, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def table_words(n_tuples: int, count_bits: int) -> int:
    """int32 words needed to hold the [rank][3] table (u16 tables are padded to a whole word)."""
    return (n_tuples * 3 * (count_bits // 8) + 3) // 4


def _wire_words_per_2_tuples(wire) -> int:
    """int32 words that two tuples occupy on the wire: u32 cells 6, u16 cells 3, two-cell formats 2 (u16) / 4 (u32)."""
    return {32: 6, "u32": 6, 16: 3, "u16": 3, "u16x2": 2, "u32x2": 4}[wire]


def scatter_layout(n_tuples: int, world: int, wire) -> Tuple[int, int]:
    """Reduce-scatter layout of the [rank][3] table: rank r ends with tuples [r * T, min((r + 1) * T, n_tuples)).
    `wire` = 32 | 16 (cell width of the three-cell formats) or "u16x2" (one word n0 | n1 << 16 per tuple).
    Returns (T, words): T tuples per rank (even, so every chunk is a whole number of 32-bit words) and the int32
    words per chunk; the send buffer is world * words long (zero padded behind the table)."""
    t = -(-n_tuples // world)
    t += t & 1
    return t, t // 2 * _wire_words_per_2_tuples(wire)


def scatter_owned(n_tuples: int, world: int, rank: int, wire) -> Tuple[int, int]:
    """(first tuple rank, number of tuples) that `rank` owns after reduce_scatter_table."""
    t, _ = scatter_layout(n_tuples, world, wire)
    lo = min(rank * t, n_tuples)
    return lo, min(lo + t, n_tuples) - lo


def reduce_scatter_table(send, recv, group=None, async_op=False):
    """Sum over the ranks of `send` (world * words int32 words, scatter_layout); this rank's chunk lands in `recv`
    (words int32 words). Half the bytes per rank of an all-reduce -- xGMI links are the bottleneck of the
    tree-sharded mode -- and the LQ/QP/EQP reduction works on shards anyway (score_sharded with a view)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return coll.reduce_scatter_tensor(recv, send, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    recv.copy_(send[: recv.numel()])
    return None


def shard_bounds(n: int, shards: int, by: str = "c4"):
    """bounds[0..shards] of the largest taxon id: qs_shard_bounds of the C-ABI (host-only arithmetic, no GPU needed).
    by="c4": balanced by the tuples held; by="cost": balanced by the count kernel's work (tools/scaling_model.py)."""
    import ctypes as C
    from . import _lib
    out = (C.c_uint32 * (shards + 1))()
    rc = _lib.load().qs_shard_bounds(n, shards, _lib.QS_SHARDS_BY_COST if by == "cost" else _lib.QS_SHARDS_BY_TUPLES, out)
    if rc != 0:
        raise ValueError(f"qs_shard_bounds({n}, {shards}, {by}) failed: {rc}")
    return [int(x) for x in out]


def shard_of_largest_id(n: int, world: int, rank: int, by: str = "c4") -> Tuple[int, int]:
    """Table-sharded mode: [d_lo, d_hi) of the largest taxon id, balanced by C(d,4) (ranks are
    contiguous in d because rank's leading term is C(s3,4), quartet_lookup_table.hpp:161-165).
    by="cost": balanced by what the count kernel spends on a shard instead of the tuples it holds (shard_bounds)."""
    if by == "cost":
        b = shard_bounds(n, world, "cost")
        return b[rank], b[rank + 1]

    def c4(x):
        return x * (x - 1) * (x - 2) * (x - 3) // 24
    total = c4(n)
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        d = bounds[-1]
        while d < n and c4(d) < target:
            d += 1
        bounds.append(d)
    bounds.append(n)
    return bounds[rank], bounds[rank + 1]


XGMI_LINK_GBS = 153.0            # MI355X_MICROARCH.md: per xGMI link and direction, 7 links per GPU


def auto_mode(n, m_total, world, binary_full=True, rccl_init_ms=0.0):
    """tree- or table-sharded for N > 1, from what one GPU could settle (tools/scaling_model.py, profiles/r06_scaling_model.json):
    the count work per rank is the same either way (trees x owned tuples); the tree-sharded mode adds ONE collective on the table,
    the table-sharded mode adds the replicated panel build, the shards' imbalance and the tails of its smaller launches. Returns
    (mode, the estimate). Constants: 1.6e-9 ms per (tree, taxon pair) of panel build (measured 1.3e-9 at 512 taxa, 1.8e-9 at
    256), 9.4e13 quartets/s per GPU, ~8 % imbalance + tails, 1 ms per launch (one per slice of >= 350 MB of panel); the
    collective is priced at the ring bound: reduce-scatter bytes per rank / 153 GB/s (one xGMI link at a time)."""
    nq = n * (n - 1) * (n - 2) * (n - 3) // 24
    npairs = n * (n - 1) // 2
    bpt = (4 if binary_full else 6) if m_total < 65536 else (8 if binary_full else 12)
    coll_ms = nq * bpt * (world - 1) / world / (XGMI_LINK_GBS * 1e9) * 1e3 + rccl_init_ms   # (+ the communicators of a one-shot run)
    count_ms = m_total * nq / 9.4e13 * 1e3 / world
    panel_ms = 1.6e-9 * m_total * npairs
    groups = -(-m_total // 32)
    slice_groups = max(256, int(350e6 // (npairs * 16)))
    table_extra_ms = panel_ms * (world - 1) / world + 0.08 * count_ms + 1.0 * -(-groups // slice_groups)
    mode = "table" if table_extra_ms < coll_ms else "tree"
    return mode, {"tree_collective_ms_ring_bound": round(coll_ms, 3), "table_extra_ms": round(table_extra_ms, 3), "count_ms_per_rank": round(count_ms, 3),
                  "rule": "table if replicated panel + imbalance/tails < table collective at one xGMI link", "source": "profiles/r06_scaling_model.json"}


def all_reduce_table(table, group=None):
    """In-place sum of the count table over all ranks (RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        coll.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return table


def count_tree_sharded(n_trees: int, count_local: Callable[[int, int], None], table, group=None, pre_reduce=None):
    """Run `count_local(lo, hi)` on this rank's tree range, then all-reduce `table`.

    count_local adds the quartet counts of trees [lo, hi) into `table` (on the GPU this is
    Context.count_trees on a sliced batch with `table` attached). pre_reduce, if given, runs between the two
    (the u16 wire format packs the counted u32 table into `table` there)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_trees, world, rank)
    if hi > lo:
        count_local(lo, hi)
    if pre_reduce is not None:
        pre_reduce()
    return all_reduce_table(table, group)


def count_trees_multi_gpu(ref, batch, count_bits: int = 32, algo: int = 0, device=None, wire: str = "auto"):
    """Tree-sharded counting on the current torch.distributed world (one rank per GPU).
    Returns (Context, table tensor); every rank ends with the full reduced table.

    wire: cell width the ranks exchange. With a u32 table and fewer than 65536 trees in total ("auto", or "u16")
    each rank packs its table to u16 cells (Context.table_pack16) and the all-reduce moves half the bytes; the
    returned Context is then a count_bits=16 one attached to the reduced packed table."""
    import torch
    import torch.distributed as dist
    from . import engine
    if count_bits == 16 and batch.n_trees >= (1 << 16):
        raise ValueError("u16 tables need fewer than 65536 trees in total")
    if wire not in ("auto", "u16", "u32"):
        raise ValueError("wire must be auto, u16 or u32")
    if wire == "u16" and batch.n_trees >= (1 << 16):
        raise ValueError("a u16 wire format needs fewer than 65536 trees in total")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream(dev)
    ctx = engine.Context(ref.n_taxa, count_bits, device=dev.index or 0, stream=stream.cuda_stream)
    table = torch.zeros(table_words(ctx.table_tuples, count_bits), dtype=torch.int32, device=dev)
    ctx.table_attach(table)

    def local(lo, hi):
        ctx.count_trees(batch.slice(lo, hi), algo)

    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if count_bits == 32 and (wire == "u16" or (wire == "auto" and multi and batch.n_trees < (1 << 16))):
        packed = torch.zeros(table_words(ctx.table_tuples, 16), dtype=torch.int32, device=dev)
        count_tree_sharded(batch.n_trees, local, packed, pre_reduce=lambda: ctx.table_pack16(packed))
        ctx16 = engine.Context(ref.n_taxa, 16, device=dev.index or 0, stream=stream.cuda_stream)
        ctx16.table_attach(packed)
        ctx.close()
        torch.cuda.synchronize(dev)
        return ctx16, packed
    count_tree_sharded(batch.n_trees, local, table)
    if dist.is_initialized():
        torch.cuda.synchronize(dev)
    return ctx, table


def count_trees_reduce_scatter(ref, batch, algo: int = 0, device=None, wire: str = "auto", group=None):
    """Tree-sharded counting that ends with the table SHARDED over the ranks (reduce-scatter instead of
    all-reduce): every rank holds the whole `batch` and counts its own slice of it.
    Returns (ctx, shard, bits, rank_lo, n_owned), see reduce_scatter_counts."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    lo, hi = shard_range(batch.n_trees, world, rank)
    return reduce_scatter_counts(ref, batch.slice(lo, hi), batch.n_trees, algo, device, wire, group)


def reduce_scatter_counts(ref, local_batch, total_trees: int, algo: int = 0, device=None, wire: str = "auto", group=None):
    """Count THIS rank's trees (`local_batch`; total_trees = trees over all ranks), reduce-scatter the table.
    Returns (ctx, shard, bits, rank_lo, n_owned): `shard` holds tuples [rank_lo, rank_lo + n_owned) of the reduced
    table with `bits`-bit cells; ctx has that range set as its scoring view, so score_sharded(ctx, ref) gives the
    scores.

    wire: "u32" | "u16" (three cells per tuple; u16 needs fewer than 65536 trees in total) | "u16x2" (one word per
    tuple; only for batches of binary trees that hold all taxa -- anything else is reported as an error by the
    library) | "auto": the narrowest of these that the trees of ALL ranks allow (qs_batch_flags, agreed with a MIN
    all-reduce)."""
    import torch
    import torch.distributed as dist
    from . import _lib, engine
    if wire not in ("auto", "u16x2", "u16", "u32", "u32x2"):
        raise ValueError("wire must be auto, u16x2, u16, u32x2 or u32")
    if wire in ("u16", "u16x2") and total_trees >= (1 << 16):
        raise ValueError("a u16 wire format needs fewer than 65536 trees in total")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    stream = torch.cuda.current_stream(dev)
    ctx = engine.Context(ref.n_taxa, 32, device=dev.index or 0, stream=stream.cuda_stream)
    hb = ctx.batch_upload(local_batch, with_nodes=(algo == engine.QS_ALGO_SCATTER)) if local_batch.n_trees else None
    try:
        if wire == "auto":   # the narrowest format the trees allow, the same on every rank
            both = _lib.QS_BATCH_ALL_TAXA | _lib.QS_BATCH_BINARY
            ok = 1 if (hb is None or (ctx.batch_flags(hb) & both) == both) and algo != engine.QS_ALGO_SCATTER else 0
            if multi and world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                coll.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if total_trees >= (1 << 16):
                wire = "u32x2" if ok else "u32"     # binary full trees: two cells (n0, n1) instead of three
            else:
                wire = "u16x2" if ok else "u16"
        bits = 32 if wire in ("u32", "u32x2") else 16
        t_chunk, words = scatter_layout(ctx.table_tuples, world, wire)
        send = torch.zeros(world * words, dtype=torch.int32, device=dev)
        if wire == "u16x2":
            # counted straight into the wire words (QS_COUNT_WIRE16X2): no table, no pack pass; the library refuses
            # batches that are not binary with all taxa
            ctx.wire_attach(send)
            if hb is not None:
                ctx.count_batch(hb, engine.QS_ALGO_GATHER | engine.QS_COUNT_WIRE16X2)
        else:
            if wire == "u32":
                ctx.table_attach(send)                 # counted in place, padded to world chunks
            else:
                table = torch.zeros(table_words(ctx.table_tuples, 32), dtype=torch.int32, device=dev)
                ctx.table_attach(table)
            if hb is not None:
                ctx.count_batch(hb, algo)
            if wire == "u16":
                ctx.table_pack16(send)
            elif wire == "u32x2":
                ctx.table_pack32x2(send)               # (n0, n1) per tuple; refused by qs_sync if a tuple does not sum to the trees
        ctx.sync()
    finally:
        if hb is not None:
            ctx.batch_free(hb)
    recv = torch.zeros(words, dtype=torch.int32, device=dev)
    reduce_scatter_table(send, recv, group)
    rank_lo, n_owned = scatter_owned(ctx.table_tuples, world, rank, wire)
    shard = recv
    if wire == "u32x2":
        # the full 12 B / tuple table is not needed any more: detach and release it before the shard is allocated
        ctx.table_attach(None)
        table = None
    # the reduced shard holds ALL ranks' trees: the scoring kernels size their k*log k table from it, not from this rank's share
    ctx.set_tuning(_lib.QS_TUNE_TABLE_TREES, int(total_trees))
    if wire == "u16x2":                            # restore the third cell: n2 = total trees - n0 - n1
        shard = torch.zeros(table_words(max(n_owned, 1), 16), dtype=torch.int32, device=dev)
        ctx.unpack16x2(recv, n_owned, total_trees, shard)
    elif wire == "u32x2":
        shard = torch.zeros(table_words(max(n_owned, 1), 32), dtype=torch.int32, device=dev)
        ctx.unpack32x2(recv, n_owned, total_trees, shard)
    ctx.sync()                                     # raises if a tuple did not fit the wire format
    ctx.score_set_view(shard, bits, rank_lo, n_owned)
    return ctx, shard, bits, rank_lo, n_owned


def score_sharded(ctx, ref, flags: int = 0, group=None, device=None):
    """LQ/QP/EQP-IC when every rank holds only a SHARD of the count table (table-sharded mode,
    SURVEY.md 8(e)): each rank reduces its own quartets to per-node-pair accumulators, which are
    combined over the ranks (SUM of the count sums, MIN of the best device QIC, all-gather of the
    near-minimal count triples) before the O(#node pairs) host finalisation. Also correct on a
    single rank. Returns (lq, qp, eqp, is_bifurcating) indexed by child node like Context.score."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _lib
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    P = ctx.score_pair_slots(ref)
    sums = torch.empty(3 * P, dtype=torch.int64, device=dev)
    mins = torch.empty(P, dtype=torch.int64, device=dev)
    cand = torch.empty(_lib.QS_SCORE_CAND_SLOTS * P, dtype=torch.int64, device=dev)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    ctx.score_pass1(ref, sums, mins)
    if multi:
        coll.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        coll.all_reduce(mins, op=dist.ReduceOp.MIN, group=group)
    ctx.score_pass2(ref, mins, cand)
    extra = ctx.score_overflow(ref, mins, cand)     # (k, 4) near-minimal quartets of overflowed node pairs; k = 0 almost always
    if multi:
        parts = [torch.empty_like(cand) for _ in range(dist.get_world_size(group))]
        coll.all_gather(parts, cand, group=group)
        cand_host = np.stack([p.cpu().numpy() for p in parts])
        lists = [None] * dist.get_world_size(group)
        dist.all_gather_object(lists, extra, group=group)
        extra = np.concatenate(lists) if any(len(x) for x in lists) else extra
    else:
        cand_host = cand.cpu().numpy()[None, :]
    return ctx.score_finish(ref, sums.cpu().numpy(), cand_host, flags, extra=extra)


def shards_of_rank(n_shards: int, world: int, rank: int):
    """Table-sharded mode with K >= world shards: shard s lives on rank s mod world."""
    return [k for k in range(n_shards) if k % world == rank]


def score_table_shards(open_shard: Callable, my_shards, ref, flags: int = 0, group=None, device=None, close=None):
    """LQ/QP/EQP-IC in the table-sharded mode (BASELINE configs[4]; blueprint: the reference's quartet-major loop,
    QuartetScoreComputer.hpp:212-371): the table is cut into K shards by the largest taxon id, this rank owns
    `my_shards` (shards_of_rank). open_shard(k) returns a context whose table holds shard k with ALL evaluation trees
    counted (engine.Context created with shard_of_largest_id(n, K, k); there is no table collective).
      round 1: score pass 1 per owned shard, SUM of the sums / MIN of the minima over the rank's shards, then over the ranks;
      round 2: score pass 2 against the global minima per owned shard (a rank with ONE shard keeps its context between the
               rounds, otherwise open_shard is called again = count again), all-gather of the candidates and overflow lists;
      qs_score_finish on every rank.
    close(ctx), if given, releases a context this function is done with. Returns (lq, qp, eqp, is_bifurcating)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _lib
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if multi else 1
    my_shards = list(my_shards)
    per_rank = len(my_shards)
    if multi:   # ranks may own different numbers of shards: pad the gathers to the maximum
        cnt = torch.tensor([per_rank], dtype=torch.int64, device=dev)
        coll.all_reduce(cnt, op=dist.ReduceOp.MAX, group=group)
        per_rank = int(cnt.item())
    P = None
    kept = None
    sums = mins = None
    finisher = None
    for k in my_shards:
        ctx = open_shard(k)
        finisher = finisher or ctx
        if P is None:
            P = ctx.score_pair_slots(ref)
            sums = torch.zeros(3 * P, dtype=torch.int64, device=dev)
            mins = torch.full((P,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
            s1 = torch.empty_like(sums)
            m1 = torch.empty_like(mins)
        ctx.score_pass1(ref, s1, m1)
        sums += s1
        mins = torch.minimum(mins, m1)
        if len(my_shards) == 1:
            kept = ctx
        elif close is not None:
            if hasattr(ctx, "sync"):
                ctx.sync()
            close(ctx)
    if P is None:            # a rank without a shard still takes part in the collectives
        probe = open_shard(None)
        finisher = probe
        P = probe.score_pair_slots(ref)
        sums = torch.zeros(3 * P, dtype=torch.int64, device=dev)
        mins = torch.full((P,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    if multi:
        coll.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        coll.all_reduce(mins, op=dist.ReduceOp.MIN, group=group)
    slots = _lib.QS_SCORE_CAND_SLOTS
    cand_all = torch.full((max(per_rank, 1), slots * P), -1, dtype=torch.int64, device=dev)   # -1 = empty slot
    extras = []
    for i, k in enumerate(my_shards):
        ctx = kept if kept is not None else open_shard(k)
        finisher = ctx
        ctx.score_pass2(ref, mins, cand_all[i])
        ex = ctx.score_overflow(ref, mins, cand_all[i])
        if len(ex):
            extras.append(ex)
        if kept is None and close is not None and i + 1 < len(my_shards):
            ctx.sync()
            close(ctx)
    extra = np.concatenate(extras) if extras else np.zeros((0, 4), dtype=np.int64)
    if multi:
        parts = [torch.empty_like(cand_all) for _ in range(world)]
        coll.all_gather(parts, cand_all, group=group)
        cand_host = np.concatenate([p_.cpu().numpy() for p_ in parts])
        lists = [None] * world
        dist.all_gather_object(lists, extra, group=group)
        extra = np.concatenate(lists) if any(len(x) for x in lists) else extra
    else:
        cand_host = cand_all.cpu().numpy()
    return finisher.score_finish(ref, sums.cpu().numpy(), cand_host, flags, extra=extra)
