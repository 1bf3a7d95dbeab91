"""world_size-2 gloo tests of the N>1 path (CPU): tree sharding + table all-reduce give the
single-process table bit for bit, for u32 and for packed-u16 tables. The per-rank counter here
is the numpy emulation of the device arithmetic (tests/emulate.py) because there is no GPU in
this container; on the GPU the same host logic drives Context.count_trees."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, count_bits, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulate
    from quartetscores_amd import distributed, flatten, ranks, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 9, 21
    ref = flatten.flatten_reference(synth.reference_tree(n, 3))
    batch = flatten.flatten_eval_trees(synth.tree_set(n, m, 4, collapse=0.2, dropout=0.1), ref.name_to_id)
    nq = ranks.n_quartets(n)
    table = torch.zeros(distributed.table_words(nq, count_bits), dtype=torch.int32)
    dt = np.uint32 if count_bits == 32 else np.uint16
    view = table.numpy().view(dt)[: nq * 3].reshape(nq, 3)

    def local(lo, hi):
        view[...] += emulate.counts_from_batch(batch.slice(lo, hi), n).astype(dt)

    distributed.count_tree_sharded(m, local, table)
    np.save(os.path.join(out_dir, f"t{rank}.npy"), view.copy())
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), emulate.counts_from_batch(batch, n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("count_bits", [32, 16])
def test_tree_sharded_allreduce_equals_single_process(tmp_path, count_bits):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), count_bits, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    for r in range(world):
        assert (np.load(tmp_path / f"t{r}.npy").astype(np.uint64) == full).all()


def _rs_worker(rank, world, port, count_bits, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulate
    from quartetscores_amd import distributed, flatten, ranks, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 9, 23
    ref = flatten.flatten_reference(synth.reference_tree(n, 5))
    batch = flatten.flatten_eval_trees(synth.tree_set(n, m, 6, collapse=0.2, dropout=0.1), ref.name_to_id)
    nq = ranks.n_quartets(n)   # 126 tuples over 2 ranks: 63 -> 64 per rank (even), 2 padding tuples
    t_chunk, words = distributed.scatter_layout(nq, world, count_bits)
    send = torch.zeros(world * words, dtype=torch.int32)
    dt = np.uint32 if count_bits == 32 else np.uint16
    lo, hi = distributed.shard_range(m, world, rank)
    send.numpy().view(dt)[: nq * 3].reshape(nq, 3)[...] = emulate.counts_from_batch(batch.slice(lo, hi), n).astype(dt)
    recv = torch.zeros(words, dtype=torch.int32)
    distributed.reduce_scatter_table(send, recv)
    r_lo, n_own = distributed.scatter_owned(nq, world, rank, count_bits)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), recv.numpy().view(dt)[: n_own * 3].reshape(n_own, 3).copy())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([r_lo, n_own, t_chunk]))
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), emulate.counts_from_batch(batch, n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("count_bits", [32, 16])
def test_tree_sharded_reduce_scatter_gives_each_rank_its_shard(tmp_path, count_bits):
    world = 2
    mp.spawn(_rs_worker, args=(world, _free_port(), count_bits, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    covered = 0
    for r in range(world):
        r_lo, n_own, t_chunk = (int(v) for v in np.load(tmp_path / f"r{r}.npy"))
        assert r_lo == covered and t_chunk % 2 == 0
        assert (np.load(tmp_path / f"s{r}.npy").astype(np.uint64) == full[r_lo:r_lo + n_own]).all()
        covered += n_own
    assert covered == full.shape[0]


def test_scatter_layout_properties():
    from quartetscores_amd import distributed
    for nq in (1, 2, 70, 126, 10668000, 2862209280):
        for w in (1, 2, 3, 8):
            for bits in (16, 32, "u16x2", "u32x2"):
                t, words = distributed.scatter_layout(nq, w, bits)
                bytes_per_tuple = {"u16x2": 4, "u32x2": 8}.get(bits) or 3 * bits // 8
                assert t % 2 == 0 and t * w >= nq and words * 4 == t * bytes_per_tuple
                owned = [distributed.scatter_owned(nq, w, r, bits) for r in range(w)]
                assert owned[0][0] == 0 and sum(c for _, c in owned) == nq
                assert all(a[0] + a[1] == b[0] for a, b in zip(owned, owned[1:]))


def test_shard_ranges_cover_and_balance():
    from quartetscores_amd import distributed
    for m in (0, 1, 7, 1000, 100001):
        for w in (1, 2, 3, 8):
            spans = [distributed.shard_range(m, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    # table sharding by the largest id: contiguous, balanced by C(d,4)
    n, w = 1024, 8
    b = [distributed.shard_of_largest_id(n, w, r) for r in range(w)]
    assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    c4 = lambda x: x * (x - 1) * (x - 2) * (x - 3) // 24
    sizes = [c4(hi) - c4(lo) for lo, hi in b]
    assert max(sizes) / (c4(n) / w) < 1.02


def test_packed_u16_sum_has_no_cross_carry():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 30000, size=1000, dtype=np.uint16)
    b = rng.integers(0, 30000, size=1000, dtype=np.uint16)
    s = (a.view(np.int32) + b.view(np.int32)).view(np.uint16)
    assert (s == a + b).all()


def _score_worker(rank, world, port, bif, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulate
    from quartetscores_amd import distributed, flatten, ranks, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 11, 40
    ref_nw = synth.reference_tree(n, 31) if bif else "((t0,t1,t2),(t3,t4),(t5,(t6,t7,t8)),(t9,t10));"
    ref = flatten.flatten_reference(ref_nw)
    trees = synth.tree_set(n, m, 32, collapse=0.15)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    nq = ranks.n_quartets(n)
    full = emulate.counts_from_batch(batch, n)
    # the shard this rank would hold after the reduce-scatter of the table
    r_lo, n_own = distributed.scatter_owned(nq, world, rank, 32)
    ctx = emulate.ScoreEmu(ref, full[r_lo:r_lo + n_own], r_lo)
    lq, qp, eqp, is_bif = distributed.score_sharded(ctx, ref, device=torch.device("cpu"))
    np.save(os.path.join(out_dir, f"lq{rank}.npy"), lq)
    if is_bif:
        np.save(os.path.join(out_dir, f"qp{rank}.npy"), qp)
        np.save(os.path.join(out_dir, f"eqp{rank}.npy"), eqp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bif", [True, False])
def test_score_sharded_sum_min_allgather_plumbing(tmp_path, bif):
    """distributed.score_sharded over 2 gloo ranks, each holding half of the tuples (the numpy emulation of score pass
    1 / 2 stands in for the kernels, the finish is the library's host code): SUM of the sums, MIN of the minima,
    all-gather of the candidates give the oracle's scores exactly, on every rank (QuartetScoreComputer.hpp:379-593)."""
    from oracle_api import Oracle
    from quartetscores_amd import flatten, newick, synth
    world = 2
    mp.spawn(_score_worker, args=(world, _free_port(), bif, str(tmp_path)), nprocs=world, join=True)
    n = 11
    ref_nw = synth.reference_tree(n, 31) if bif else "((t0,t1,t2),(t3,t4),(t5,(t6,t7,t8)),(t9,t10));"
    trees = synth.tree_set(n, 40, 32, collapse=0.15)
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    o.score()
    want = o.scores_by_bipartition()
    ref = flatten.flatten_reference(ref_nw)
    names = ref.names
    for r in range(world):
        lq = np.load(tmp_path / f"lq{r}.npy")
        qp = np.load(tmp_path / f"qp{r}.npy") if bif else None
        eqp = np.load(tmp_path / f"eqp{r}.npy") if bif else None
        got = {}
        for e in range(ref.n_nodes - 1):
            below = frozenset(x.name for x in newick.preorder(ref.nodes[e + 1]) if x.is_leaf)
            if len(below) <= 1 or len(below) >= n - 1:
                continue
            other = frozenset(names) - below
            key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
            got[key] = (lq[e + 1], None if qp is None else qp[e + 1], None if eqp is None else eqp[e + 1])
        assert set(got) == set(want)
        for k in got:
            assert got[k] == want[k], (r, sorted(k), got[k], want[k])


def _table_shard_worker(rank, world, port, n_shards, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulate
    from quartetscores_amd import distributed, flatten, ranks, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 12, 50
    ref_nw = synth.reference_tree(n, 41)
    ref = flatten.flatten_reference(ref_nw)
    trees = synth.tree_set(n, m, 42, collapse=0.2, dropout=0.1)
    batch = flatten.flatten_eval_trees(trees, ref.name_to_id)
    full = emulate.counts_from_batch(batch, n)
    opened = []

    def open_shard(k):
        # shard k of K by the largest id: the tuples [C(d_lo,4), C(d_hi,4)) of the whole table (every "rank" counted all trees)
        if k is None:
            return emulate.ScoreEmu(ref, full[:0], 0)
        d_lo, d_hi = distributed.shard_of_largest_id(n, n_shards, k)
        r_lo, r_hi = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)
        opened.append(k)
        return emulate.ScoreEmu(ref, full[r_lo:r_hi], r_lo)
    mine = distributed.shards_of_rank(n_shards, world, rank)
    lq, qp, eqp, is_bif = distributed.score_table_shards(open_shard, mine, ref, device=torch.device("cpu"))
    # a rank with one shard keeps it between the rounds; with several every shard is opened once per round
    assert opened == (mine if len(mine) == 1 else mine + mine), (rank, opened)
    np.save(os.path.join(out_dir, f"lq{rank}.npy"), lq)
    np.save(os.path.join(out_dir, f"qp{rank}.npy"), qp)
    np.save(os.path.join(out_dir, f"eqp{rank}.npy"), eqp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_shards", [2, 3, 5])
def test_score_table_shards_over_two_ranks(tmp_path, n_shards):
    """distributed.score_table_shards (BASELINE configs[4]'s mode: K table shards by largest taxon id, shard s on rank
    s mod world, no table collective) over 2 gloo ranks with 2, 3 and 5 shards -- ranks own different numbers of shards,
    the candidate gather is padded -- gives the oracle's scores on every rank (numpy emulation of the score passes,
    the library's own host finish). Reference scheme: QuartetScoreComputer.hpp:212-371."""
    from oracle_api import Oracle
    from quartetscores_amd import flatten, newick, synth
    world = 2
    mp.spawn(_table_shard_worker, args=(world, _free_port(), n_shards, str(tmp_path)), nprocs=world, join=True)
    n = 12
    ref_nw = synth.reference_tree(n, 41)
    trees = synth.tree_set(n, 50, 42, collapse=0.2, dropout=0.1)
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    o.score()
    want = o.scores_by_bipartition()
    ref = flatten.flatten_reference(ref_nw)
    names = ref.names
    for r in range(world):
        lq, qp, eqp = (np.load(tmp_path / f"{x}{r}.npy") for x in ("lq", "qp", "eqp"))
        got = {}
        for e in range(ref.n_nodes - 1):
            below = frozenset(x.name for x in newick.preorder(ref.nodes[e + 1]) if x.is_leaf)
            if len(below) <= 1 or len(below) >= n - 1:
                continue
            other = frozenset(names) - below
            key = below if (len(below) < len(other) or (len(below) == len(other) and min(names) not in below)) else other
            got[key] = (lq[e + 1], qp[e + 1], eqp[e + 1])
        assert set(got) == set(want)
        for k in got:
            assert got[k] == want[k], (r, sorted(k), got[k], want[k])


def _both_modes_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulate
    from quartetscores_amd import distributed, flatten, ranks, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 13, 46
    ref = flatten.flatten_reference(synth.reference_tree(n, 51))
    batch = flatten.flatten_eval_trees(synth.tree_set(n, m, 52, collapse=0.1), ref.name_to_id)
    nq = ranks.n_quartets(n)
    cpu = torch.device("cpu")
    # tree mode: trees / N per rank into a full table, reduce-scatter, every rank scores the shard it received
    lo, hi = distributed.shard_range(m, world, rank)
    _t, words = distributed.scatter_layout(nq, world, 32)
    send = torch.zeros(world * words, dtype=torch.int32)
    send.numpy().view(np.uint32)[: nq * 3].reshape(nq, 3)[...] = emulate.counts_from_batch(batch.slice(lo, hi), n).astype(np.uint32)
    recv = torch.zeros(words, dtype=torch.int32)
    distributed.reduce_scatter_table(send, recv)
    r_lo, n_own = distributed.scatter_owned(nq, world, rank, 32)
    shard = recv.numpy().view(np.uint32)[: n_own * 3].reshape(n_own, 3).astype(np.uint64)
    tree_scores = distributed.score_sharded(emulate.ScoreEmu(ref, shard, r_lo), ref, device=cpu)
    # table mode: ALL trees per rank into the rank's shard by largest taxon id (cost-balanced bounds), no table collective
    d_lo, d_hi = distributed.shard_of_largest_id(n, world, rank, by="cost")
    full = emulate.counts_from_batch(batch, n)
    q_lo, q_hi = ranks.n_quartets(d_lo), ranks.n_quartets(d_hi)

    def open_shard(k):
        return emulate.ScoreEmu(ref, full[q_lo:q_hi] if k is not None else full[:0], q_lo if k is not None else 0)
    table_scores = distributed.score_table_shards(open_shard, [rank], ref, device=cpu)
    for name, (lq, qp, eqp, _bif) in (("tree", tree_scores), ("table", table_scores)):
        np.save(os.path.join(out_dir, f"{name}{rank}.npy"), np.stack([lq, qp, eqp]))
    dist.barrier()
    dist.destroy_process_group()


def test_tree_and_table_sharded_modes_give_identical_scores(tmp_path):
    """bench.py --mode tree | table, `QuartetScores --gpus N --mode ...` over 2 gloo ranks on the same trees: the
    tree-sharded route (trees / N per rank, reduce-scatter of the table, score_sharded) and the table-sharded route (all trees per
    rank, shard by largest taxon id with the cost-balanced bounds of qs_shard_bounds, score_table_shards) end with the SAME
    LQ/QP/EQP-IC, bit for bit, on every rank. Counting = the numpy emulation; scoring host code = the library's."""
    world = 2
    mp.spawn(_both_modes_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref_scores = np.load(tmp_path / "tree0.npy")
    assert np.isfinite(ref_scores[0][1:]).any()
    for name in ("tree", "table"):
        for r in range(world):
            got = np.load(tmp_path / f"{name}{r}.npy")
            assert got.shape == ref_scores.shape and (got.view(np.int64) == ref_scores.view(np.int64)).all(), (name, r)


def test_shard_bounds_cover_the_table_and_balance_what_they_claim():
    """qs_shard_bounds (host-only): contiguous shards by largest taxon id (quartet_lookup_table.hpp:161-165: the rank's leading
    term is C(s3,4)); by tuples = the Python arithmetic of shard_of_largest_id; by cost = no shard above the bound the bisection
    found, and a smaller largest shard (in the cost model) than the tuple-balanced cut."""
    from quartetscores_amd import distributed

    def tiles(c):
        t = (c + 7) // 8
        return (t * t) // 4 + (t + 1) // 2 if c >= 2 else 0

    def cost(d_lo, d_hi, alpha=4.0):
        d_lo = max(d_lo, 3)
        tot, d1 = 0.0, d_hi
        while d1 > d_lo:
            d0 = max(d1 - 8, d_lo)
            for c in range(2, d1 - 1):
                tot += tiles(c) * (alpha + ((d1 - d0) if c < d0 else (d1 - 1 - c)))
            d1 = d0
        return tot
    for n, k in ((512, 8), (256, 8), (128, 4), (97, 3), (33, 2), (8, 4), (4, 2)):
        c4 = distributed.shard_bounds(n, k, "c4")
        by_cost = distributed.shard_bounds(n, k, "cost")
        assert c4 == [distributed.shard_of_largest_id(n, k, r)[0] for r in range(k)] + [n]
        for b in (c4, by_cost):
            assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:])) and len(b) == k + 1
        if n >= 33:
            worst_c4 = max(cost(lo, hi) for lo, hi in zip(c4, c4[1:]))
            worst = max(cost(lo, hi) for lo, hi in zip(by_cost, by_cost[1:]))
            assert worst <= worst_c4 * (1 + 1e-9), (n, k, by_cost, c4)
            assert worst <= 1.12 * cost(0, n) / k or n < 256, (n, k, worst / (cost(0, n) / k))   # (every shard pays for one partial d-block)
    assert distributed.shard_bounds(512, 8, "cost") == [0, 303, 362, 402, 433, 457, 478, 496, 512]
