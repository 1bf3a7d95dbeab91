"""numpy emulation of the device algorithm's arithmetic (TEST-ONLY; never shipped or timed).

Lets the CPU suite check, without a GPU, (1) that the host flattening feeds the kernels what
they assume and (2) the SWAR field tricks of qs_count.hip::swar_step, bit for bit.
"""
import numpy as np

from helpers import quads_in_rank_order


def pair_depth_matrix(leaf_ids, adj_depth, n, flag):
    """M[x,y] = LCA depth of taxa x,y in one tree; `flag` where a taxon is missing."""
    L = len(leaf_ids)
    M = np.full((n, n), flag, dtype=np.int64)
    D = np.asarray(adj_depth[: max(L - 1, 0)], dtype=np.int64)
    ids = np.asarray(leaf_ids, dtype=np.int64)
    for i in range(L):
        run = np.minimum.accumulate(D[i:]) if i < L - 1 else np.zeros(0, dtype=np.int64)
        M[ids[i], ids[i + 1:]] = run
        M[ids[i + 1:], ids[i]] = run
    return M


def counts_from_batch(batch, n):
    """(C(n,4),3) semantic counts via the four-point test on LCA depths."""
    q = quads_in_rank_order(n)
    a, b, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    T = np.zeros((len(q), 3), dtype=np.uint64)
    FLAG = 1 << 20
    for t in range(batch.n_trees):
        lo, hi = int(batch.leaf_off[t]), int(batch.leaf_off[t + 1])
        M = pair_depth_matrix(batch.leaf_ids[lo:hi], batch.adj_depth[lo:hi], n, FLAG)
        ok = (M[a, b] < FLAG) & (M[c, d] < FLAG)
        s1, s2, s3 = M[a, b] + M[c, d], M[a, c] + M[b, d], M[a, d] + M[b, c]
        T[:, 0] += (ok & (s1 > s2)).astype(np.uint64)
        T[:, 1] += (ok & (s2 > s1)).astype(np.uint64)
        T[:, 2] += (ok & (s1 == s2) & (s3 > s1)).astype(np.uint64)
    return T


def swar_step(bits, mode, ab, cd, ac, bd, ad, bc):
    """Exact transcription of qs_count.hip::swar_step on uint32 words -> (n0, n1, n2) hit counts."""
    H, ONES = (0x80808080, 0x01010101) if bits == 8 else (0x80008000, 0x00010001)
    u = lambda x: np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    ab, cd, ac, bd, ad, bc = map(u, (ab, cd, ac, bd, ad, bc))
    s1h = (ab + cd + H) & 0xFFFFFFFF
    s2 = (ac + bd) & 0xFFFFFFFF
    x = (s1h - s2) & 0xFFFFFFFF
    t = (x - ONES) & 0xFFFFFFFF
    pop = lambda v: np.array([bin(int(z)).count("1") for z in np.atleast_1d(v)])
    if mode == 0:
        return pop(t & H), pop(~x & H & 0xFFFFFFFF), None
    s3 = (ad + bc) & 0xFFFFFFFF
    w = (s1h - s3) & 0xFFFFFFFF
    v = x & ~(t | w) & 0xFFFFFFFF
    hv = H
    if mode == 2:
        hv = ~((ab | cd) << 2) & H & 0xFFFFFFFF
    return pop(t & hv), pop(~x & hv & 0xFFFFFFFF), pop(v & hv)
