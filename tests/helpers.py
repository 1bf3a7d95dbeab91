"""Shared test helpers: table remapping between taxon-id orders, ulp distance, D5 trees."""
import numpy as np


def binom(n, k):
    n = np.asarray(n, dtype=np.int64)
    if k == 2:
        return n * (n - 1) // 2
    if k == 3:
        return n * (n - 1) * (n - 2) // 6
    if k == 4:
        return n * (n - 1) * (n - 2) * (n - 3) // 24
    raise ValueError(k)


def rank4(s0, s1, s2, s3):
    """rank of sorted ids s0<s1<s2<s3 (quartet_lookup_table.hpp:161-165)."""
    return binom(s3, 4) + binom(s2, 3) + binom(s1, 2) + np.asarray(s0, dtype=np.int64)


def quads_in_rank_order(n):
    q = np.array([(a, b, c, d) for d in range(n) for c in range(d) for b in range(c) for a in range(b)],
                 dtype=np.int64).reshape(-1, 4)
    return q


def remap_table(T_src, perm):
    """T_dst over target ids 0..n-1 where target id i is source id perm[i].

    T_* are (C(n,4),3) arrays in rank order with slots (s0s1|s2s3, s0s2|s1s3, s0s3|s1s2)."""
    perm = np.asarray(perm, dtype=np.int64)
    n = len(perm)
    q = quads_in_rank_order(n)                    # target quartets x<y<z<w
    s = perm[q]                                   # source ids, same column roles
    order = np.argsort(s, axis=1)
    ss = np.take_along_axis(s, order, axis=1)
    r = rank4(ss[:, 0], ss[:, 1], ss[:, 2], ss[:, 3])
    pos = np.argsort(order, axis=1)               # pos[:,j] = sorted position of column j
    out = np.zeros_like(T_src[: len(q)])
    # target slot k pairs column 0 with column k+1; source slot = partner position of the min - 1
    for k in range(3):
        partner_col = k + 1
        p0, pp = pos[:, 0], pos[:, partner_col]
        others = [c for c in (1, 2, 3) if c != partner_col]
        po1, po2 = pos[:, others[0]], pos[:, others[1]]
        # the pair containing sorted position 0
        in_first = (p0 == 0) | (pp == 0)
        partner_of_min = np.where(p0 == 0, pp, np.where(pp == 0, p0, np.where(po1 == 0, po2, po1)))
        del in_first
        src_slot = partner_of_min - 1
        out[:, k] = T_src[r, src_slot]
    return out


def ulp_diff(a, b):
    """Distance in units in the last place between two float64 arrays (inf==inf -> 0)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-(2 ** 63)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2 ** 63)) - ib, ib)
    return np.abs(ia - ib)


def d5_trees(n=64, block=16):
    from quartetscores_amd.synth import balanced_block
    X = [balanced_block(k * block, (k + 1) * block) for k in range(n // block)]
    ref = f"({X[0]},{X[1]},({X[2]},{X[3]}));"
    alt = f"({X[0]},{X[2]},({X[1]},{X[3]}));"
    return ref, alt


def key_of(csv_names):
    return frozenset(csv_names.split(","))
