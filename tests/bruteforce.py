"""Independent split-based quartet counter (numpy), used to cross-check the oracle.

Shares no code with oracle/qs_oracle.c or the HIP path: a tree displays ab|cd iff
one of its edge bipartitions has a,b on one side and c,d on the other.
"""
import itertools

import numpy as np


def parse_newick(s):
    s = s.strip()
    if s.endswith(";"):
        s = s[:-1]
    pos = 0

    def rec():
        nonlocal pos
        kids = []
        if pos < len(s) and s[pos] == "(":
            pos += 1
            kids.append(rec())
            while s[pos] == ",":
                pos += 1
                kids.append(rec())
            assert s[pos] == ")", s[pos:]
            pos += 1
        j = pos
        while j < len(s) and s[j] not in ",():;":
            j += 1
        name = s[pos:j].strip()
        pos = j
        if pos < len(s) and s[pos] == ":":
            pos += 1
            while pos < len(s) and s[pos] not in ",()":
                pos += 1
        return (name, kids)

    return rec()


def splits_of(tree, ids):
    """List of boolean membership vectors (len n) for every edge of the tree."""
    n = len(ids)
    out = []

    def rec(node):
        name, kids = node
        v = np.zeros(n, dtype=bool)
        if not kids:
            v[ids[name]] = True
        for k in kids:
            v |= rec(k)
        out.append(v)
        return v

    allv = rec(tree)
    return out, allv


def count_table(names, eval_newicks, mult=None):
    """(C(n,4),3) uint64 semantic counts; rank = C(d,4)+C(c,3)+C(b,2)+a for ids a<b<c<d
    (ids = index in `names`), slots = (ab|cd, ac|bd, ad|bc)."""
    n = len(names)
    ids = {nm: i for i, nm in enumerate(names)}
    quads = np.array([(a, b, c, d) for d in range(n) for c in range(d) for b in range(c) for a in range(b)],
                     dtype=np.int64).reshape(-1, 4)
    # the list above is in rank order by construction (d major ... a minor)
    T = np.zeros((len(quads), 3), dtype=np.uint64)
    for ti, nw in enumerate(eval_newicks):
        k = 1 if mult is None else int(mult[ti])
        splits, present = splits_of(parse_newick(nw), ids)
        S = np.array(splits, dtype=bool)  # (E, n)
        A, B, Cc, D = (S[:, quads[:, i]] for i in range(4))
        t0 = ((A == B) & (Cc == D) & (A != Cc)).any(axis=0)
        t1 = ((A == Cc) & (B == D) & (A != B)).any(axis=0)
        t2 = ((A == D) & (B == Cc) & (A != B)).any(axis=0)
        ok = present[quads].all(axis=1)
        T[:, 0] += (t0 & ok).astype(np.uint64) * np.uint64(k)
        T[:, 1] += (t1 & ok).astype(np.uint64) * np.uint64(k)
        T[:, 2] += (t2 & ok).astype(np.uint64) * np.uint64(k)
    return T


def rank_order_quads(n):
    return [(a, b, c, d) for d in range(n) for c in range(d) for b in range(c) for a in range(b)]


def all_quartets(n):
    return itertools.combinations(range(n), 4)


def quartet_counts_for(eval_newicks, names, quads, mult=None):
    """Semantic counts (len(quads), 3) uint64 of the given id 4-tuples (a,b,c,d) -> (#ab|cd, #ac|bd, #ad|bc), any order of
    ids, for large n: the same split test as count_table, with every leaf's split memberships packed into 64-bit
    words (a tree displays ab|cd iff some split has a,b on one side and c,d on the other)."""
    ids = {nm: i for i, nm in enumerate(names)}
    quads = np.asarray(quads, dtype=np.int64).reshape(-1, 4)
    T = np.zeros((len(quads), 3), dtype=np.uint64)
    for ti, nw in enumerate(eval_newicks):
        k = 1 if mult is None else int(mult[ti])
        splits, present = splits_of(parse_newick(nw), ids)
        S = np.array(splits, dtype=bool)                      # (E, n)
        E = S.shape[0]
        pad = (-E) % 64
        if pad:
            S = np.concatenate([S, np.zeros((pad, S.shape[1]), dtype=bool)])
        # member[x] = packed bits over the splits: bit e set iff leaf x is on the "below" side of split e
        member = np.ascontiguousarray(np.packbits(np.ascontiguousarray(S.T), axis=1, bitorder="little")).view(np.uint64)   # (n, words)
        A, B, Cc, D = (member[quads[:, i]] for i in range(4))
        ok = present[quads].all(axis=1)

        def displayed(p, q, r, s):   # pq|rs
            return ((~(p ^ q)) & (~(r ^ s)) & (p ^ r)).any(axis=1)
        # padding bits: all four are 0 there -> (p ^ r) = 0, never a hit
        for slot, hit in enumerate((displayed(A, B, Cc, D), displayed(A, Cc, B, D), displayed(A, D, B, Cc))):
            T[:, slot] += (hit & ok).astype(np.uint64) * np.uint64(k)
    return T
