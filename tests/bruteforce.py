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
