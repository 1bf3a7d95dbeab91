"""Seeded synthetic tree sets.

Taxa are named t0..t{n-1}. A tree is built by uniformly random pairwise joining of
a shuffled taxon list until three subtrees remain, which gives an unrooted binary
tree with a trifurcating root, all n taxa, no branch lengths. PRNG: numpy PCG64,
seed = 1000 * config + tree-set id (documented so runs are reproducible).

Options used by the parity tests: taxon dropout (partial trees), edge collapse
(multifurcations), rooted output (degree-2 root), and the "reference + random
NNIs" distribution for concentrated counts.
"""
from __future__ import annotations

import numpy as np


def _join_random(items, rng, stop_at=3):
    items = list(items)
    while len(items) > stop_at:
        i, j = rng.choice(len(items), size=2, replace=False)
        if i > j:
            i, j = j, i
        b = items.pop(j)
        a = items.pop(i)
        items.append((a, b))
    return tuple(items)


def _to_newick(t):
    if isinstance(t, tuple):
        return "(" + ",".join(_to_newick(c) for c in t) + ")"
    return str(t)


def _collapse(t, rng, p, is_root=True):
    """Collapse each internal (non-root) edge with probability p."""
    if not isinstance(t, tuple):
        return t
    kids = []
    for c in t:
        c2 = _collapse(c, rng, p, False)
        if isinstance(c2, tuple) and rng.random() < p:
            kids.extend(c2)
        else:
            kids.append(c2)
    return tuple(kids)


def _drop(t, keep):
    if not isinstance(t, tuple):
        return t if t in keep else None
    kids = [k for k in (_drop(c, keep) for c in t) if k is not None]
    if not kids:
        return None
    if len(kids) == 1:
        return kids[0]
    return tuple(kids)


def random_tree(n, rng, names=None, dropout=0.0, collapse=0.0, rooted=False, min_taxa=4):
    names = names or [f"t{i}" for i in range(n)]
    order = list(rng.permutation(n))
    t = _join_random([names[i] for i in order], rng, stop_at=2 if rooted else 3)
    if dropout > 0:
        keep = {nm for nm in names if rng.random() >= dropout}
        while len(keep) < min(min_taxa, n):
            keep.add(names[int(rng.integers(n))])
        t = _drop(t, keep)
        if not isinstance(t, tuple):
            t = (t,)
        if len(t) == 2 and not rooted:
            # re-unroot: merge one child into the top level when possible
            a, b = t
            if isinstance(a, tuple):
                t = tuple(a) + (b,)
            elif isinstance(b, tuple):
                t = (a,) + tuple(b)
    if collapse > 0:
        t = _collapse(t, rng, collapse)
    return _to_newick(t) + ";"


def tree_set(n, m, seed, **kw):
    """m random trees on n taxa as a list of Newick strings."""
    rng = np.random.default_rng(seed)
    return [random_tree(n, rng, **kw) for _ in range(m)]


def reference_tree(n, seed):
    rng = np.random.default_rng(seed)
    return random_tree(n, rng)


# ---- reference + k random NNIs (k ~ Poisson(n/8)) ---------------------------------

def _parse_simple(s):
    """Parse the generator's own Newick dialect (names, parentheses, commas)."""
    s = s.strip().rstrip(";")
    pos = 0

    def rec():
        nonlocal pos
        if s[pos] == "(":
            pos += 1
            kids = [rec()]
            while s[pos] == ",":
                pos += 1
                kids.append(rec())
            assert s[pos] == ")"
            pos += 1
            return tuple(kids)
        j = pos
        while j < len(s) and s[j] not in ",()":
            j += 1
        name = s[pos:j]
        pos = j
        return name

    return rec()


def _nni(t, rng):
    """One random NNI on a nested-tuple binary tree with trifurcating root."""
    # collect internal edges as paths (list of child indices) to internal non-root nodes
    paths = []

    def walk(node, path):
        if isinstance(node, tuple):
            if path:
                paths.append(list(path))
            for i, c in enumerate(node):
                walk(c, path + [i])

    walk(t, [])
    if not paths:
        return t
    path = paths[int(rng.integers(len(paths)))]

    def rebuild(node, path):
        if len(path) == 1:
            i = path[0]
            child = node[i]
            sibs = [c for k, c in enumerate(node) if k != i]
            s = int(rng.integers(len(sibs)))
            g = int(rng.integers(len(child)))
            new_child = tuple(sibs[s] if k == g else c for k, c in enumerate(child))
            new_sibs = [child[g] if k == s else c for k, c in enumerate(sibs)]
            out = list(new_sibs)
            out.insert(i, new_child)
            return tuple(out)
        i = path[0]
        return tuple(rebuild(c, path[1:]) if k == i else c for k, c in enumerate(node))

    return rebuild(t, path)


def nni_tree_set(ref_newick, m, seed, mean_nni=None):
    rng = np.random.default_rng(seed)
    base = _parse_simple(ref_newick)
    n = ref_newick.count(",") + 1
    lam = mean_nni if mean_nni is not None else n / 8.0
    out = []
    for _ in range(m):
        t = base
        for _ in range(int(rng.poisson(lam))):
            t = _nni(t, rng)
        out.append(_to_newick(t) + ";")
    return out


def balanced_block(lo, hi):
    """Perfectly balanced binary subtree on t{lo}..t{hi-1} (recursive halving), SURVEY D5."""
    if hi - lo == 1:
        return f"t{lo}"
    mid = (lo + hi) // 2
    return "(" + balanced_block(lo, mid) + "," + balanced_block(mid, hi) + ")"
