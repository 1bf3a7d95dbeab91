"""Flatten trees into the plain arrays the C-ABI takes (include/quartetscores_hip.h).

Reference tree  -> qs_ref_tree   (parent[], leaf_node[]; lookup ids = depth-first leaf order,
                                  QuartetCounterLookup.hpp:252-258)
Evaluation trees -> qs_tree_batch (leaf_ids / adj_depth per tree = the tour of
                                  QuartetCounterLookup.hpp:211-221 plus LCA depths; and, for the
                                  scatter kernel, the circular leaf ranges behind every link of
                                  every inner node = subtreeLeafIndices, :117-121)

Every evaluation tree is re-rooted at its centre so that LCA depths stay small (the gather
kernel compares 4 trees per 32-bit lane while depths are <= 63); counts do not depend on
the rooting.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

from . import newick


class UnknownTaxonError(KeyError):
    """An evaluation tree holds a label the reference tree lacks (the reference program dies
    with an uncaught std::out_of_range, QuartetCounterLookup.hpp:218)."""


@dataclass
class RefTree:
    root: newick.Node
    nodes: List[newick.Node]        # preorder; index == node.index
    parent: np.ndarray              # int32[n_nodes]
    leaf_node: np.ndarray           # uint32[n_taxa]: node index of lookup id i
    names: List[str]                # lookup id -> taxon name
    name_to_id: Dict[str, int]

    @property
    def n_taxa(self):
        return len(self.names)

    @property
    def n_nodes(self):
        return len(self.nodes)


def flatten_reference(text_or_node) -> RefTree:
    root = newick.parse_tree(text_or_node) if isinstance(text_or_node, str) else text_or_node
    nodes = newick.preorder(root)
    for i, x in enumerate(nodes):
        x.index = i
    parent = np.array([x.parent.index if x.parent is not None else -1 for x in nodes], dtype=np.int32)
    leaves = [x for x in nodes if x.is_leaf]  # preorder restricted to leaves = depth-first leaf order
    names = [x.name for x in leaves]
    if len(set(names)) != len(names):
        raise ValueError("duplicate taxon name in the reference tree")
    leaf_node = np.array([x.index for x in leaves], dtype=np.uint32)
    return RefTree(root, nodes, parent, leaf_node, names, {nm: i for i, nm in enumerate(names)})


@dataclass
class TreeBatch:
    n_trees: int
    leaf_off: np.ndarray   # uint32[n_trees+1]
    leaf_ids: np.ndarray   # uint16
    adj_depth: np.ndarray  # uint16
    node_off: np.ndarray   # uint32[n_trees+1]
    rng_off: np.ndarray    # uint32[n_nodes+1]
    ranges: np.ndarray     # uint16[2*n_links]

    def slice(self, lo, hi) -> "TreeBatch":
        """Trees [lo, hi) as an independent batch (used to shard trees over GPUs)."""
        l0, l1 = int(self.leaf_off[lo]), int(self.leaf_off[hi])
        v0, v1 = int(self.node_off[lo]), int(self.node_off[hi])
        k0, k1 = int(self.rng_off[v0]), int(self.rng_off[v1])
        return TreeBatch(hi - lo, (self.leaf_off[lo:hi + 1] - l0).astype(np.uint32), self.leaf_ids[l0:l1].copy(),
                         self.adj_depth[l0:l1].copy(), (self.node_off[lo:hi + 1] - v0).astype(np.uint32),
                         (self.rng_off[v0:v1 + 1] - k0).astype(np.uint32), self.ranges[2 * k0:2 * k1].copy())


def _centre(adj: List[List[int]], start: int) -> int:
    def bfs(src):
        dist = [-1] * len(adj)
        prev = [-1] * len(adj)
        dist[src] = 0
        order = [src]
        for x in order:
            for y in adj[x]:
                if dist[y] < 0:
                    dist[y] = dist[x] + 1
                    prev[y] = x
                    order.append(y)
        far = order[-1]
        return far, dist, prev

    u, _, _ = bfs(start)
    v, dist, prev = bfs(u)
    path = [v]
    while path[-1] != u:
        path.append(prev[path[-1]])
    return path[len(path) // 2]


def flatten_tree(root: newick.Node, name_to_id: Dict[str, int], recentre: bool = True):
    """-> (leaf_ids, adj_depth, node_ranges) for one evaluation tree.

    node_ranges: list over inner nodes (degree >= 3) of lists of (start, end) circular ranges,
    one per link, parent side first, like the cycle link().next() walks
    (QuartetCounterLookup.hpp:170-176)."""
    nodes = newick.preorder(root)
    for i, x in enumerate(nodes):
        x.index = i
    adj: List[List[int]] = [[] for _ in nodes]
    for x in nodes:
        if x.parent is not None:
            adj[x.parent.index].append(x.index)  # filled below in child order
    # neighbour order = [parent, children...] (cyclic link order of the reference's tree model)
    adj = [([x.parent.index] if x.parent is not None else []) + [c.index for c in x.children] for x in nodes]
    r = root.index
    if recentre and len(nodes) > 2:
        r = _centre(adj, r)
        if len(adj[r]) == 1:  # centre landed on a leaf (2-leaf tree)
            r = adj[r][0]
    # iterative DFS from r following the cyclic neighbour order starting after the entry link
    L = 0
    leaf_ids: List[int] = []
    adj_depth: List[int] = []
    start = [0] * len(nodes)
    end = [0] * len(nodes)
    depth = [0] * len(nodes)
    order_children: List[List[int]] = [[] for _ in nodes]
    cur_min = 0
    stack = [(r, -1, 0)]
    depth[r] = 0
    while stack:
        x, par, k = stack.pop()
        nb = adj[x]
        if k == 0:
            start[x] = L
            if par >= 0:
                i = nb.index(par)
                order_children[x] = nb[i + 1:] + nb[:i]
            else:
                order_children[x] = list(nb)
            if not order_children[x]:
                nm = nodes[x].name
                if nm not in name_to_id:
                    raise UnknownTaxonError(nm)
                if L > 0:
                    adj_depth.append(cur_min)
                cur_min = 1 << 30
                leaf_ids.append(name_to_id[nm])
                L += 1
                end[x] = L
                continue
        ch = order_children[x]
        if k < len(ch):
            stack.append((x, par, k + 1))
            cur_min = min(cur_min, depth[x])
            depth[ch[k]] = depth[x] + 1
            stack.append((ch[k], x, 0))
        else:
            end[x] = L
    adj_depth.append(0)
    if L == 0:
        adj_depth = []
    node_ranges = []
    for x in range(len(nodes)):
        ch = order_children[x]
        nlinks = len(ch) + (1 if x != r else 0)
        if not ch or nlinks < 3 or L == 0:
            continue
        rl = []
        if x != r:
            rl.append((end[x] % L, start[x] % L))
        for c in ch:
            rl.append((start[c] % L, end[c] % L))
        node_ranges.append(rl)
    return leaf_ids, adj_depth, node_ranges


def flatten_eval_trees(trees: Sequence, name_to_id: Dict[str, int], recentre: bool = True) -> TreeBatch:
    """trees: iterable of Newick strings (one or many trees each) or parsed newick.Node."""
    leaf_off, node_off, rng_off = [0], [0], [0]
    ids: List[int] = []
    dep: List[int] = []
    rng: List[int] = []
    n_trees = 0

    def add(root):
        nonlocal n_trees
        li, ad, nr = flatten_tree(root, name_to_id, recentre)
        if len(set(li)) != len(li):
            raise ValueError(f"evaluation tree {n_trees}: duplicate taxon")
        ids.extend(li)
        dep.extend(ad)
        leaf_off.append(len(ids))
        for rl in nr:
            for (s, e) in rl:
                rng.extend((s, e))
            rng_off.append(len(rng) // 2)
        node_off.append(len(rng_off) - 1)
        n_trees += 1

    for t in trees:
        if isinstance(t, str):
            for root in newick.parse_trees(t):
                add(root)
        else:
            add(t)
    return TreeBatch(n_trees, np.asarray(leaf_off, dtype=np.uint32), np.asarray(ids, dtype=np.uint16),
                     np.asarray(dep, dtype=np.uint16), np.asarray(node_off, dtype=np.uint32),
                     np.asarray(rng_off, dtype=np.uint32), np.asarray(rng, dtype=np.uint16))
