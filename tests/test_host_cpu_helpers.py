"""Helpers shared by CPU tests: concatenation of flattened tree batches."""
import numpy as np

from quartetscores_amd import flatten


def concat_batches(a, b):
    return flatten.TreeBatch(
        a.n_trees + b.n_trees,
        np.concatenate([a.leaf_off, b.leaf_off[1:] + a.leaf_off[-1]]).astype(np.uint32),
        np.concatenate([a.leaf_ids, b.leaf_ids]), np.concatenate([a.adj_depth, b.adj_depth]),
        np.concatenate([a.node_off, b.node_off[1:] + a.node_off[-1]]).astype(np.uint32),
        np.concatenate([a.rng_off, b.rng_off[1:] + a.rng_off[-1]]).astype(np.uint32),
        np.concatenate([a.ranges, b.ranges]))
