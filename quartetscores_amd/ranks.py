"""Vectorised rank <-> 4-set arithmetic of the count table (quartet_lookup_table.hpp:161-165)."""
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
    return binom(s3, 4) + binom(s2, 3) + binom(s1, 2) + np.asarray(s0, dtype=np.int64)


def _largest_with(binom_k, r, guess):
    x = guess.astype(np.int64)
    while True:
        too_big = binom_k(x) > r
        if not too_big.any():
            break
        x = np.where(too_big, x - 1, x)
    while True:
        can_grow = binom_k(x + 1) <= r
        if not can_grow.any():
            break
        x = np.where(can_grow, x + 1, x)
    return x


def unrank4_np(r):
    """ranks -> (N,4) sorted ids s0<s1<s2<s3."""
    r = np.asarray(r, dtype=np.int64).copy()
    d = _largest_with(lambda x: binom(x, 4), r, np.floor((24.0 * r + 1) ** 0.25 + 1.5))
    r -= binom(d, 4)
    c = _largest_with(lambda x: binom(x, 3), r, np.floor(np.cbrt(6.0 * r + 1) + 1.0))
    r -= binom(c, 3)
    b = _largest_with(lambda x: binom(x, 2), r, np.floor((1 + np.sqrt(1.0 + 8.0 * r)) / 2))
    a = r - binom(b, 2)
    return np.stack([a, b, c, d], axis=1)


def n_quartets(n: int) -> int:
    return n * (n - 1) * (n - 2) * (n - 3) // 24
