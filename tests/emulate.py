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


# ---- scoring passes (qs_score.hip) on numpy arrays: stands in for a GPU context in the multi-process CPU tests ----

def _sortable(v):
    """order-preserving int64 encoding of float64 (qs_common.hpp f64_to_sortable)."""
    i = np.asarray(v, dtype=np.float64).view(np.int64).astype(object)
    out = np.array([int(x) if x >= 0 else (-(2 ** 63) - int(x)) for x in np.atleast_1d(i)], dtype=object)
    return np.array([int(x) for x in out], dtype=np.int64)


def _unsortable(s):
    s = int(s)
    i = s if s >= 0 else (-(2 ** 63) - s)
    return np.array([i], dtype=np.int64).view(np.float64)[0]


class ScoreEmu:
    """The three methods distributed.score_sharded calls on a context, for the tuples [rank_lo, rank_lo + len(T)) of a
    count table T ((k,3) array): classify every quartet like qs_score.hip::classify, per node pair 64-bit sums and the
    minimum QIC (pass 1), the gcd-reduced near-minimal triples (pass 2). The finish is the library's own host code."""
    KSORT_MAX = 0x7F7F7F7F7F7F7F7F

    def __init__(self, ref, T, rank_lo):
        from helpers import quads_in_rank_order
        self.ref, self.T, self.rank_lo = ref, np.asarray(T, dtype=np.int64), int(rank_lo)
        n, N = ref.n_taxa, ref.n_nodes
        par = ref.parent
        depth = np.zeros(N, dtype=np.int64)
        for v in range(N):       # preorder numbering: parents come first
            if par[v] >= 0:
                depth[v] = depth[par[v]] + 1
        nchild = np.bincount(par[par >= 0], minlength=N)
        inner_id = -np.ones(N, dtype=np.int64)
        inner_id[nchild > 0] = np.arange(int((nchild > 0).sum()))
        self.n_inner = int((nchild > 0).sum())
        lca = np.zeros((n, n), dtype=np.int64)
        for i in range(n):
            for j in range(i + 1, n):
                x, y = int(ref.leaf_node[i]), int(ref.leaf_node[j])
                while x != y:
                    if depth[x] >= depth[y]:
                        x = int(par[x])
                    else:
                        y = int(par[y])
                lca[i, j] = lca[j, i] = x
        self.lca, self.depth, self.inner_id = lca, depth, inner_id
        self.bif = int((nchild + (par >= 0)).max() - 1) == 2
        self.quads = quads_in_rank_order(n)[self.rank_lo:self.rank_lo + len(self.T)]

    def score_pair_slots(self, ref):
        return self.n_inner * self.n_inner

    def _classified(self):
        from quartetscores_amd.engine import log_score
        frame = 0 if self.bif else 1
        for (a, b, c, d), (n0, n1, n2) in zip(self.quads, self.T):
            e01, e12, e23 = self.lca[a, b], self.lca[b, c], self.lca[c, d]
            d01, d12, d23 = self.depth[e01], self.depth[e12], self.depth[e23]
            mx = max(d01, d23)
            if d12 < mx:
                q = (n0, n1, n2)
                j1 = e01 if d01 > d12 else e12
                j2 = e23 if d23 > d12 else e12
            elif d12 > mx:
                q = (n2, n1, n0) if frame == 0 else (n2, n0, n1)
                j1 = e12
                j2 = e01 if d01 >= d23 else e23
            else:
                continue
            i1, i2 = int(self.inner_id[j1]), int(self.inner_id[j2])
            key = min(i1, i2) * self.n_inner + max(i1, i2)
            yield key, tuple(int(x) for x in q), log_score(*(int(x) for x in q))

    def score_pass1(self, ref, sums, mins):
        s = np.zeros(sums.numel(), dtype=np.int64)
        m = np.full(mins.numel(), self.KSORT_MAX, dtype=np.int64)
        for key, q, qic in self._classified():
            s[3 * key:3 * key + 3] += q
            m[key] = min(int(m[key]), int(_sortable(qic)[0]))
        sums.copy_(__import__("torch").from_numpy(s))
        mins.copy_(__import__("torch").from_numpy(m))

    def score_pass2(self, ref, mins, cand):
        import math
        mn = mins.numpy()
        out = -np.ones(cand.numel(), dtype=np.int64)
        for key, q, qic in self._classified():
            if not qic <= _unsortable(mn[key]) + 1e-12:
                continue
            g = math.gcd(math.gcd(q[0], q[1]), q[2]) or 1
            packed = ((q[0] // g) << 42) | ((q[1] // g) << 21) | (q[2] // g)
            slots = out[8 * key:8 * key + 8]
            if packed in slots:
                continue
            free = np.where(slots == -1)[0]
            assert len(free), "candidate overflow"
            slots[free[0]] = packed
        cand.copy_(__import__("torch").from_numpy(out))

    def score_overflow(self, ref, mins, cand):
        return np.zeros((0, 4), dtype=np.int64)   # the emulation asserts instead of overflowing

    def score_finish(self, ref, sums_host, cand_host, flags=0, extra=None):
        from quartetscores_amd.engine import score_finish_host
        return score_finish_host(ref, sums_host, cand_host, flags, extra=extra)


def scores_from_table(ref, T, qp_exact64=False):
    """LQ-/QP-/EQP-IC per child node from a whole count table, by the closed form of SURVEY.md 3.3 in plain Python (exact
    log_score on every quartet, 32-bit wrap of the QP sums): the expected values for tables the oracle cannot produce by
    counting (hand-made tables with huge counts). Bifurcating references with a degree-3 root."""
    from quartetscores_amd.engine import log_score
    emu = ScoreEmu(ref, T, 0)
    N = ref.n_nodes
    inf = float("inf")
    lq = [inf] * N; qp = [inf] * N; eqp = [inf] * N
    per = {}
    for key, q, qic in emu._classified():
        s = per.setdefault(key, [0, 0, 0, inf])
        s[0] += q[0]; s[1] += q[1]; s[2] += q[2]
        s[3] = min(s[3], qic)
    inner_nodes = [v for v in range(N) if emu.inner_id[v] >= 0]
    par, depth = ref.parent, emu.depth
    for key in sorted(per):
        iu, iv = divmod(key, emu.n_inner)
        p1, p2, p3, lqmin = per[key]
        if not qp_exact64:
            p1 &= 0xFFFFFFFF; p2 &= 0xFFFFFFFF; p3 &= 0xFFFFFFFF
        val = log_score(p1, p2, p3)
        x, y = inner_nodes[iu], inner_nodes[iv]
        edges = []
        while x != y:
            if depth[x] >= depth[y]:
                edges.append(x); x = int(par[x])
            else:
                edges.append(y); y = int(par[y])
        for e in edges:
            lq[e] = min(lq[e], lqmin)
            eqp[e] = min(eqp[e], val)
        if len(edges) == 1:
            qp[edges[0]] = val
    return np.array(lq), np.array(qp), np.array(eqp)


# ---- depth clamp of the bit-sliced count classes (qs_abi.hip plan_depth_clamp, qs_count.hip clamp_fix_kernel) ----

def clamp_runs(adj_depth, L, clamp):
    """Maximal runs of tour-adjacent LCA depths >= clamp: (first leaf position, leaves) of every subtree whose root sits at
    depth `clamp` and that holds at least three leaves. With all labels cut at `clamp` the four-point test of a tree calls a
    quartet unresolved exactly when three of its leaves lie in one such subtree (otherwise it answers as before)."""
    runs, i, D = [], 0, [int(x) for x in adj_depth[: max(L - 1, 0)]]
    while i < len(D):
        if D[i] >= clamp:
            j = i
            while j < len(D) and D[j] >= clamp:
                j += 1
            if j - i + 1 >= 3:
                runs.append((i, j - i + 1))
            i = j
        else:
            i += 1
    return runs


def clamp_corrections(leaf_ids, adj_depth, clamp):
    """The (sorted quartet, table slot) list the correction kernel adds for ONE tree whose labels were cut at `clamp`:
    every quartet with >= 3 leaves in one run that the tree resolves, with its true topology."""
    def slot_of_pairing(x, y, z, w):      # qs_common.hpp: position of the minimum's partner among the sorted ids, minus 1
        partner = {x: y, y: x, z: w, w: z}[min(x, y, z, w)]
        return sorted((x, y, z, w)).index(partner) - 1
    L = len(leaf_ids)
    ids = [int(x) for x in leaf_ids]
    D = [int(x) for x in adj_depth[: max(L - 1, 0)]]
    out = []
    for i0, s in clamp_runs(adj_depth, L, clamp):
        rm = lambda a, b: min(D[i0 + a: i0 + b])      # LCA depth of run positions a < b
        for i in range(s):
            for j in range(i + 1, s):
                for k in range(j + 1, s):
                    mij, mjk = rm(i, j), rm(j, k)
                    x1, x2, x3 = ids[i0 + i], ids[i0 + j], ids[i0 + k]
                    # fourth leaf outside the run: the triple's cherry decides
                    if mij != mjk:
                        pair, third = ((x1, x2), x3) if mij > mjk else ((x2, x3), x1)
                        for p in list(range(0, i0)) + list(range(i0 + s, L)):
                            x = ids[p]
                            out.append((tuple(sorted((x1, x2, x3, x))), slot_of_pairing(pair[0], pair[1], third, x)))
                    # fourth leaf inside the run, behind k (every inside quartet once)
                    for l in range(k + 1, s):
                        mkl, x4 = rm(k, l), ids[i0 + l]
                        mx = max(mij, mkl)
                        if mjk < mx:
                            out.append((tuple(sorted((x1, x2, x3, x4))), slot_of_pairing(x1, x2, x3, x4)))
                        elif mjk > mx:
                            out.append((tuple(sorted((x1, x2, x3, x4))), slot_of_pairing(x1, x4, x2, x3)))
    return out
