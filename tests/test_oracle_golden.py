"""Pin the CPU oracle (oracle/qs_oracle.c) before trusting it:
  * SURVEY.md Appendix D known-answer vectors D1..D6 (tests/golden/appendix_d.json),
  * an independent split-based brute-force counter (tests/bruteforce.py),
  * closed-form invariants (tuple sums, fast == savemem/2, scale invariance).
CPU only.
"""
import numpy as np
import pytest

import bruteforce
from helpers import d5_trees, key_of, remap_table, ulp_diff
from oracle_api import Oracle, OracleError, canonical_split
from quartetscores_amd import synth


def by_label_number(o):
    """oracle counts re-expressed with taxon id = number in the label tN."""
    perm = np.zeros(o.n, dtype=np.int64)
    for lookup, nm in enumerate(o.names):
        perm[int(nm[1:])] = lookup
    return remap_table(o.counts(), perm)


def complement_key(o, key):
    return frozenset(set(o.names) - set(key))


def get_score(sc, o, key):
    if key in sc:
        return sc[key]
    return sc[complement_key(o, key)]


@pytest.mark.parametrize("savemem", [False, True])
def test_D1_counts_and_scores(golden, savemem):
    g = golden["D1"]
    o = Oracle(g["ref"])
    o.count("\n".join(g["eval"]), savemem=savemem, nthreads=1)
    T = by_label_number(o)
    f = 2 if savemem else 1  # reference savemem stores 2x
    for r, trip in g["counts"].items():
        assert list(T[int(r)]) == [f * x for x in trip]
    assert (T.sum(axis=1) == f * g["tuple_sum"]).all()
    r = np.arange(len(T), dtype=np.uint64)
    chk = int(((r + 1) * (T[:, 0] + 3 * T[:, 1] + 7 * T[:, 2])).sum())
    assert chk == f * g["checksum"]
    o.score()
    sc = o.scores_by_bipartition()
    assert len(sc) == len(g["scores"])
    for k, (lq, qp, eqp) in g["scores"].items():
        got = get_score(sc, o, key_of(k))
        assert got[0] == lq and got[1] == qp and got[2] == eqp, (k, got)


def test_D1_fast_threads_deterministic(golden):
    g = golden["D1"]
    o = Oracle(g["ref"])
    o.count("\n".join(g["eval"]), savemem=False, nthreads=1)
    a = o.counts().copy()
    o.count("\n".join(g["eval"]), savemem=False, nthreads=4)
    assert (o.counts() == a).all()


def test_D2_multifurcating_reference(golden):
    g = golden["D2"]
    o = Oracle(g["ref"])
    assert not o.bifurcating
    o.count("\n".join(golden["D1"]["eval"]))
    o.score()
    lq, qp, eqp = o.scores_by_edge()
    assert qp is None and eqp is None
    sc = o.scores_by_bipartition()
    assert len(sc) == len(g["lq"])
    for k, v in g["lq"].items():
        assert get_score(sc, o, key_of(k))[0] == v


def test_D3_mix_and_savemem_overflow(golden):
    g = golden["D3"]
    o = Oracle(g["ref"])
    ids = {nm: i for i, nm in enumerate(o.names)}
    q = [ids[x] for x in "abcd"]
    # fast mode at the reference's own width (m=200 -> u8)
    o.count("\n".join(g["eval"]), mult=g["mult"])
    assert o.lookup(*q) == tuple(g["occ_abcd"])
    o.score()
    sc = o.scores_by_bipartition()
    for k, v in g["scores"].items():
        assert list(get_score(sc, o, key_of(k))) == v
    # savemem with >= 16-bit counters: 2x counts, same scores
    o.count("\n".join(g["eval"]), mult=g["mult"], savemem=True, cint_bits=16)
    assert o.lookup(*q) == tuple(2 * x for x in g["occ_abcd"])
    o.score()
    sc = o.scores_by_bipartition()
    for k, v in g["scores"].items():
        assert list(get_score(sc, o, key_of(k))) == v
    # savemem at the reference's u8 width: 2*150 wraps (reference defect Q1)
    o.count("\n".join(g["eval"]), mult=g["mult"], savemem=True)
    assert o.lookup(*q) == tuple(g["savemem_u8_occ_abcd"])
    o.score()
    sc = o.scores_by_bipartition()
    for k, v in g["savemem_u8_scores"].items():
        assert list(get_score(sc, o, key_of(k))) == v


def test_D4_rooted_reference_quirk(golden):
    g = golden["D4"]
    o = Oracle(g["ref"])
    assert o.bifurcating  # degree-2 root passes is_bifurcating (quirk Q5)
    o.count("\n".join(golden["D1"]["eval"]))
    o.score()
    lq, qp, eqp = o.scores_by_edge()
    found = {}
    for e in range(o.n_edges):
        side = o.edge_side(e)
        names = frozenset(n for n, m in zip(o.names, side) if m)
        found[names] = (lq[e], qp[e], eqp[e])
    for k, v in g["scores_changed"].items():
        assert list(found[key_of(k)]) == v, (k, found[key_of(k)])
    # every other internal edge as in D1
    for k, v in golden["D1"]["scores"].items():
        if k in g["scores_changed"]:
            continue
        kk = key_of(k)
        got = found.get(kk) or found.get(frozenset(set(o.names) - kk))
        assert list(got) == v, k


def test_D5_u32_wrap_of_qp_sums(golden):
    g = golden["D5"]
    ref, alt = d5_trees(g["n"], g["block"])
    o = Oracle(ref)
    o.count(ref + "\n" + alt, mult=g["mult"], nthreads=8)
    central = frozenset(f"t{i}" for i in range(32, 64))
    o.score(nthreads=8)
    sc = o.scores_by_bipartition()
    got = get_score(sc, o, central)
    assert got[0] == g["lq"]
    assert got[1] == g["qp_wrap32"] and got[2] == g["qp_wrap32"]
    for k, v in sc.items():
        if k == central or complement_key(o, k) == central:
            continue
        assert v == (g["other_internal_edges"],) * 3
    o.score(nthreads=8, qp_exact64=True)
    got = get_score(o.scores_by_bipartition(), o, central)
    assert got[1] == g["qp_exact64"] and got[2] == g["qp_exact64"]


def test_D6_unknown_taxon_and_rooted_eval(golden):
    g = golden["D6"]
    o = Oracle(golden["D1"]["ref"])
    with pytest.raises(OracleError):
        o.count(g["bad_tree"])
    # rooted evaluation trees (first two top-level children grouped) -> identical table
    o.count("\n".join(golden["D1"]["eval"]))
    base = o.counts().copy()
    rooted = []
    for nw in golden["D1"]["eval"]:
        t = bruteforce.parse_newick(nw)
        kids = t[1]

        def w(node):
            return node[0] if not node[1] else "(" + ",".join(w(k) for k in node[1]) + ")"
        rooted.append("((" + w(kids[0]) + "," + w(kids[1]) + ")," + ",".join(w(k) for k in kids[2:]) + ");")
    o.count("\n".join(rooted))
    assert (o.counts() == base).all()


@pytest.mark.parametrize("n,m,dropout,collapse,seed", [(8, 20, 0.0, 0.0, 1), (12, 50, 0.3, 0.3, 2), (10, 30, 0.0, 0.5, 3),
                                                       (9, 25, 0.4, 0.0, 4)])
def test_oracle_matches_bruteforce(n, m, dropout, collapse, seed):
    ref = synth.reference_tree(n, seed)
    trees = synth.tree_set(n, m, 1000 + seed, dropout=dropout, collapse=collapse)
    o = Oracle(ref)
    o.count("\n".join(trees))
    bf = bruteforce.count_table(o.names, trees)
    assert (o.counts() == bf).all()
    o.count("\n".join(trees), savemem=True, cint_bits=32)
    assert (o.counts() == 2 * bf).all()


def test_scores_scale_invariant():
    n, m = 10, 12
    ref = synth.reference_tree(n, 7)
    trees = synth.tree_set(n, m, 8)
    o = Oracle(ref)
    o.count("\n".join(trees))
    o.score()
    a = o.scores_by_edge()
    o.count("\n".join(trees), mult=[3] * m, cint_bits=16)
    o.score()
    b = o.scores_by_edge()
    for x, y in zip(a, b):
        assert (ulp_diff(x, y) <= 4).all()  # p_i identical up to rounding of q/sum


def test_log_score_known_values():
    from oracle_api import lib
    L = lib()
    assert L.qso_log_score(0, 0, 0) == 0.0
    assert L.qso_log_score(5, 0, 0) == 1.0
    assert L.qso_log_score(0, 5, 0) == -1.0
    assert L.qso_log_score(150, 50, 0) == 0.48814049285708511
    assert L.qso_cint_bits_for_m(255) == 8 and L.qso_cint_bits_for_m(256) == 16
    assert L.qso_cint_bits_for_m(65535) == 16 and L.qso_cint_bits_for_m(65536) == 32


def test_canonical_split_helper():
    assert canonical_split(["a", "b", "c", "d"], [1, 1, 0, 0]) == frozenset(["c", "d"])


def test_config1_fixture_is_reproduced_by_the_oracle():
    """tests/golden/config1.json (BASELINE configs[0], frozen by make_config1_fixture.py): the oracle of today and
    the tree generator of today still give the committed table hash and the committed scores, bit for bit."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_config1_fixture", os.path.join(here, "make_config1_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.build()
    want = json.load(open(os.path.join(here, "config1.json")))
    assert got == want


def test_modes_fixture_is_reproduced_by_the_oracle():
    """tests/golden/modes.json (SURVEY 8(c) fixture F2: dropout + collapsed edges; a batch interleaving all four kernel modes),
    frozen by make_modes_fixture.py: the oracle and the tree generator of today still give the committed table hashes and scores."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_modes_fixture", os.path.join(here, "make_modes_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.build() == json.load(open(os.path.join(here, "modes.json")))
