#!/usr/bin/env python3
"""Generates tests/golden/rooted_compact.json: rooted reference trees (degree-2 root) in the reference's memory-efficient
table mode (`-s`). For the node pairs (root, v) the reference looks up quartets with a REPEATED id
(QuartetScoreComputer.hpp:393-396); its compact table sorts the ids and throws std::runtime_error when the index lies
behind the table (quartet_lookup_table.hpp:79-85), which ends the run. Expected output per case = that exception's what()
for the first throwing call in the reference's sequential order (-t 1), produced by the CPU oracle
(oracle/qs_oracle.c; its index / slot arithmetic on repeated ids and the exception text are pinned on the unmodified
reference header by tests/test_oracle_reftable.py). Cases: SURVEY Appendix D4's reference tree with D1's evaluation
trees, and three seeded random rooted references.

    python tests/golden/make_rooted_compact_fixture.py        # rewrites rooted_compact.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("D4", None, None), ("rooted9", 9, 71), ("rooted24", 24, 72), ("rooted41", 41, 73)]


def inputs(name, n, seed):
    import numpy as np
    from quartetscores_amd import synth
    if name == "D4":
        with open(os.path.join(HERE, "appendix_d.json")) as f:
            g = json.load(f)
        return g["D4"]["ref"], list(g["D1"]["eval"])
    return synth.random_tree(n, np.random.default_rng(seed), rooted=True), synth.tree_set(n, 20, seed + 100, collapse=0.1)


def build():
    from oracle_api import Oracle, OracleError
    out = {"_provenance": __doc__.strip().split("\n\n")[0].replace("\n", " ")}
    for name, n, seed in CASES:
        ref_nw, trees = inputs(name, n, seed)
        o = Oracle(ref_nw)
        o.count("\n".join(trees), savemem=True, cint_bits=16, nthreads=1)
        try:
            o.score(nthreads=1)
            what = None
        except OracleError as e:
            what = str(e)
        out[name] = {"ref": ref_nw, "eval": trees, "n_taxa": o.n, "savemem": True, "reference_throws": what}
        o.close()
    return out


if __name__ == "__main__":
    doc = build()
    with open(os.path.join(HERE, "rooted_compact.json"), "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    for k, v in doc.items():
        if k != "_provenance":
            print(k, v["reference_throws"])
