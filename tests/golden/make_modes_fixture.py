#!/usr/bin/env python3
"""Generates tests/golden/modes.json: evaluation-tree sets that exercise every kernel mode of the count path, counted and
scored by the CPU oracle (oracle/qs_oracle.c) and frozen -- SURVEY.md 8(c) fixture F2 (12 taxa, 50 trees, 30 % taxon dropout +
30 % collapsed edges) and a batch of 26 taxa x 120 trees that interleaves full binary trees, binary trees with missing taxa
(gene trees), multifurcating trees and trees that are both. Per case: the inputs (seeds + the first and last tree as a
check of the generator), sha256 of the count table in canonical form (rows in rank order of the NAME-sorted taxon ids, u32
little endian), its checksum, and every internal edge's LQ-/QP-/EQP-IC as hex doubles keyed by the sorted smaller side.

    python tests/golden/make_modes_fixture.py        # rewrites modes.json
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def trees_of(case):
    from quartetscores_amd import synth
    n = case["n"]
    if case["name"] == "F2":
        return synth.tree_set(n, case["m"], case["eval_seed"], dropout=0.3, collapse=0.3)
    k = case["m"] // 4
    sets = [synth.tree_set(n, k, case["eval_seed"]), synth.tree_set(n, k, case["eval_seed"] + 1, dropout=0.15),
            synth.tree_set(n, k, case["eval_seed"] + 2, collapse=0.2), synth.tree_set(n, k, case["eval_seed"] + 3, collapse=0.2, dropout=0.1)]
    return [sets[i % 4][i // 4] for i in range(4 * k)]


CASES = [{"name": "F2", "n": 12, "m": 50, "ref_seed": 2000, "eval_seed": 2001},
         {"name": "four_modes", "n": 26, "m": 120, "ref_seed": 2010, "eval_seed": 2011}]


def build():
    import numpy as np
    from helpers import remap_table
    from oracle_api import Oracle
    from quartetscores_amd import synth
    out = {"_provenance": "tests/golden/make_modes_fixture.py (oracle run in the build container)"}
    for case in CASES:
        n = case["n"]
        ref_nw = synth.reference_tree(n, case["ref_seed"])
        trees = trees_of(case)
        o = Oracle(ref_nw)
        o.count("\n".join(trees))
        o.score()
        names = list(o.names)
        perm = [names.index(f"t{i}") for i in range(n)]
        table = remap_table(o.counts(), perm).astype("<u4")
        scores = {",".join(sorted(k, key=lambda s: int(s[1:]))): [float(x).hex() for x in v]
                  for k, v in o.scores_by_bipartition().items()}
        e = dict(case)
        e.update({"ref": ref_nw, "first_tree": trees[0], "last_tree": trees[-1], "n_trees": len(trees),
                  "table_sha256": hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest(),
                  "checksum": int(table.astype(np.uint64).sum()), "max_tuple_sum": int(table.sum(axis=1).max()),
                  "scores_lq_qp_eqp_hex": dict(sorted(scores.items()))})
        out[case["name"]] = e
        o.close()
    return out


if __name__ == "__main__":
    doc = build()
    with open(os.path.join(HERE, "modes.json"), "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    for k, v in doc.items():
        if k != "_provenance":
            print(k, v["table_sha256"][:16], v["checksum"], len(v["scores_lq_qp_eqp_hex"]), "internal edges")
