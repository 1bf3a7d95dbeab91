#!/usr/bin/env python3
"""Generates tests/golden/config1.json: BASELINE configs[0] (32 taxa, 200 random evaluation trees, seeds
1000/1001) counted and scored by the CPU oracle (oracle/qs_oracle.c, itself pinned against SURVEY Appendix D and
the reference's own quartet_lookup_table.hpp). Frozen so that a change of the oracle, of the tree generator or of
the GPU path shows up as a diff against committed data: sha256 of the count table in canonical form (rows in rank
order of the NAME-sorted taxon ids, u32 little endian) and every internal edge's scores as hex doubles keyed by
the sorted smaller side of its bipartition (fixture F7).

    python tests/golden/make_config1_fixture.py        # rewrites config1.json
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    import numpy as np
    from helpers import remap_table
    from oracle_api import Oracle
    from quartetscores_amd import synth
    n, m = 32, 200
    ref_nw = synth.reference_tree(n, 1000)
    trees = synth.tree_set(n, m, 1001)
    o = Oracle(ref_nw)
    o.count("\n".join(trees))
    o.score()
    names = list(o.names)
    perm = [names.index(f"t{i}") for i in range(n)]          # canonical id i = taxon t<i>
    table = remap_table(o.counts(), perm).astype("<u4")
    scores = {",".join(sorted(k, key=lambda s: int(s[1:]))): [float(x).hex() for x in v]
              for k, v in o.scores_by_bipartition().items()}
    return {
        "_provenance": "tests/golden/make_config1_fixture.py (oracle run in the build container)",
        "n": n, "m": m, "ref_seed": 1000, "eval_seed": 1001,
        "ref": ref_nw,
        "first_tree": trees[0], "last_tree": trees[-1],
        "table_sha256": hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest(),
        "tuple_sum": int(table[0].sum()),
        "checksum": int(table.astype(np.uint64).sum()),
        "first_rows": table[:4].tolist(),
        "scores_lq_qp_eqp_hex": dict(sorted(scores.items())),
    }


if __name__ == "__main__":
    out = build()
    with open(os.path.join(HERE, "config1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote config1.json:", out["table_sha256"], len(out["scores_lq_qp_eqp_hex"]), "internal edges")
