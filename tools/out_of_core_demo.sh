#!/bin/bash
# A count table that does not fit the GPU: 1200 taxa, u16 cells = 515 GB through one 288 GB MI355X in shards
# (QuartetScores --table-shards 0 --spill recount). The evaluation trees are copies of the reference tree, so every
# internode must come out with lq-ic = qp-ic = eqp-ic = 1.
set -u
N=${1:-1200}; M=${2:-40}
D=$(mktemp -d)
python3 - "$N" "$M" "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from quartetscores_amd import synth
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ref = synth.reference_tree(n, 5000)
open(d + "/ref.nwk", "w").write(ref + "\n")
open(d + "/eval.nwk", "w").write((ref + "\n") * m)
PY
quartetscores_amd/bin/QuartetScores -r $D/ref.nwk -e $D/eval.nwk -o $D/out.nwk --table-shards 0 --spill recount 2>&1 | grep -vE "^Counting quartets"
python3 - "$D" <<'PY'
import re, sys
t = open(sys.argv[1] + "/out.nwk").read()
vals = re.findall(r"(qp-ic|lq-ic|eqp-ic):([-0-9.e+]+)", t)
bad = [v for v in vals if float(v[1]) != 1.0]
print("annotated internodes:", len(vals) // 3, "scores:", len(vals), "not equal to 1:", len(bad))
PY
rm -rf $D
