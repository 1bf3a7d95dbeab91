D=$(mktemp -d)
python3 - 512 10000 $D <<'PY'
import sys
sys.path.insert(0, ".")
from quartetscores_amd import synth
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
open(d + "/ref.nwk", "w").write(synth.reference_tree(n, 4000) + "\n")
base = synth.tree_set(n, 2000, 4001)
with open(d + "/eval.nwk", "w") as f:
    for i in range(m):
        f.write(base[i % len(base)] + "\n")
PY
for t in 8 0; do quartetscores_amd/bin/QuartetScores -r $D/ref.nwk -e $D/eval.nwk -o $D/out$t.nwk -t $t -v 2>&1 | tail -40; done
rm -rf $D
