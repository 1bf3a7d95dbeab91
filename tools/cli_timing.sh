#!/bin/bash
# End-to-end timing of the CLIs on the GPU box: tools/cli_timing.sh [taxa] [trees]
set -u
N=${1:-256}; M=${2:-20000}
D=$(mktemp -d)
python3 - "$N" "$M" "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from quartetscores_amd import synth
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
open(d + "/ref.nwk", "w").write(synth.reference_tree(n, 4000) + "\n")
base = synth.tree_set(n, min(m, 2000), 4001)
with open(d + "/eval.nwk", "w") as f:
    for i in range(m):
        f.write(base[i % len(base)] + "\n")
PY
ls -la $D/eval.nwk | awk '{print "eval file bytes:", $5}'
for t in 1 8 64; do
  rm -f $D/out.nwk
  /usr/bin/time -f "QuartetScores -t $t: %e s wall" quartetscores_amd/bin/QuartetScores -r $D/ref.nwk -e $D/eval.nwk -o $D/out.nwk -t $t | grep -E "Elapsed|took" | tr '\n' ' '; echo
done
rm -f $D/out2.nwk
/usr/bin/time -f "dist_cli (1 process): %e s wall" python -m quartetscores_amd.dist_cli -r $D/ref.nwk -e $D/eval.nwk -o $D/out2.nwk | grep -E "Elapsed|took" | tr '\n' ' '; echo
cmp $D/out.nwk $D/out2.nwk && echo "outputs identical"
rm -rf $D
