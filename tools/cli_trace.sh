#!/bin/bash
# The single-GPU CLI's wall time with its own pipeline stamps (--trace), on BASELINE configs[2]-shaped input:
#   tools/cli_trace.sh [taxa] [trees] [threads] [repetitions]   -> stdout: per run wall, phases, trace stamps
set -u
N=${1:-512}; M=${2:-10000}; T=${3:-8}; R=${4:-3}
D=$(mktemp -d)
python3 - "$N" "$M" "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest as ni
n, m, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
open(d + "/ref.nwk", "wb").write(ni.synth_trees(n, 1, 2000))
open(d + "/eval.nwk", "wb").write(ni.synth_trees(n, m, 2001))
PY
ls -la $D
for r in $(seq 1 $R); do
  rm -f $D/out.nwk; sleep 3
  s=$(date +%s.%N)
  quartetscores_amd/bin/QuartetScores -r $D/ref.nwk -e $D/eval.nwk -o $D/out.nwk -t $T --trace > $D/stdout.txt 2> $D/stderr.txt
  e=$(date +%s.%N)
  echo "== run $r: wall $(python3 -c "print(round(($e-$s)*1000,1))") ms, -t $T"
  grep -E "It took|Elapsed" $D/stdout.txt
  cat $D/stderr.txt
done
rm -rf $D
