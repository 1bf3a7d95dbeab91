#!/bin/bash
# Launch-order parameters of the count kernel's tiles on the default line (QS_TUNE_TILE_ORDER = chunk | c-block << 16 through the Python
# harness' QS_PY_TUNING): tools/tile_order_sweep.sh [bench.py args...]   (one lease; prints ms per step per setting)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/tile_order_sweep; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-score --no-impl-check --steps 8 --warmup 2 $*"
for v in "2 32" "1 32" "4 32" "2 16" "2 64" "4 64" "3 48" "2 32"; do
  set -- $v; val=$(( $1 | ($2 << 16) ))
  QS_PY_TUNING="4=$val" python3 bench.py $B > $O/chunk$1_cblock$2.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python3 - "$O/chunk$1_cblock$2.json" "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("chunk", sys.argv[2], "c-block", sys.argv[3], "%.2f ms"%d["ms_per_step"], "%.4g"%d["value"], c.get("kernels_of_last_timed_step"), c.get("box_issue_probe_ns_per_inst"))
PY
done
