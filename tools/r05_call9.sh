#!/bin/bash
# round 5, lease 9: binary_partial at 4 depth bits, 3 waves (133 VGPRs) against 4 waves (128 VGPRs, 6 spills); the whole GPU suite
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c9; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-score --no-impl-check"
one() { out=$1; shift; "$@" > $O/$out.json 2> $O/$out.err || { tail -20 $O/$out.err; exit 1; }
python3 - "$O/$out.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], c.get("algo"), c.get("box_issue_probe_ns_per_inst"))
PY
}
one dropout_product python3 bench.py $B --trees 1500 --dropout 0.1
one dropout_bp4 env QS_PY_LIB=$PWD/tools/bin/libqs_expBP4.so python3 bench.py $B --trees 1500 --dropout 0.1
one dropout_product2 python3 bench.py $B --trees 1500 --dropout 0.1
one dropout_bp4_2 env QS_PY_LIB=$PWD/tools/bin/libqs_expBP4.so python3 bench.py $B --trees 1500 --dropout 0.1
one mixed_product python3 bench.py $B --trees 1500 --mixed
one mixed_bp4 env QS_PY_LIB=$PWD/tools/bin/libqs_expBP4.so python3 bench.py $B --trees 1500 --mixed
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
