#!/bin/bash
# round 5, lease 5: what the general-mode step wants -- A: round-4 codegen (no opaque columns, nothing between reads and writes),
# B: + wave barrier, F: + wavefront fence, C: opaque columns only, G: opaque + fence (= product), H: opaque + row-ahead prefetch
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c5; mkdir -p $O
B="--no-cpu-baseline --no-e2e --no-score --no-impl-check"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
python3 - "$O/$name.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], c.get("algo"), c.get("box_issue_probe_ns_per_inst"))
PY
}
for w in "collapse:--trees 1500 --collapse 0.2" "colldrop:--trees 1500 --collapse 0.2 --dropout 0.1"; do
  name=${w%%:*}; a=${w#*:}
  for v in A B F C G H A; do
    run bench_${name}_$v env QS_PY_LIB=$PWD/tools/bin/libqs_exp$v.so python3 bench.py $B $a
  done
done
bash tools/cli_trace.sh 512 10000 8 3 > $O/cli_trace_512x10000_t8.txt 2>&1
tail -40 $O/cli_trace_512x10000_t8.txt
