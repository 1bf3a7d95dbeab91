#!/bin/bash
# round 5, lease 4: clamp tests with the 128-leaf run cap; general-mode codegen A/B (opaque LDS columns, row-ahead prefetch);
# configs[2] in one slice and the configs[4] shard without the 8-tree class; new bench fields
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -k "depth_clamp or bench_launcher or gather_counts_bit_exact or mixed_batches or every_depth_width" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
B="--no-cpu-baseline --no-e2e --no-score"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
python3 - "$O/$name.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), c.get("algo"), c.get("box_issue_probe_ns_per_inst"))
print("   ", c.get("kernels_of_last_timed_step"), (c.get("depth_clamp") or {}).get("tree_quartet_corrections"))
PY
}
for w in "collapse:--trees 1500 --collapse 0.2" "colldrop:--trees 1500 --collapse 0.2 --dropout 0.1" "mixed:--trees 1500 --mixed"; do
  name=${w%%:*}; a=${w#*:}
  run bench_${name}_product python3 bench.py $B $a
  run bench_${name}_pf1 env QS_PY_LIB=$PWD/tools/bin/libqs_exppf1.so python3 bench.py $B $a
  run bench_${name}_pf2 env QS_PY_LIB=$PWD/tools/bin/libqs_exppf2.so python3 bench.py $B $a
done
run bench_default python3 bench.py $B
run bench_cfg4 python3 bench.py $B --config 4
run bench_dropout python3 bench.py $B --trees 1500 --dropout 0.1
