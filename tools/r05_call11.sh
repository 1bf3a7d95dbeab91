#!/bin/bash
# round 5, lease 11: final source -- the new class-rule test, the default line in full, the general workloads, PMC of all three
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c11; mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -q -k "join_the_larger_class or depth_clamp or mixed_batches or depth_classes" > $O/pytest_sel.log 2>&1 || { tail -40 $O/pytest_sel.log; exit 1; }
tail -2 $O/pytest_sel.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
B="--no-cpu-baseline --no-e2e --no-score"
show() { python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), c.get("algo"), c.get("box_issue_probe_ns_per_inst"), c.get("kernels_of_last_timed_step"))
PY
}
show $O/bench_default.json
for w in "collapse0.2:--collapse 0.2" "collapse0.2_dropout0.1:--collapse 0.2 --dropout 0.1" "dropout0.1:--dropout 0.1" "mixed:--mixed"; do
  name=${w%%:*}; a=${w#*:}
  python3 bench.py $B --trees 1500 $a > $O/bench_$name.json 2> $O/bench_$name.err || { tail -20 $O/bench_$name.err; exit 1; }
  show $O/bench_$name.json
done
export PMC_STEPS="--steps 3 --warmup 1"
bash tools/pmc_collect.sh r05_c11/pmc_cfg2
bash tools/pmc_collect.sh r05_c11/pmc_collapse --trees 1500 --collapse 0.2
bash tools/pmc_collect.sh r05_c11/pmc_collapse_dropout --trees 1500 --collapse 0.2 --dropout 0.1
