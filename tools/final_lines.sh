#!/bin/bash
# The round's final bench lines on one lease: tools/final_lines.sh <out dir under gpurun_out>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-final_lines}; mkdir -p $O
show() { python3 - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), c.get("baseline_config"), c.get("algo"), c.get("box_issue_probe_ns_per_inst"))
PY
}
python3 bench.py > $O/bench_default.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }; show $O/bench_default.json
for c in 1 3 4; do python3 bench.py --config $c --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }; show $O/bench_cfg$c.json; done
B="--no-cpu-baseline --no-e2e --no-score --trees 1500"
for w in "collapse0.2:--collapse 0.2" "collapse0.2_dropout0.1:--collapse 0.2 --dropout 0.1" "dropout0.1:--dropout 0.1" "mixed:--mixed"; do
  name=${w%%:*}; a=${w#*:}
  python3 bench.py $B $a > $O/bench_$name.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }; show $O/bench_$name.json
done
bash tools/multi_path_on_one_gpu.sh
