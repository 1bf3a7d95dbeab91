#!/bin/bash
# round 5, lease 2: depth clamp -- parity tests, then A/B of the default line and the general workloads on one box
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c2; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "depth_clamp or gather_counts_bit_exact or mixed_batches or every_depth_width or wire_format or depth_classes or deep_ladders or only_the_deep" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
B="--no-cpu-baseline --no-e2e --no-score"
python3 bench.py $B > $O/bench_default_clamp.json 2> $O/bench_default_clamp.err || { tail -20 $O/bench_default_clamp.err; exit 1; }
tail -c 300 $O/bench_default_clamp.json; echo
QS_PY_TUNING="17=0" python3 bench.py $B > $O/bench_default_noclamp.json 2> $O/bench_default_noclamp.err || { tail -20 $O/bench_default_noclamp.err; exit 1; }
tail -c 300 $O/bench_default_noclamp.json; echo
python3 bench.py $B --trees 1500 --collapse 0.2 > $O/bench_collapse.json 2> $O/bench_collapse.err || { tail -20 $O/bench_collapse.err; exit 1; }
python3 bench.py $B --trees 1500 --collapse 0.2 --dropout 0.1 > $O/bench_collapse_dropout.json 2> $O/bench_collapse_dropout.err || { tail -20 $O/bench_collapse_dropout.err; exit 1; }
python3 bench.py $B --trees 1500 --dropout 0.1 > $O/bench_dropout.json 2> $O/bench_dropout.err || { tail -20 $O/bench_dropout.err; exit 1; }
python3 bench.py $B --config 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err || { tail -20 $O/bench_cfg4.err; exit 1; }
QS_PY_TUNING="17=0" python3 bench.py $B --config 4 > $O/bench_cfg4_noclamp.json 2> $O/bench_cfg4_noclamp.err || { tail -20 $O/bench_cfg4_noclamp.err; exit 1; }
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"].get("algo"), d["config"].get("box_issue_probe_ns_per_inst"))
PY
done
