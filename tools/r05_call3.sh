#!/bin/bash
# round 5, lease 3: whole GPU suite, then the clamp's breakdown (per-kernel events) at configs[2] and on a configs[4] shard
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c3; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -k "cooperative or bench_launcher or baseline_sizes" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
B="--no-cpu-baseline --no-e2e --no-score"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }
python3 - "$O/$name.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), c.get("algo"), c.get("box_issue_probe_ns_per_inst"))
print("   ", c.get("kernels_of_last_timed_step"), c.get("depth_clamp"))
PY
}
run bench_default python3 bench.py $B
run bench_default_oneslice python3 bench.py $B --slice-bytes 700000000
run bench_default_noclamp env QS_PY_TUNING="17=0" python3 bench.py $B
run bench_cfg4 python3 bench.py $B --config 4
run bench_cfg4_noclamp env QS_PY_TUNING="17=0" python3 bench.py $B --config 4
run bench_cfg3 python3 bench.py $B --config 3
run bench_cfg1 python3 bench.py $B --config 1
