#!/bin/bash
# round 5, lease 10: final source -- the whole GPU suite, the default line in full, the general workloads, smoke
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c10; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
B="--no-cpu-baseline --no-e2e --no-score"
one() { out=$1; shift; "$@" > $O/$out.json 2> $O/$out.err || { tail -20 $O/$out.err; exit 1; }
python3 - "$O/$out.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), c.get("algo"), c.get("box_issue_probe_ns_per_inst"), c.get("kernels_of_last_timed_step"))
PY
}
one bench_default python3 -c "print(open('$O/bench_default.json').read().strip().splitlines()[-1])"
one bench_collapse0.2 python3 bench.py $B --trees 1500 --collapse 0.2
one bench_collapse0.2_dropout0.1 python3 bench.py $B --trees 1500 --collapse 0.2 --dropout 0.1
one bench_dropout0.1 python3 bench.py $B --trees 1500 --dropout 0.1
one bench_mixed python3 bench.py $B --trees 1500 --mixed
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
