#!/bin/bash
# round-3 final evidence, part A (through gpurun from the repo root): full GPU suite, the default bench line, the other
# configs / shapes / modes. Everything lands in gpurun_out/r03_final/ and is copied into profiles/r03_final/ afterwards.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03_final; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest all rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_gpu.log"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> "$OUT/pytest_gpu.log" 2>&1; tail -1 "$OUT/pytest_gpu.log"
timeout -k 10 600 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default rc $?" | tee -a "$OUT/summary.txt"
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc $?" | tee -a "$OUT/summary.txt"; }
run bench_cfg1 --config 1
run bench_cfg3_share --config 3 --trees 12500 --split-trees 0
run bench_cfg4_shard --config 4
run bench_ladder --shape ladder
run bench_nni --nni
run bench_cfg2_collapse --config 2 --trees 1500 --collapse 0.2 --steps 5
run bench_cfg2_dropout --config 2 --trees 1500 --dropout 0.1 --steps 5
QS_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e --config 3 --steps 3 > "$OUT/bench_cfg3_1gpu_forced_dist.json" 2> "$OUT/bench_cfg3_1gpu_forced_dist.err"; echo "forced dist rc $?" | tee -a "$OUT/summary.txt"
python3 - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1]); c = d["config"]
        print(f.split("/")[-1], "%.3e" % d["value"], round(d["ms_per_step"], 3), d["roofline"].get("frac"), c.get("algo"), "score", c.get("score_phase_ms"), c.get("score_mode"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
