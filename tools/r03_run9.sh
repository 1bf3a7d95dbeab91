#!/bin/bash
# round-3 GPU session 9: single-read scoring with plain bound loads: test + timing on random and NNI trees, 512 and 256 taxa
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3k; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_read or scores" > "$OUT/pytest_score.log" 2>&1; echo "pytest score rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_score.log"
for args in "" "--nni" "--config 3 --trees 12500" "--config 1"; do
  tag=$(echo "cfg2 $args" | tr -d ' -')
  timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-impl-check $args > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "bench $tag rc $?" | tee -a "$OUT/summary.txt"
done
python3 - "$OUT" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        c = d["config"]
        print(f.split("/")[-1], c["workload"][:40], "score", round(c["score_phase_ms"], 2), "cold", round(c["score_phase_ms_cold"], 2), c["score_phases_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
