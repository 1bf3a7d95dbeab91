#!/bin/bash
# round-3 GPU session 8: single-read scoring (tests + timing), full GPU suite, bench lines
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3i; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_read or score or two_cell or deep_instance" > "$OUT/pytest_score.log" 2>&1; echo "pytest score rc $?" | tee "$OUT/summary.txt"
tail -6 "$OUT/pytest_score.log"
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest all rc $?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"; echo "bench rc $?" | tee -a "$OUT/summary.txt"
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --nni --no-e2e > "$OUT/bench_cfg2_nni.json" 2> "$OUT/bench_cfg2_nni.err"; echo "bench nni rc $?" | tee -a "$OUT/summary.txt"
python3 - "$OUT" <<'PY'
import json, sys
for f in ("bench_cfg2.json", "bench_cfg2_nni.json"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + f).read().strip().split("\n")[-1])
        c = d["config"]
        print(f, d["ms_per_step"], c["score_phase_ms"], c["score_phase_ms_cold"], c["score_phases_ms"], d.get("e2e"))
    except Exception as e:
        print(f, "failed", e)
PY
