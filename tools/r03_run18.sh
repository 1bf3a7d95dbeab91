#!/bin/bash
# round-3 GPU session 18: automatic single-read scoring after calibration + cached accumulators: tests, timings, bench line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3s; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "single_read or automatic_scoring or score or cli" > "$OUT/pytest_score.log" 2>&1; echo "pytest score rc $?" | tee "$OUT/summary.txt"
tail -5 "$OUT/pytest_score.log"
timeout -k 10 500 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 1024:300 > "$OUT/score_single_read.txt" 2>&1; grep -v chunk "$OUT/score_single_read.txt"
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc $?"; cut -c1-1500 "$OUT/bench_default.json"
