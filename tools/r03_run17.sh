#!/bin/bash
# round-3 GPU session 17: automatic single-read scoring (pre-pass + estimate), tests + timings
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3r; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_read or automatic_scoring or score" > "$OUT/pytest_score.log" 2>&1; echo "pytest score rc $?" | tee "$OUT/summary.txt"
tail -5 "$OUT/pytest_score.log"
timeout -k 10 500 python3 tools/score_single_read.py > "$OUT/score_single_read.txt" 2>&1; cat "$OUT/score_single_read.txt"
