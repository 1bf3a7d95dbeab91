#!/bin/bash
# round-3 GPU session 25: randomised soak of the single-read scoring against two passes
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zb; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 800 python3 tools/score_soak.py 60 7 > "$OUT/score_soak.txt" 2>&1; echo "soak rc $?" | tee "$OUT/summary.txt"
tail -15 "$OUT/score_soak.txt" | cut -c1-400
