#!/bin/bash
# round-3 GPU session 21: tie filter of the logging pass (last logged triple per node pair)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3v; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_read or automatic_scoring or score or sharded" > "$OUT/pytest_score.log" 2>&1; echo "pytest score rc $?" | tee "$OUT/summary.txt"
tail -5 "$OUT/pytest_score.log"
timeout -k 10 500 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 256:12500:1 > "$OUT/score_single_read.txt" 2>&1; grep -v "chunk\|round 1/16\|round 1/32" "$OUT/score_single_read.txt" | cut -c1-230
