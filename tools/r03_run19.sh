#!/bin/bash
# round-3 GPU session 19: full GPU suite after the scoring / ingest / shard changes + scoring timings
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3t; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 500 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 > "$OUT/score_single_read.txt" 2>&1; grep -v chunk "$OUT/score_single_read.txt"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest gpu rc $?" | tee "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
