#!/bin/bash
# round-3 GPU session 16: single-read scoring with a sampled pre-pass, timings
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3q; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 500 python3 tools/score_single_read.py > "$OUT/score_single_read.txt" 2>&1; cat "$OUT/score_single_read.txt"
