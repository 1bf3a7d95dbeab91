#!/bin/bash
# round-3 GPU session 37: randomised soak of the count kernels with the new launch plan
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zw; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 800 python3 tools/count_soak.py 80 5 > "$OUT/count_soak.txt" 2>&1; echo "count soak rc $?" | tee "$OUT/summary.txt"
tail -6 "$OUT/count_soak.txt" | cut -c1-200
