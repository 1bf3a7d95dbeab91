#!/bin/bash
# round-3 GPU session 35: depth-class merge threshold with the new launch plan
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zt; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { w="$1"; shift; echo "== $w | $*" | tee -a "$OUT/class_pct.txt"; env "$@" timeout -k 10 300 tools/bin/count_bench $w 3 $LIB 2>&1 | tail -1 | cut -c60-200 | tee -a "$OUT/class_pct.txt"; }
for p in 10 30 10 30; do run "512 10000 32" CB_CLASS_PCT=$p; done
for p in 10 30; do run "512 30000 32" CB_CLASS_PCT=$p; done
