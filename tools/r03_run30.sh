#!/bin/bash
# round-3 GPU session 30: more waves per workgroup for the logging pass 1
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zj; mkdir -p "$OUT"; export TMPDIR=/tmp
for w in 8 10 12 14 16; do
  lib=tools/bin/libqs_exp0w$w.so; [ $w = 8 ] && lib=$(find . -name libquartetscores_hip.so | head -1)
  echo "== W1=$w ($lib)" | tee -a "$OUT/waves.txt"
  QS_LIB=$lib timeout -k 10 200 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 2>&1 | grep "two passes\|automatic (default)" | cut -c1-180 | tee -a "$OUT/waves.txt"
done
