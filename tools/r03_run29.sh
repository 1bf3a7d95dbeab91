#!/bin/bash
# round-3 GPU session 29: waves per workgroup of the bundle score kernel's pass 1 x load mode (one chunk / next chunk ahead)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zi; mkdir -p "$OUT"; export TMPDIR=/tmp
for w in 5 6 7 8 10; do
  lib=tools/bin/libqs_exp0w$w.so; [ $w = 8 ] && lib=quartetscores_amd/libquartetscores_hip.so
  for t in "" "14=2"; do
    echo "== W1=$w QS_PY_TUNING=$t" | tee -a "$OUT/waves_x_load.txt"
    QS_LIB=$lib QS_PY_TUNING="$t" timeout -k 10 200 python3 tools/score_single_read.py 512:10000 2>&1 | grep "two passes\|automatic (default)" | cut -c1-180 | tee -a "$OUT/waves_x_load.txt"
  done
done
