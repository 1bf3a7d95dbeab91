#!/bin/bash
# round-3 GPU session 36: c-block 16 against 32 on ONE box, alternating (the boxes of the pool differ by ~2 %)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zv; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { w="$1"; shift; echo "== $w | $*" | tee -a "$OUT/cblock_ab.txt"; env "$@" timeout -k 10 300 tools/bin/count_bench $w 3 $LIB 2>&1 | tail -1 | cut -c60-200 | tee -a "$OUT/cblock_ab.txt"; }
for i in 1 2 3; do
  run "512 10000 32" CB_TILE_ORDER=$((2 | 16 << 16))
  run "512 10000 32" CB_TILE_ORDER=$((2 | 32 << 16))
  run "512 10000 32" CB_TILE_ORDER=$((4 | 16 << 16)) CB_SLICE_BYTES=336000000
done
