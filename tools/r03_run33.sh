#!/bin/bash
# round-3 GPU session 33: how large may a panel slice get? (tree groups per slice; chunk 2)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zo; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { w="$1"; shift; echo "== $w | $*" | tee -a "$OUT/count_sweep3.txt"; env "$@" timeout -k 10 300 tools/bin/count_bench $w 2 $LIB 2>&1 | tail -1 | cut -c60-200 | tee -a "$OUT/count_sweep3.txt"; }
T2=$((2 | 16 << 16))
# 256 taxa: group bytes (B = 4: 4 words) ~ 522 KB, (B = 5) 653 KB
run "256 100000 32" CB_X=default
for g in 256 512 1024 2048 4096; do run "256 100000 32" CB_SLICE_BYTES=$((g * 653000)) CB_TILE_ORDER=$T2; done
# 512 taxa x 30000 trees: group bytes 2.62 MB
run "512 30000 32" CB_X=default
for g in 256 512 1024; do run "512 30000 32" CB_SLICE_BYTES=$((g * 2620000)) CB_TILE_ORDER=$T2; done
# chunk 1 and cblock variants at the big slice
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((1 | 16 << 16))
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((2 | 32 << 16))
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((2 | 12 << 16))
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$T2
