#!/bin/bash
# round-3 GPU session 32: larger panel slices (fewer table passes) x tile chunk, three workloads
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zn; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { w="$1"; shift; echo "== $w | $*" | tee -a "$OUT/count_sweep2.txt"; env "$@" timeout -k 10 200 tools/bin/count_bench $w 3 $LIB 2>&1 | tail -1 | cut -c60-200 | tee -a "$OUT/count_sweep2.txt"; }
run "512 10000 32" CB_X=default
run "512 10000 32" CB_SLICE_BYTES=700000000
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((2 | 16 << 16))
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((3 | 16 << 16))
run "512 10000 32" CB_SLICE_BYTES=700000000 CB_TILE_ORDER=$((2 | 8 << 16))
run "512 10000 32" CB_X=default
run "256 12500 32" CB_X=default
run "256 12500 32" CB_SLICE_BYTES=131000000
run "256 12500 32" CB_SLICE_BYTES=262000000
run "256 12500 32" CB_SLICE_BYTES=262000000 CB_TILE_ORDER=$((2 | 16 << 16))
run "1024 5000 16" CB_DLO=869 CB_DHI=896
run "1024 5000 16" CB_DLO=869 CB_DHI=896 CB_SLICE_BYTES=1700000000
run "1024 5000 16" CB_DLO=869 CB_DHI=896 CB_SLICE_BYTES=1700000000 CB_TILE_ORDER=$((2 | 16 << 16))
run "128 1000 32" CB_X=default
run "128 1000 32" CB_TILE_ORDER=$((2 | 16 << 16))
