#!/bin/bash
# round-3 GPU session 34: waves per count workgroup (2 / 4 / 8: the lockstep group), cblock with the new slices
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zs; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { w="$1"; lib="$2"; shift; shift; echo "== $w | $lib | $*" | tee -a "$OUT/count_sweep4.txt"; env "$@" timeout -k 10 300 tools/bin/count_bench $w 3 $lib 2>&1 | tail -1 | cut -c60-200 | tee -a "$OUT/count_sweep4.txt"; }
for lib in $LIB tools/bin/libqs_wg128.so tools/bin/libqs_wg512.so $LIB; do
  run "512 10000 32" $lib CB_X=1
  run "256 12500 32" $lib CB_X=1
done
for cb in 8 24 32 48 64; do run "512 10000 32" $LIB CB_TILE_ORDER=$((2 | cb << 16)); done
run "512 10000 32" $LIB CB_X=1
