#!/bin/bash
# round-3 GPU session 31: slice size and tile order of the count kernel once more on the final kernel (512 taxa x 10000 trees)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zm; mkdir -p "$OUT"
LIB=$(find . -name libquartetscores_hip.so | head -1)
run() { echo "== $*" | tee -a "$OUT/count_sweep.txt"; env "$@" timeout -k 10 120 tools/bin/count_bench 512 10000 32 3 $LIB 2>&1 | tail -2 | cut -c1-200 | tee -a "$OUT/count_sweep.txt"; }
run CB_X=default
run CB_SLICE_BYTES=168000000
run CB_SLICE_BYTES=252000000
run CB_SLICE_BYTES=420000000
run CB_SLICE_BYTES=700000000
for to in $((2 | 16 << 16)) $((8 | 16 << 16)) $((4 | 8 << 16)) $((4 | 32 << 16)) $((4 | 64 << 16)) $((6 | 16 << 16)) $((3 | 16 << 16)); do run CB_TILE_ORDER=$to; done
run CB_X=default_again
