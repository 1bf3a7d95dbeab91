#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3n; mkdir -p "$OUT"
timeout -k 10 500 tools/bin/valu_yardstick 200 > "$OUT/valu_yardstick.txt" 2>&1; cat "$OUT/valu_yardstick.txt" | cut -c1-230
