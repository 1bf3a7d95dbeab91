#!/bin/bash
# round-3 GPU session 2: balanced-placement yardstick; occupancy sensitivity (LDS pad -> 3 / 2 waves per SIMD), pure
# compute (zero-record + no barrier), stamp diagnostics
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3b; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 tools/bin/valu_yardstick 250 > "$OUT/valu_yardstick.txt" 2>&1 || echo "yardstick rc $?" >> "$OUT/errors.txt"
B=tools/bin
timeout -k 10 400 $B/count_bench 512 10000 32 3 $B/libqs_exp1.so $B/libqs_exp1w3.so $B/libqs_exp1w2.so $B/libqs_exp17.so $B/libqs_exp19.so $B/libqs_exp33.so $B/libqs_exp49.so $B/libqs_exp1.so > "$OUT/cb_512.txt" 2>&1 || echo "cb512 rc $?" >> "$OUT/errors.txt"
timeout -k 10 200 $B/count_bench 256 12500 32 3 $B/libqs_exp1.so $B/libqs_exp1w3.so $B/libqs_exp1w2.so $B/libqs_exp19.so $B/libqs_exp33.so > "$OUT/cb_256.txt" 2>&1 || echo "cb256 rc $?" >> "$OUT/errors.txt"
cat "$OUT/valu_yardstick.txt" "$OUT"/cb_*.txt
