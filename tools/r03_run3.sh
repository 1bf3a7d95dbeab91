#!/bin/bash
# round-3 GPU session 3: yardstick v3; tile-order variants (c innermost in groups: workgroup = same (a,b,d) tile, 4 c's);
# GPU parity tests of the changed kernel
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3c; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 400 tools/bin/valu_yardstick 220 > "$OUT/valu_yardstick.txt" 2>&1 || echo "yardstick rc $?" >> "$OUT/errors.txt"
B=tools/bin; P=quartetscores_amd/lib/libquartetscores_hip.so
run() { # tag, order value, workload args...
  tag=$1; ord=$2; shift 2
  CB_TILE_ORDER=$ord timeout -k 10 300 $B/count_bench "$@" $P > "$OUT/cb_$tag.txt" 2>&1 || echo "$tag rc $?" >> "$OUT/errors.txt"
}
# chunk | cblock << 16 | cgroup << 32
run 512_default  $((4 + (16<<16)))            512 10000 32 3
run 512_g4       $((4 + (16<<16) + (4<<32)))  512 10000 32 3
run 512_g2       $((4 + (16<<16) + (2<<32)))  512 10000 32 3
run 512_g4_ch2   $((2 + (16<<16) + (4<<32)))  512 10000 32 3
run 512_g4_ch8   $((8 + (16<<16) + (4<<32)))  512 10000 32 3
run 512_g4_cb32  $((4 + (32<<16) + (4<<32)))  512 10000 32 3
run 512_g8_cb32  $((4 + (32<<16) + (8<<32)))  512 10000 32 3
run 256_default  $((4 + (16<<16)))            256 12500 32 3
run 256_g4       $((4 + (16<<16) + (4<<32)))  256 12500 32 3
run 128_default  $((4 + (16<<16)))            128 1000 32 20
run 128_g4       $((4 + (16<<16) + (4<<32)))  128 1000 32 20
grep -h "count " "$OUT"/cb_*.txt | cut -c1-20,100-260
for f in "$OUT"/cb_*.txt; do echo "$(basename $f): $(grep -h 'count ' $f | sed 's/.*count *\([0-9.]*\) ms.*checksum \(.*\)/\1 ms \2/')"; done > "$OUT/summary.txt"
cat "$OUT/summary.txt"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
