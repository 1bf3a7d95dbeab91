#!/bin/bash
# round-3 GPU session 5: the cooperative count kernel (count_bitslice4_kernel): parity test, then A/B against the plain kernel
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3e; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cooperative or bitsliced or table_matches" > "$OUT/pytest_coop.log" 2>&1; echo "pytest rc $?" | tee "$OUT/summary.txt"
tail -15 "$OUT/pytest_coop.log"
B=tools/bin; P=quartetscores_amd/lib/libquartetscores_hip.so
run() { tag=$1; coop=$2; lib=$3; shift 3; CB_COOP=$coop timeout -k 10 300 $B/count_bench "$@" $lib > "$OUT/cb_$tag.txt" 2>&1 || echo "$tag rc $?" >> "$OUT/errors.txt"; }
run 512_plain 2 $P 512 10000 32 3
run 512_coop  1 $P 512 10000 32 3
run 512_coop5 1 $B/libqs_exp0c5.so 512 10000 32 3
run 256_plain 2 $P 256 12500 32 3
run 256_coop  1 $P 256 12500 32 3
run 128_plain 2 $P 128 1000 32 20
run 128_coop  1 $P 128 1000 32 20
CB_DLO=869 CB_DHI=896 CB_COOP=2 timeout -k 10 300 $B/count_bench 1024 5000 16 3 $P > "$OUT/cb_1024_plain.txt" 2>&1
CB_DLO=869 CB_DHI=896 CB_COOP=1 timeout -k 10 300 $B/count_bench 1024 5000 16 3 $P > "$OUT/cb_1024_coop.txt" 2>&1
CB_NNI=1 CB_COOP=2 timeout -k 10 300 $B/count_bench 512 10000 32 3 $P > "$OUT/cb_512nni_plain.txt" 2>&1
CB_NNI=1 CB_COOP=1 timeout -k 10 300 $B/count_bench 512 10000 32 3 $P > "$OUT/cb_512nni_coop.txt" 2>&1
for f in "$OUT"/cb_*.txt; do echo "$(basename $f): $(grep -h 'count ' $f | sed 's/.*count *\([0-9.]*\) ms.*checksum \(.*\)/\1 ms \2/')"; done | tee -a "$OUT/summary.txt"
