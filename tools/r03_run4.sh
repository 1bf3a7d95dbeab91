#!/bin/bash
# round-3 GPU session 4: row/y load-lane experiments; new CLI shard tests; bench line with the new fields; CLI timing
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3d; mkdir -p "$OUT"; export TMPDIR=/tmp
B=tools/bin; P=quartetscores_amd/lib/libquartetscores_hip.so
timeout -k 10 300 $B/count_bench 512 10000 32 3 $P $B/libqs_exp64.so $B/libqs_exp192.so $P > "$OUT/cb_512.txt" 2>&1 || echo "cb512 rc $?" >> "$OUT/errors.txt"
timeout -k 10 200 $B/count_bench 256 12500 32 3 $P $B/libqs_exp64.so $B/libqs_exp192.so > "$OUT/cb_256.txt" 2>&1 || echo "cb256 rc $?" >> "$OUT/errors.txt"
grep -h "count " "$OUT"/cb_*.txt | cut -c1-40,110-260
timeout -k 10 600 python3 -m pytest tests/test_cli.py -m gpu -x -q > "$OUT/pytest_cli.log" 2>&1; echo "pytest cli rc $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/pytest_cli.log"
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"; echo "bench rc $?" | tee -a "$OUT/summary.txt"
tail -c 3000 "$OUT/bench_cfg2.json"
timeout -k 10 300 bash tools/cli_timing.sh 512 10000 > "$OUT/cli_timing_512.txt" 2>&1; cat "$OUT/cli_timing_512.txt"
