#!/bin/bash
# round-3 final evidence (run through gpurun from the repo root): counters + kernel stats of the bench workload, the full
# GPU test-suite, the default bench line, the out-of-core demo, CLI timings. Copies what is to be judged into profiles/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03_final; mkdir -p "$OUT"; export TMPDIR=/tmp
bash tools/pmc_collect.sh r03_cfg2 --config 2 > "$OUT/pmc_collect.log" 2>&1; echo "pmc rc $?" | tee "$OUT/summary.txt"
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest all rc $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 600 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc $?" | tee -a "$OUT/summary.txt"
tail -c 1500 "$OUT/bench_default.json"
timeout -k 10 600 bash tools/out_of_core_demo.sh 1200 40 > "$OUT/out_of_core.txt" 2>&1; echo "ooc rc $?" | tee -a "$OUT/summary.txt"
tail -8 "$OUT/out_of_core.txt"
timeout -k 10 300 bash tools/cli_timing.sh 512 10000 > "$OUT/cli_timing_512.txt" 2>&1; tail -8 "$OUT/cli_timing_512.txt"
