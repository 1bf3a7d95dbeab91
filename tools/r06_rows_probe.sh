#!/bin/bash
# Timing probe of a 4-d-row general tile WITHOUT building its geometry: libqs_probe_rows4_*.so run the general step
# over d slots 0..3 only (half the quartets of every tile: the tables are incomplete on purpose). If T(4 rows) < T(8 rows) / 2 a tile of 4
# rows -- twice the waves, each with half the counters -- could win; see profiles/r06_experiments.md 4.
out=${1:-gpurun_out/r06_rows}
mkdir -p "$out"
common="--taxa 512 --trees 1500 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-score --no-impl-check --secondary 0"
for wl in "--collapse 0.2" "--collapse 0.2 --dropout 0.1"; do
  tag=$(echo "$wl" | tr -d ' -' | tr '.' 'p')
  for lib in product rows4_w3 rows4_w4; do
    if [ $lib = product ]; then unset QS_PY_LIB; else export QS_PY_LIB=$PWD/quartetscores_amd/lib/libqs_probe_$lib.so; fi
    python bench.py $common $wl > "$out/bench_${tag}_${lib}.json" 2> "$out/bench_${tag}_${lib}.err" || echo "FAILED $wl $lib"
    python - "$out/bench_${tag}_${lib}.json" "$wl" "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:30s} {sys.argv[3]:10s} {d['ms_per_step']:.2f} ms per step, count kernels {d['config']['count_kernels_ms_per_step']:.2f} ms  {d['config']['algo'][:80]}")
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
  done
done
