#!/bin/bash
# A/B of compile-time variants of the general / partial instances against the product library: tools/r06_gen_probe.sh <outdir> <probe lib names...>
out=${1:?outdir}; shift
mkdir -p "$out"
common="--taxa 512 --trees 1500 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-score --secondary 0"
for wl in "--collapse 0.2" "--collapse 0.2 --dropout 0.1" "--dropout 0.1" "--mixed"; do
  tag=$(echo "$wl" | tr -d ' -' | tr '.' 'p')
  for lib in product "$@"; do
    if [ $lib = product ]; then unset QS_PY_LIB; else export QS_PY_LIB=$PWD/quartetscores_amd/lib/libqs_probe_$lib.so; fi
    python bench.py $common $wl > "$out/bench_${tag}_${lib}.json" 2> "$out/bench_${tag}_${lib}.err" || echo "FAILED $wl $lib"
    python - "$out/bench_${tag}_${lib}.json" "$wl" "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:30s} {sys.argv[3]:12s} {d['ms_per_step']:.2f} ms  {d['value']:.3e}  swar-equal {d['config']['parity_bitslice_equals_swar_impl']}  {d['config']['algo'][:70]}")
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
  done
done
