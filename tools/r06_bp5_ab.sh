#!/bin/bash
# A/B: binary_partial trees at 5 depth bits (clamp off: QS_TUNE_DEPTH_CLAMP = key 17 = 0) through the fused binary kernel on 4 waves per SIMD
# (product) against 3 waves (libqs_probe_bin5_w3.so) and against the one-class instances (QS_TUNE_FUSE_CLASSES = key 18 = 0)
out=${1:-gpurun_out/r06_bp5}; mkdir -p "$out"
common="--taxa 512 --trees 1500 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-score --secondary 0 --dropout 0.1"
run() { # tag, tuning, lib
  if [ -n "$3" ]; then export QS_PY_LIB=$PWD/quartetscores_amd/lib/$3; else unset QS_PY_LIB; fi
  QS_PY_TUNING="$2" python bench.py $common > "$out/bench_$1.json" 2> "$out/bench_$1.err"
  python - "$out/bench_$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:.3e} q/s  {d['ms_per_step']:.2f} ms  launches {d['config']['count_launches_per_step']}  {d['config']['algo'][:110]}  swar-equal {d['config']['parity_bitslice_equals_swar_impl']}")
PY
}
for rep in 1 2; do
  run fused_b5_4waves_$rep "17=0" ""
  run fused_b5_3waves_$rep "17=0" libqs_probe_bin5_w3.so
  run unfused_$rep "17=0,18=0" ""
done
