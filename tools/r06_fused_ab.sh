#!/bin/bash
# Round 6 A/B of the fused launches (QS_TUNE_FUSE_CLASSES = key 18): mixed-shape workloads at 512 taxa x 1500 trees with one launch per
# depth-bits group (default) against one launch per class (round 5), plus the default line as a regression check of the refactored kernel.
#   bash tools/r06_fused_ab.sh <outdir>
out=${1:-gpurun_out/r06_fused}
mkdir -p "$out"
common="--taxa 512 --trees 1500 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-score"
for wl in "--mixed" "--collapse 0.2" "--collapse 0.2 --dropout 0.1" "--dropout 0.1"; do
  tag=$(echo "$wl" | tr -d ' -' | tr '.' 'p')
  for fuse in 1 0; do
    QS_PY_TUNING="18=$fuse" python bench.py $common $wl > "$out/bench_${tag}_fuse${fuse}.json" 2> "$out/bench_${tag}_fuse${fuse}.err" || echo "FAILED $wl fuse $fuse"
    python - "$out/bench_${tag}_fuse${fuse}.json" "$wl" "$fuse" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:34s} fuse={sys.argv[3]}  {d['value']:.3e} q/s  {d['ms_per_step']:.2f} ms  frac {d['roofline']['frac']:.3f}  launches {d['config']['count_launches_per_step']}  {d['config']['algo'][:110]}")
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
  done
done
