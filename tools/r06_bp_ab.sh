#!/bin/bash
# A/B: a lone binary_partial class through its one-class instance (QS_TUNE_FUSE_CLASSES = 1) or through the fused binary kernel (= 2)
out=${1:-gpurun_out/r06_bp}; mkdir -p "$out"
common="--taxa 512 --trees 1500 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-score --secondary 0 --dropout 0.1"
for rep in 1 2; do for fuse in 1 2; do
  QS_PY_TUNING="18=$fuse" python bench.py $common > "$out/bench_fuse${fuse}_$rep.json" 2> "$out/bench_fuse${fuse}_$rep.err"
  python - "$out/bench_fuse${fuse}_$rep.json" "$fuse" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"fuse={sys.argv[2]}  {d['value']:.3e} q/s  {d['ms_per_step']:.2f} ms  {d['config']['algo'][:100]}  impl_match {d['config']['parity_bitslice_equals_swar_impl']}")
PY
done; done
