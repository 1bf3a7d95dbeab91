#!/bin/bash
# A/B of compile-time variants of the count kernels on the default workload (configs[2]): tools/r06_headline_probe.sh <outdir> <probe lib names...>
out=${1:?outdir}; shift
mkdir -p "$out"
for rep in 1 2; do for lib in product "$@"; do
  if [ $lib = product ]; then unset QS_PY_LIB; else export QS_PY_LIB=$PWD/quartetscores_amd/lib/libqs_probe_$lib.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-score --secondary 0 > "$out/bench_${lib}_$rep.json" 2> "$out/bench_${lib}_$rep.err" || echo "FAILED $lib"
  python - "$out/bench_${lib}_$rep.json" "$lib" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} {d['ms_per_step']:.2f} ms  {d['value']:.3e}  frac {d['roofline']['frac']:.4f}  swar-equal {d['config']['parity_bitslice_equals_swar_impl']}  probe {d['config']['box_issue_probe_ns_per_inst']:.4f}")
PY
done; done
