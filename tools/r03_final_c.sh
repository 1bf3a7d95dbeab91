#!/bin/bash
# round-3 final refresh on the last commit: full GPU suite + smoke + the default bench line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03_final; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest all rc $?" | tee "$OUT/summary_c.txt"
tail -2 "$OUT/pytest_gpu.log"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> "$OUT/pytest_gpu.log" 2>&1; tail -1 "$OUT/pytest_gpu.log"
timeout -k 10 600 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default rc $?" | tee -a "$OUT/summary_c.txt"
python3 -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1]); c=d['config']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], c['score_phase_ms'], d['e2e']['cli'])"
