#!/bin/bash
# round-3 GPU session 20: log reuse through qs_score_pass1 / qs_score_pass2; full GPU suite; bench --config 4 scoring
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3u; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest gpu rc $?" | tee "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-e2e > "$OUT/bench_cfg4_shard.json" 2> "$OUT/bench_cfg4_shard.err"; echo "bench cfg4 rc $?"
python3 -c "
import json
d=json.loads(open('$OUT/bench_cfg4_shard.json').read().strip().split('\n')[-1]); c=d['config']
print(d['ms_per_step'], c.get('score_mode'), c.get('score_phase_ms'), c.get('score_phase_ms_cold'))"
