#!/bin/bash
# round-3 GPU session 24: accumulators home through the copy stream: CLI scoring phase, bench cold / warm, score tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3za; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prepare or single_read or automatic_scoring or score" > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest.log"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest
out = sys.argv[1]
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
open(d + "/r.nwk", "wb").write(native_ingest.synth_trees(512, 1, 2000))
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(512, 10000, 2001))
log = []
for extra, env in (([], {}), ([], {}), ([], {})):
    o = d + "/o.nwk"
    if os.path.exists(o): os.remove(o)
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"] + extra, capture_output=True, text=True)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    el = re.findall(r"Elapsed time: (\d+) microseconds", p.stdout)
    log.append(f"== QuartetScores -t 8 --trace, 512 taxa x 10000 trees: counting {took[0] / 1e3:.1f} ms, scoring {took[1] / 1e3:.1f} ms, elapsed {int(el[0]) / 1e3:.1f} ms\n" + p.stderr)
open(out + "/cli_trace.txt", "w").write("\n".join(log))
print("\n".join(log)[:5000])
PY
timeout -k 10 600 python3 bench.py --no-cpu-baseline --steps 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1]); c=d['config']
print(d['ms_per_step'], c.get('score_phase_ms'), c.get('score_phase_ms_cold'), c.get('score_phases_ms_cold'), d.get('e2e'))"
