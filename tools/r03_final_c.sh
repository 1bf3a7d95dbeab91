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
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re, time
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest
out = sys.argv[1]
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
open(d + "/r.nwk", "wb").write(native_ingest.synth_trees(512, 1, 2000))
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(512, 10000, 2001))
best = None
for _ in range(3):
    o = d + "/o.nwk"
    if os.path.exists(o): os.remove(o)
    time.sleep(3.0)
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"], capture_output=True, text=True)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    if best is None or took[0] < best[0]: best = (took[0], took[1], p.stderr)
open(out + "/cli_trace.txt", "w").write(f"QuartetScores -t 8 --trace, 512 taxa x 10000 trees (best of 3 by counting phase): counting {best[0] / 1e3:.1f} ms, scoring {best[1] / 1e3:.1f} ms\n" + best[2])
print(open(out + "/cli_trace.txt").read()[:1800])
PY
