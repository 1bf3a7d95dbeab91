#!/bin/bash
# round-3 final evidence, part B (through gpurun from the repo root): counters of the score kernels in single-read mode,
# the out-of-core demo, CLI timings (incl. --trace), rooted reference, table shards on one GPU.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03_final; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default rc $?" | tee "$OUT/summary_b.txt"
bash tools/score_pmc.sh r03_score_single 512 10000 0 > "$OUT/score_pmc.log" 2>&1; echo "score pmc rc $?" | tee -a "$OUT/summary_b.txt"
timeout -k 10 400 python3 tools/score_soak.py 120 11 > "$OUT/score_soak.txt" 2>&1; echo "score soak rc $?" | tee -a "$OUT/summary_b.txt"; tail -1 "$OUT/score_soak.txt"
timeout -k 10 600 bash tools/out_of_core_demo.sh 1200 40 > "$OUT/out_of_core.txt" 2>&1; echo "ooc rc $?" | tee -a "$OUT/summary_b.txt"
tail -6 "$OUT/out_of_core.txt"
timeout -k 10 300 bash tools/cli_timing.sh 512 10000 > "$OUT/cli_timing_512.txt" 2>&1; tail -8 "$OUT/cli_timing_512.txt"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
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
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"], capture_output=True, text=True)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    if best is None or took[0] < best[0]: best = (took[0], took[1], p.stderr)
open(out + "/cli_trace.txt", "w").write(f"QuartetScores -t 8 --trace, 512 taxa x 10000 trees (best of 3 by counting phase): counting {best[0] / 1e3:.1f} ms, scoring {best[1] / 1e3:.1f} ms\n" + best[2])
print(open(out + "/cli_trace.txt").read()[:1500])
PY
