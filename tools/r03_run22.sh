#!/bin/bash
# round-3 GPU session 22: where does the CLI's scoring phase go (cold call)? --gpus 1 trace
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3y; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest
out = sys.argv[1]
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
open(d + "/r.nwk", "wb").write(native_ingest.synth_trees(512, 1, 2000))
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(512, 10000, 2001))
log = []
for extra in ([], [], ["--gpus", "1"]):
    o = d + "/o.nwk"
    if os.path.exists(o): os.remove(o)
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8", "--trace"] + extra, capture_output=True, text=True)
    took = [int(x) for x in re.findall(r"It took: (\d+) microseconds", p.stdout)]
    log.append(f"== {' '.join(extra) or 'one GPU'}: counting {took[0] / 1e3:.1f} ms, scoring {took[1] / 1e3:.1f} ms\n" + p.stderr)
open(out + "/cli_trace.txt", "w").write("\n".join(log))
print("\n".join(log)[:6000])
PY
