#!/bin/bash
# round-3 GPU session 15: single-read scoring with a sampled pre-pass; --table-shards with one table allocation per GPU
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3p; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_read or shards" > "$OUT/pytest_single.log" 2>&1; echo "pytest single-read rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_single.log"
timeout -k 10 400 python3 tools/score_single_read.py > "$OUT/score_single_read.txt" 2>&1; cat "$OUT/score_single_read.txt"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re, time
sys.path.insert(0, ".")
import numpy as np
from quartetscores_amd import native_ingest, synth
out = sys.argv[1]
d = "/tmp/qs_spill"; os.makedirs(d, exist_ok=True)
lines = []
n, m = 1024, 500
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(n, m, 31))
open(d + "/r.nwk", "w").write(synth.random_tree(n, np.random.default_rng(30)) + "\n")
outs = []
for extra in (["--table-shards", "8", "--trace"], ["--table-shards", "0"]):
    o = d + "/o%d.nwk" % len(outs)
    t0 = time.time()
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8"] + extra, capture_output=True, text=True)
    took = re.findall(r"It took: (\d+) microseconds", p.stdout)
    lines.append(f"{n} taxa x {m} trees {' '.join(extra)}: rc {p.returncode}, wall {time.time() - t0:.1f} s, counting {int(took[0]) / 1e3:.1f} ms, scoring {int(took[1]) / 1e3:.1f} ms")
    if "--trace" in extra: lines += ["    " + l for l in p.stderr.split("\n") if "shard" in l][:40]
    outs.append(open(o).read() if os.path.exists(o) else None)
lines.append("outputs identical: %s" % (outs[0] is not None and all(x == outs[0] for x in outs)))
open(out + "/spill_policy.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
