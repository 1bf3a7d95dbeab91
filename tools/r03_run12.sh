#!/bin/bash
# round-3 GPU session 12: rooted reference tree at 512 / 1024 taxa: what does root_pair_sums_kernel cost? + rooted tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3m; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "rooted or shards" > "$OUT/pytest_rooted.log" 2>&1; echo "pytest rooted rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_rooted.log"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
sys.path.insert(0, ".")
import numpy as np
from quartetscores_amd import native_ingest, synth
out = sys.argv[1]
d = "/tmp/qs_rooted"; os.makedirs(d, exist_ok=True)
lines = []
for n, m, extra in ((512, 2000, []), (1024, 500, ["--table-shards", "8"])):
    open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(n, m, 31))
    for rooted in (False, True):
        open(d + "/r.nwk", "w").write(synth.random_tree(n, np.random.default_rng(30), rooted=rooted) + "\n")
        o = d + "/o.nwk"
        if os.path.exists(o): os.remove(o)
        p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8"] + extra, capture_output=True, text=True)
        took = re.findall(r"It took: (\d+) microseconds", p.stdout)
        lines.append(f"{n} taxa x {m} trees {' '.join(extra)} {'rooted' if rooted else 'unrooted'} reference: rc {p.returncode}, counting {int(took[0]) / 1e3:.1f} ms, scoring {int(took[1]) / 1e3:.1f} ms")
open(out + "/rooted_timing.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
