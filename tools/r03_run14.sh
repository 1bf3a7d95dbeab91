#!/bin/bash
# round-3 GPU session 14: VGPR source banks in the yardstick; the automatic spill policy of --table-shards on one GPU
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3o; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 500 tools/bin/valu_yardstick 200 > "$OUT/valu_yardstick.txt" 2>&1; cut -c1-20,105-135,160-190 "$OUT/valu_yardstick.txt"
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
for extra in (["--table-shards", "8"], ["--table-shards", "8", "--spill", "recount"], ["--table-shards", "0"]):
    o = d + "/o%d.nwk" % len(outs)
    t0 = time.time()
    p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", "8"] + extra, capture_output=True, text=True)
    took = re.findall(r"It took: (\d+) microseconds", p.stdout)
    pol = [l for l in p.stdout.split("\n") if "table shard" in l]
    lines.append(f"{n} taxa x {m} trees {' '.join(extra)}: rc {p.returncode}, wall {time.time() - t0:.1f} s, counting {int(took[0]) / 1e3:.1f} ms, scoring {int(took[1]) / 1e3:.1f} ms | {pol[0][:200] if pol else ''}")
    outs.append(open(o).read() if os.path.exists(o) else None)
lines.append("outputs identical: %s" % (outs[0] is not None and all(x == outs[0] for x in outs)))
open(out + "/spill_policy.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
# rooted reference after the change of the item order in root_pair_sums_kernel
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "rooted" > "$OUT/pytest_rooted.log" 2>&1; echo "pytest rooted rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_rooted.log"
python3 - "$OUT" <<'PY'
import subprocess, sys, os, re
sys.path.insert(0, ".")
import numpy as np
from quartetscores_amd import native_ingest, synth
out = sys.argv[1]
d = "/tmp/qs_rooted"; os.makedirs(d, exist_ok=True)
lines = []
for n, m, extra in ((512, 2000, []),):
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
