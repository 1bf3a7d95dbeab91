#!/bin/bash
# round-3 GPU session 6: deep-tree instances (8..10 depth bits): parity tests, ladder workloads against random ones;
# CLI trace; full GPU test-suite
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3g; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep or depth_width or cooperative or ladders or two_cell" > "$OUT/pytest_deep.log" 2>&1; echo "pytest deep rc $?" | tee "$OUT/summary.txt"
tail -8 "$OUT/pytest_deep.log"
B=tools/bin; P=quartetscores_amd/lib/libquartetscores_hip.so
timeout -k 10 300 $B/count_bench 512 10000 32 3 $P > "$OUT/cb_512_random.txt" 2>&1
CB_LADDER=1 CB_NNI=1 timeout -k 10 300 $B/count_bench 512 10000 32 3 $P > "$OUT/cb_512_ladder.txt" 2>&1
CB_LADDER=1 CB_NNI=1 CB_DLO=869 CB_DHI=896 timeout -k 10 300 $B/count_bench 1024 5000 16 3 $P > "$OUT/cb_1024_ladder_shard.txt" 2>&1
CB_DLO=869 CB_DHI=896 timeout -k 10 300 $B/count_bench 1024 5000 16 3 $P > "$OUT/cb_1024_random_shard.txt" 2>&1
CB_LADDER=1 CB_NNI=1 timeout -k 10 300 $B/count_bench 256 12500 32 3 $P > "$OUT/cb_256_ladder.txt" 2>&1
for f in "$OUT"/cb_*.txt; do echo "$(basename $f): $(grep -h 'workload' $f | sed 's/.*max LCA/max LCA/') | $(grep -h 'count ' $f | sed 's/.*gather\/\([^ ]*\) *count *\([0-9.]*\) ms.*checksum \(.*\)/\1 \2 ms \3/')"; done | tee -a "$OUT/summary.txt"
# CLI trace at configs[2]
python3 - "$OUT" <<'PY'
import subprocess, sys, os
sys.path.insert(0, ".")
from quartetscores_amd import native_ingest
out = sys.argv[1]
d = "/tmp/qs_trace"; os.makedirs(d, exist_ok=True)
n, m = 512, 10000
ref = native_ingest.synth_trees(n, 1, 2000)
open(d + "/r.nwk", "wb").write(ref)
open(d + "/e.nwk", "wb").write(native_ingest.synth_trees(n, m, 2001))
with open(out + "/cli_trace.txt", "w") as f:
    for t in ("8", "0"):
        for rep in range(2):
            o = d + f"/o{t}{rep}.nwk"
            if os.path.exists(o): os.remove(o)
            p = subprocess.run(["quartetscores_amd/bin/QuartetScores", "-r", d + "/r.nwk", "-e", d + "/e.nwk", "-o", o, "-t", t, "--trace"], capture_output=True, text=True)
            f.write(f"== -t {t} run {rep} rc {p.returncode}\n" + p.stderr + "\n".join(l for l in p.stdout.split("\n") if "took" in l or "Elapsed" in l) + "\n")
print(open(out + "/cli_trace.txt").read())
PY
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest all rc $?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
