#!/bin/bash
# round-3 GPU session 11: out-of-core demo (515 GB through one GPU, random trees, lookups against the brute force);
# bench --config 4 (one table shard incl. sharded scoring), --config 1, ladder; CLI --gpus 1 timing (RCCL init beside counting)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3l; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 bash tools/out_of_core_demo.sh 1200 40 > "$OUT/out_of_core.txt" 2>&1; echo "ooc rc $?" | tee "$OUT/summary.txt"
tail -12 "$OUT/out_of_core.txt"
for args in "--config 4" "--config 1" "--shape ladder" "--config 3 --trees 12500"; do
  tag=$(echo "b $args" | tr -d ' -')
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-e2e $args > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "bench $tag rc $?" | tee -a "$OUT/summary.txt"
done
python3 - "$OUT" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1]); c = d["config"]
        print(f.split("/")[-1], c["workload"][:60], "| ms", round(d["ms_per_step"], 3), "value %.3e" % d["value"], "frac", round(d["roofline"]["frac"], 3), "| score", c.get("score_mode"), c["score_phase_ms"], "| gates", c["parity_tuple_sums_ok"], c["parity_bitslice_equals_swar_impl"], c["parity_lookup_equals_bruteforce"], "|", c["algo"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-400:])
PY
timeout -k 10 300 bash tools/cli_timing.sh 512 10000 > "$OUT/cli_timing_512.txt" 2>&1; cat "$OUT/cli_timing_512.txt"
timeout -k 10 300 bash tools/cli_timing.sh 256 20000 > "$OUT/cli_timing_256.txt" 2>&1; cat "$OUT/cli_timing_256.txt"
