#!/bin/bash
# the counters of tools/pmc_score.sh for the other load modes of the bundle kernel (QS_TUNE_SCORE_LOAD = 14): a reduced set
set -u
ROOT=$(pwd); export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-impl-check --no-e2e --prewarm-ms 0 --steps 1 --warmup 0"
for m in ${@:-1 2 3}; do
  export QS_PY_TUNING="14=$m"
  OUT=$ROOT/gpurun_out/r05_score_pmc_load$m; mkdir -p $OUT
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
  for grp in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
  done
  cd "$ROOT"
  python3 - "$OUT" <<'PY' > "$OUT/score_counters.txt"
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_bundle_kernel" not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c in sorted(agg):
    vals = sorted(agg[c].values())
    print(f"{c:45s} dispatches {len(vals):3d}  max {vals[-1]:.6g}")
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score" in r.get("Name", ""): print("stats:", r["Name"][:80], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MaxNs"))
PY
  find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
  find "$OUT" -name "*counter_collection.csv" -size +2M -delete
  find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete
  echo "== load mode $m"; cat "$OUT/score_counters.txt"
done
