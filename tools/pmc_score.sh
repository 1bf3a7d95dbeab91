#!/bin/bash
# Counters of the scoring pass (score_bundle_kernel) at configs[2]: tools/pmc_score.sh <tag>   (through gpurun from the repo root)
# One --pmc pass per counter group (never combined with other trace domains); groups are built from what `rocprofv3 -L` lists on the box.
set -u
TAG=${1:-r05_score_pmc}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-impl-check --no-e2e --prewarm-ms 0 --steps 1 --warmup 0"
cd /tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1
have() { grep -qw "$1" "$OUT/avail.txt"; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE"; do
    use=""
    for c in $grp; do if have $c; then use="$use $c"; else echo "not on this box: $c" >> "$OUT/missing.txt"; fi; done
    [ -z "$use" ] && continue
    name=$(echo $use | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $use -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $use" >> "$OUT/errors.txt"
    echo "done: $use"
done
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/score_counters.txt"
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> value
names = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "score_bundle_kernel" not in k: continue
        agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = k[:90]
for c in sorted(agg):
    vals = sorted(agg[c].values())
    print(f"{c:45s} dispatches {len(vals):3d}  max {vals[-1]:.6g}  (full passes = the largest) top4 {[float('%.5g' % v) for v in vals[-4:]]}")
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score" in r.get("Name", ""): print("stats:", r["Name"][:80], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MaxNs"))
PY
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete
cat "$OUT/score_counters.txt"
