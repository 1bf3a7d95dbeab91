#!/bin/bash
# where the waves of the scoring pass wait: tools/pmc_score_waits.sh <tag> [QS_PY_LIB=...]   (through gpurun from the repo root)
set -u
TAG=$1; ROOT=$(pwd); export TMPDIR=/tmp
[ -n "${2:-}" ] && export QS_PY_LIB=$ROOT/$2
ARGS="--no-cpu-baseline --no-impl-check --no-e2e --prewarm-ms 0 --steps 1 --warmup 0"
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1
have() { grep -qw "$1" "$OUT/avail.txt"; }
for grp in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VSKIPPED" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_EXP_GDS SQ_ACTIVE_INST_FLAT" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS" "SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
    use=""; for c in $grp; do have $c && use="$use $c"; done
    [ -z "$use" ] && continue
    name=$(echo $use | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $use -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $use" >> "$OUT/errors.txt"
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
    print(f"{c:32s} max {vals[-1]:.6g}")
PY
find "$OUT" -type f ! -name "*.log" ! -name "*.txt" -delete
cat "$OUT/score_counters.txt"; cat "$OUT/errors.txt" 2>/dev/null; true
