#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/pmc_collect.sh <tag> [bench.py args...]
# Writes under gpurun_out/<tag>/: kernel-trace stats, one --pmc pass per counter group (never combined with
# other trace domains), and the JSON summary tools/pmc_summary.py makes of them.
set -u
TAG=${1:?tag}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-score --steps 200 --warmup 20 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
done
cd "$ROOT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete
du -sh "$OUT" | tee "$OUT/size.txt"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
