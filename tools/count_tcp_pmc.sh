#!/bin/bash
# L1/L2 request counters of the count kernel (run through gpurun from the repo root): tools/count_tcp_pmc.sh <tag> [bench.py args]
set -u
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-score --no-impl-check --no-e2e --prewarm-ms 0 --steps 3 --warmup 1 $*"
cd /tmp
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
done
cd "$ROOT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
