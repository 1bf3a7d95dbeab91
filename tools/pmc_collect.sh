#!/bin/bash
# Collect rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/pmc_collect.sh <tag> [bench.py args...]      e.g.  tools/pmc_collect.sh r02_cfg2 --config 2
# Writes under gpurun_out/<tag>/: kernel-trace stats, one --pmc pass per counter group (never combined with other
# trace domains; separate FETCH_SIZE / WRITE_SIZE passes as MI355X_MICROARCH.md prescribes), and summary.json =
# what tools/pmc_summary.py makes of them (copy it to profiles/<round>_<tag>_pmc_summary.json: bench.py reads the
# `entries` of profiles/r*_pmc_summary.json whose workload, kernel variant and kernel source match its own run).
set -u
TAG=${1:?tag}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-score --no-impl-check --no-e2e --secondary 0 --prewarm-ms 0 ${PMC_STEPS:---steps 4 --warmup 1} $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
done
cd "$ROOT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
# the per-dispatch counter CSVs of a 34 GB-table run are large: keep the stats CSVs and the summary
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
du -sh "$OUT" | tee "$OUT/size.txt"
