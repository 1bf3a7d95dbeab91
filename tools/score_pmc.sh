#!/bin/bash
# rocprofv3 evidence for the score kernels (run through gpurun from the repo root):
#   tools/score_pmc.sh <tag> <taxa> <trees> <kernel 0|1>
# gpurun_out/<tag>/: kernel-trace stats and one --pmc pass per counter group (never combined with other trace domains).
set -u
TAG=${1:?tag}; N=$2; M=$3; K=$4
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/tools/score_prof.py" $N $M $K > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/tools/score_prof.py" $N $M $K 2 > "$OUT/pmc_$name.log" 2>&1 || echo "pmc group failed: $grp" >> "$OUT/errors.txt"
done
cd "$ROOT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find "$OUT" -name "*kernel_trace.csv" -path "*pmc_*" -delete
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
du -sh "$OUT" | tee "$OUT/size.txt"
