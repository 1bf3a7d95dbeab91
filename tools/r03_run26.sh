#!/bin/bash
# round-3 GPU session 26: cooperative row loads of the bundle score kernel (QS_TUNE_SCORE_LOAD = 1): tests, soak, timings
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zd; mkdir -p "$OUT"; export TMPDIR=/tmp
QS_PY_TUNING="14=1" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score or single_read or automatic_scoring or sharded or rooted or multifurcating or views or config" > "$OUT/pytest_coop.log" 2>&1; echo "pytest coop rc $?" | tee "$OUT/summary.txt"
tail -4 "$OUT/pytest_coop.log"
QS_PY_TUNING="14=1" timeout -k 10 300 python3 tools/score_soak.py 40 21 > "$OUT/score_soak_coop.txt" 2>&1; echo "soak coop rc $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/score_soak_coop.txt"
for t in "" "14=1"; do
  echo "== QS_PY_TUNING=$t"
  QS_PY_TUNING="$t" timeout -k 10 300 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 2>&1 | grep -v "chunk\|round 1/\|no pre-pass\|amdgpu.ids" | cut -c1-200 | tee -a "$OUT/score_load_modes.txt"
done
