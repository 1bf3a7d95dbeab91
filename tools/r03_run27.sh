#!/bin/bash
# round-3 GPU session 27: next-chunk prefetch in the bundle score kernel (QS_TUNE_SCORE_LOAD = 2)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3ze; mkdir -p "$OUT"; export TMPDIR=/tmp
QS_PY_TUNING="14=2" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score or single_read or automatic_scoring or sharded or rooted or multifurcating or views or config" > "$OUT/pytest_pf.log" 2>&1; echo "pytest prefetch rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_pf.log"
QS_PY_TUNING="14=2" timeout -k 10 300 python3 tools/score_soak.py 40 23 > "$OUT/score_soak_pf.txt" 2>&1; echo "soak prefetch rc $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/score_soak_pf.txt"
for t in "" "14=2"; do
  echo "== QS_PY_TUNING=$t" | tee -a "$OUT/score_load_modes.txt"
  QS_PY_TUNING="$t" timeout -k 10 300 python3 tools/score_single_read.py 512:10000 512:10000:1 256:12500 2>&1 | grep -v "chunk\|round 1/\|no pre-pass\|amdgpu.ids\|no tie" | cut -c1-200 | tee -a "$OUT/score_load_modes.txt"
done
