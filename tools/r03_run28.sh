#!/bin/bash
# round-3 GPU session 28: cooperative loads + next chunk requested ahead (QS_TUNE_SCORE_LOAD = 3)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3zf; mkdir -p "$OUT"; export TMPDIR=/tmp
QS_PY_TUNING="14=3" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score or single_read or automatic_scoring or sharded or views" > "$OUT/pytest_cpf.log" 2>&1; echo "pytest coop+prefetch rc $?" | tee "$OUT/summary.txt"
tail -3 "$OUT/pytest_cpf.log"
for t in "" "14=3"; do
  echo "== QS_PY_TUNING=$t" | tee -a "$OUT/score_load_modes.txt"
  QS_PY_TUNING="$t" timeout -k 10 300 python3 tools/score_single_read.py 512:10000 256:12500 2>&1 | grep -v "chunk\|round 1/\|no pre-pass\|amdgpu.ids\|no tie" | cut -c1-200 | tee -a "$OUT/score_load_modes.txt"
done
