#!/bin/bash
# qs_score and its plain passes under the load modes of the bundle kernel (QS_TUNE_SCORE_LOAD = key 14): tools/score_load_modes.sh [modes...]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_score_modes; mkdir -p $O
for m in ${@:-0 1 3}; do
  echo "== QS_TUNE_SCORE_LOAD = $m"
  QS_PY_TUNING="14=$m" timeout -k 10 300 python3 tools/score_phases.py 512:10000 256:12500 1024:2000:16:0:610 2>&1 | grep -v amdgpu.ids | grep "kernel=bundle" | tee -a $O/phases_load$m.txt
done
