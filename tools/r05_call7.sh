#!/bin/bash
# round 5, lease 7: count soak with the depth clamp forced (220 cases), rocprofv3 stats + PMC of the default line on the final source
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c7; mkdir -p $O
timeout -k 10 700 python3 tools/count_soak.py 220 5 > $O/count_soak_220.txt 2>&1; echo "soak rc $?"; tail -2 $O/count_soak_220.txt
bash tools/pmc_collect.sh r05_c7/pmc_cfg2
ls gpurun_out/r05_c7/pmc_cfg2 | head -30
