#!/bin/bash
# plain pass 1 / pass 2 and qs_score of library builds with other chunk sizes / wave counts of the bundle kernel: tools/score_chunk_ab.sh <exp names...>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_score_chunks; mkdir -p $O
echo "== product"; timeout -k 10 300 python3 tools/score_phases.py 512:10000 256:12500 2>&1 | grep "kernel=bundle" | tee $O/product.txt
for e in "$@"; do echo "== $e"; QS_LIB=tools/bin/libqs_exp$e.so timeout -k 10 300 python3 tools/score_phases.py 512:10000 256:12500 2>&1 | grep "kernel=bundle" | tee $O/$e.txt; done
