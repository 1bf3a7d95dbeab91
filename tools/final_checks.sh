set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 200 python3 tools/count_soak.py 120 6 > $O/count_soak_120_seed6.txt 2>&1; echo "count soak rc $?"; tail -1 $O/count_soak_120_seed6.txt
timeout -k 10 200 python3 tools/score_soak.py 60 5 > $O/score_soak_60.txt 2>&1; echo "score soak rc $?"; tail -1 $O/score_soak_60.txt
