#!/bin/bash
# round 5, lease 6: the whole GPU suite on the final kernels, the CLI's wall with its trace stamps, the default bench line in full
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c6; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -k "configs2 or configs3 or shard_of_configs4 or savemem or cli" > $O/pytest_gpu_rerun.log 2>&1 || { tail -40 $O/pytest_gpu_rerun.log; exit 1; }
tail -3 $O/pytest_gpu_rerun.log
bash tools/cli_trace.sh 512 10000 8 4 > $O/cli_trace_512x10000_t8.txt 2>&1
grep -E "^== run|Elapsed" $O/cli_trace_512x10000_t8.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
tail -c 1500 $O/bench_default.json; echo
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
