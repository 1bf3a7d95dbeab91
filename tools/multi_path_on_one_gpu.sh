#!/bin/bash
# The N > 1 code path of bench.py on ONE GPU (no multi-GPU lease exists): all 100 000 trees of configs[3] through the launcher parent ->
# torch.distributed.run child -> one rank with RCCL initialised and the table collective inside every step (QS_BENCH_FORCE_DIST=1), with
# the same-workload count-only step and the peer-access leg (QuartetScores --gpus 1 --reduce p2p) in the same line.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_multi; mkdir -p $O
QS_BENCH_FORCE_DIST=1 python3 bench.py --config 3 --via-launcher --p2p-leg 1 --no-cpu-baseline > $O/bench_cfg3_forced_dist_via_launcher.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 - "$O/bench_cfg3_forced_dist_via_launcher.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("%.4g"%d["value"], "%.2f ms"%d["ms_per_step"], d["scaling"], c["baseline_config"], c["collective"], c.get("collective_input_bytes_per_rank"))
print(c["one_rank_same_workload"]); print(c.get("p2p_leg")); print(d.get("collective")); print(c.get("launcher"))
PY
