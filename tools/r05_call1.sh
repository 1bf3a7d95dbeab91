#!/bin/bash
# round 5, lease 1: baselines of this box + PMC of the general / partial count instances
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r05_c1
python3 bench.py --no-cpu-baseline --no-e2e > gpurun_out/r05_c1/bench_default.json 2> gpurun_out/r05_c1/bench_default.err
tail -c 400 gpurun_out/r05_c1/bench_default.json; echo
python3 bench.py --no-cpu-baseline --no-e2e --no-score --trees 1500 --collapse 0.2 > gpurun_out/r05_c1/bench_collapse.json 2> gpurun_out/r05_c1/bench_collapse.err
tail -c 300 gpurun_out/r05_c1/bench_collapse.json; echo
python3 bench.py --no-cpu-baseline --no-e2e --no-score --trees 1500 --collapse 0.2 --dropout 0.1 > gpurun_out/r05_c1/bench_collapse_dropout.json 2> gpurun_out/r05_c1/bench_collapse_dropout.err
tail -c 300 gpurun_out/r05_c1/bench_collapse_dropout.json; echo
bash tools/pmc_collect.sh r05_c1/pmc_collapse --trees 1500 --collapse 0.2
bash tools/pmc_collect.sh r05_c1/pmc_collapse_dropout --trees 1500 --collapse 0.2 --dropout 0.1
ls gpurun_out/r05_c1
