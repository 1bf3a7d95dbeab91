#!/bin/bash
# round 5, lease 8: CLI trace with the launch order built beside the HIP start-up; PMC of the general workloads on the final source;
# final lines of the other configs; the GPU tests that touch qs_create / the CLI
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05_c8; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -k "cli or tiny_and_odd or shards or gather_counts_bit_exact or smoke or two_column" > $O/pytest_sel.log 2>&1 || { tail -40 $O/pytest_sel.log; exit 1; }
tail -2 $O/pytest_sel.log
bash tools/cli_trace.sh 512 10000 8 4 > $O/cli_trace_512x10000_t8.txt 2>&1
grep -E "^== run|Elapsed|launch order|count enqueued" $O/cli_trace_512x10000_t8.txt | head -24
bash tools/pmc_collect.sh r05_c8/pmc_collapse --trees 1500 --collapse 0.2
bash tools/pmc_collect.sh r05_c8/pmc_collapse_dropout --trees 1500 --collapse 0.2 --dropout 0.1
for c in 1 3 4; do python3 bench.py --config $c --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || { tail -20 $O/bench_cfg$c.err; exit 1; }; tail -c 300 $O/bench_cfg$c.json; echo; done
