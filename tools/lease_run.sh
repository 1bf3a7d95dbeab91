#!/bin/bash
# One parametrised lease script (replaces the per-call r03_runNN.sh files): runs on the GPU box from the repo root.
#   tools/lease_run.sh <tag> <step> [<step> ...]     steps: tests | tests:<pytest -k expr> | bench[:args] | prof[:args] | pmc[:args] | sh:<command>
# Output goes to gpurun_out/<tag>/ (merged back by gpurun). Steps run in order and stop at the first failure.
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for step in "$@"; do
    i=$((i + 1))
    kind=${step%%:*}
    arg=""
    if [[ "$step" == *:* ]]; then arg=${step#*:}; fi
    echo "== step $i: $step" | tee -a "$out/steps.log"
    case "$kind" in
        tests)
            if [ -n "$arg" ]; then
                timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "$arg" > "$out/pytest_$i.log" 2>&1 || { tail -30 "$out/pytest_$i.log"; exit 1; }
            else
                timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$out/pytest_$i.log" 2>&1 || { tail -30 "$out/pytest_$i.log"; exit 1; }
            fi
            tail -3 "$out/pytest_$i.log" ;;
        bench)
            # shellcheck disable=SC2086
            timeout -k 10 900 python3 bench.py $arg > "$out/bench_$i.json" 2> "$out/bench_$i.err" || { tail -20 "$out/bench_$i.err"; exit 1; }
            tail -c 600 "$out/bench_$i.json"; echo ;;
        prof)
            # shellcheck disable=SC2086
            (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/prof_$i" -o p -- python3 "$OLDPWD/bench.py" $arg > "$OLDPWD/$out/prof_$i.log" 2>&1) || { tail -20 "$out/prof_$i.log"; exit 1; }
            find "$out/prof_$i" -name "*kernel_stats.csv" -exec head -12 {} \; ;;
        pmc)
            # arg = "<counters separated by commas>|<bench args>"
            ctrs=${arg%%|*}; bargs=${arg#*|}
            # shellcheck disable=SC2086
            (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --pmc ${ctrs//,/ } -d "$OLDPWD/$out/pmc_$i" -o p -- python3 "$OLDPWD/bench.py" $bargs > "$OLDPWD/$out/pmc_$i.log" 2>&1) || { tail -20 "$out/pmc_$i.log"; exit 1; }
            ls "$out/pmc_$i" ;;
        sh)
            timeout -k 10 1100 bash -c "$arg" > "$out/sh_$i.log" 2>&1 || { tail -30 "$out/sh_$i.log"; exit 1; }
            tail -15 "$out/sh_$i.log" ;;
        *) echo "unknown step $step"; exit 2 ;;
    esac
done
