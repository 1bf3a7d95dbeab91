#!/bin/bash
# Round-end evidence run on the GPU box (through gpurun from the repo root): tools/final_profiles.sh <tag>
# Writes gpurun_out/<tag>/...; copy what should be judged into profiles/.
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
last() { tail -1 "$1" > "$2"; }
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_n1.log 2>&1; last $OUT/bench_n1.log $OUT/bench_n1.json
python bench.py --no-cpu-baseline --collapse 0.2 > $OUT/bench_general.log 2>&1; last $OUT/bench_general.log $OUT/bench_general_full.json
python bench.py --no-cpu-baseline --dropout 0.1 > $OUT/bench_partial.log 2>&1; last $OUT/bench_partial.log $OUT/bench_partial.json
python bench.py --no-cpu-baseline --algo scatter --steps 2 --warmup 1 > $OUT/bench_scatter.log 2>&1; last $OUT/bench_scatter.log $OUT/bench_scatter.json
QS_GATHER_IMPL=swar python bench.py --no-cpu-baseline > $OUT/bench_swar.log 2>&1; last $OUT/bench_swar.log $OUT/bench_swar.json
QS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $OUT/bench_forced_dist.log 2>&1; last $OUT/bench_forced_dist.log $OUT/bench_forced_dist_scatter.json
python bench.py --no-cpu-baseline --taxa 256 --trees 12500 --steps 3 --warmup 1 > $OUT/bench_cfg4.log 2>&1; last $OUT/bench_cfg4.log $OUT/bench_cfg4_pergpu.json
python bench.py --no-cpu-baseline --taxa 512 --trees 10000 --steps 2 --warmup 1 --no-score > $OUT/bench_cfg3.log 2>&1; last $OUT/bench_cfg3.log $OUT/bench_cfg3.json
python bench.py --no-cpu-baseline --taxa 512 --trees 10000 --steps 2 --warmup 1 --no-score --count-bits 16 > $OUT/bench_cfg3_u16.log 2>&1; last $OUT/bench_cfg3_u16.log $OUT/bench_cfg3_u16.json
python bench.py --no-cpu-baseline --taxa 1024 --trees 5000 --count-bits 16 --table-shards 8 --shard-index 3 --steps 1 --warmup 1 --no-score > $OUT/bench_cfg5.log 2>&1; last $OUT/bench_cfg5.log $OUT/bench_cfg5_shard3.json
tools/pmc_collect.sh $TAG/pmc
ls $OUT
