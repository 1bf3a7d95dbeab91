#!/bin/bash
# A/B builds of the score kernels (timing probes): tools/build_score_probe_lib.sh <name> "<flags>" -> quartetscores_amd/lib/libqs_probe_<name>.so (QS_PY_LIB=...)
set -e
name=${1:?name}; flags=${2:-}
cd "$(dirname "$0")/../quartetscores_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter $flags -c qs_score.hip -o $tmp/qs_score.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libqs_probe_$name.so qs_count.o qs_count_fused.o $tmp/qs_score.o qs_abi.o
rm -rf $tmp
ls -la ../lib/libqs_probe_$name.so
