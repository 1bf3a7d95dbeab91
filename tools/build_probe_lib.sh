#!/bin/bash
# A/B builds of the count kernels with other compile-time switches, next to the product library (never loaded unless the Python
# harness is told to: QS_PY_LIB=<path>; the library itself reads no environment):
#   tools/build_probe_lib.sh <name> "<extra hipcc flags>"   ->  quartetscores_amd/lib/libqs_probe_<name>.so
set -e
name=${1:?name}; flags=${2:-}
cd "$(dirname "$0")/../quartetscores_amd/csrc"
tmp=$(mktemp -d)
for f in qs_count qs_count_fused; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter $flags -c $f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libqs_probe_$name.so $tmp/qs_count.o $tmp/qs_count_fused.o qs_score.o qs_abi.o
rm -rf $tmp
ls -la ../lib/libqs_probe_$name.so
