#!/bin/bash
# isa_report.sh <outdir> [hipcc -D flags...]: compile qs_count.hip for gfx950 with the given flags, keep the .s, and print
# the register / spill lines of every count_bitslice3 instance (kernel work: what did the flag do to codegen?)
set -e
OUT=${1:?outdir}; shift
mkdir -p "$OUT"
cd "$(dirname "$0")/../quartetscores_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -save-temps=obj -c qs_count.hip -o "$OUT/qs_count.o" 2>/dev/null
S="$OUT/qs_count-hip-amdgcn-amd-amdhsa-gfx950.s"
awk '/^ +\.name: +_ZN2qs22count_bitslice3/{n=$2; k=1} k&&/\.private_segment_fixed_size:/{ps=$2} k&&/\.sgpr_count:/{sg=$2} k&&/\.vgpr_count:/{v=$2} k&&/\.vgpr_spill_count:/{print n, "vgpr", v, "spill", $2, "sgpr", sg, "scratch", ps; k=0}' "$S" \
  | sed 's/_ZN2qs22count_bitslice3_kernelILi\([0-9]*\)ELi\([0-9]*\)E\([jt]\)[^ ]*/B=\1 mode=\2 \3/'
