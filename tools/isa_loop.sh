#!/bin/bash
# isa_loop.sh <file.s> <kernel-symbol-regex>: print the memory / wait / barrier / branch skeleton of one kernel with the
# number of VALU instructions between the lines (kernel work: where do the loads and waits sit in the hot loop?)
S=${1:?file.s}; K=${2:?kernel regex}
awk -v K="$K" '$0 ~ "^"K".*:" {p=1} p&&/s_endpgm/{exit} p{
  if ($0 ~ /buffer_load|buffer_store|global_load|global_store|ds_read|ds_write|s_waitcnt|s_barrier|s_cbranch|s_branch|^\.LBB|s_endpgm|s_sleep|s_setprio/) { if (v) printf("        ... %d valu\n", v); v=0; print NR": "$0 }
  else if ($1 ~ /^v_/) v++ }' "$S" | cut -c1-110
