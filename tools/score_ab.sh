#!/bin/bash
# pass-1/pass-2 timing of several builds of the library (tools/Makefile exp): tools/score_ab.sh "512:10000" 0 1 2 ...
cases=$1; shift
for e in "$@"; do echo "== exp $e"; QS_LIB=tools/bin/libqs_exp$e.so timeout 200 python tools/score_phases.py $cases 2>&1 | grep -v amdgpu.ids; done
