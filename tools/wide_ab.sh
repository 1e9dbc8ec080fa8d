#!/bin/bash
# GPU box: RGBA u16 cubic time for k_wcol build variants (tools/bin/w_*) x tile heights.  usage: tools/wide_ab.sh "q2 q3w4" "50 51 66"
for v in $1; do
  lib=tools/bin/w_$v/libfixca_hip.so
  [ "$v" = head ] && lib=gimp-fix-ca_amd/libfixca_hip.so
  for th in $2; do
    printf "%-6s th=%-3s " $v $th
    FIXCA_LIB=$lib FIXCA_DEBUG=1 FIXCA_TH=$th timeout -k 10 120 python tools/wide_bench.py "${3:-RGBA u16}" 2>&1 | grep interp | cut -c1-75
  done
done
