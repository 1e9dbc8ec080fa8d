#!/bin/bash
# GPU box: like wide_ab.sh but with the planner's own tile height (no FIXCA_TH).  usage: tools/wide_ab2.sh "head q2" ["RGBA u16"]
for v in $1; do
  lib=tools/bin/w_$v/libfixca_hip.so
  [ "$v" = head ] && lib=gimp-fix-ca_amd/libfixca_hip.so
  printf "%-6s " $v
  FIXCA_LIB=$lib timeout -k 10 200 python tools/wide_bench.py "${2:-RGBA u16}" 2>&1 | grep interp | cut -c1-150
done
