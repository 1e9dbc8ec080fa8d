#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp7.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "$* : " >> $L
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --batch 16 2>>gpurun_out/r2_exp7.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'])" >> $L 2>&1; }
for v in "" r2_nofilter r2_novm r2_aligned; do
  lib=$PWD/gimp-fix-ca_amd/libfixca_hip.so; [ -n "$v" ] && lib=$PWD/tools/bin/$v/libfixca_hip.so
  run FIXCA_LIB=$lib FIXCA_DEBUG_RESIDENT=0
  run FIXCA_LIB=$lib X=two-level
done
cat $L
