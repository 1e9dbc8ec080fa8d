#!/bin/bash
# round-2 experiment 1: structure skeletons + real-kernel th / nt-load sweep
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 300 tools/bin/kbench2 12 > gpurun_out/kbench2.log 2>&1 || echo "kbench2 failed" >> gpurun_out/kbench2.log
for lib in gimp-fix-ca_amd/libfixca_hip.so tools/bin/r2_ldnt/libfixca_hip.so; do
  for th in 33 128 800; do
    echo "== lib=$lib th=$th" >> gpurun_out/r2_thsweep.log
    FIXCA_LIB=$PWD/$lib FIXCA_TH=$th timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel_plan'], d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'])" >> gpurun_out/r2_thsweep.log 2>&1
  done
done
cat gpurun_out/r2_thsweep.log
