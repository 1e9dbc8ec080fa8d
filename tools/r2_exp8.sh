#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp8.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "B=$B $* : " >> $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --batch $B 2>>gpurun_out/r2_exp8.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['config']['kernel_plan'])" >> $L 2>&1; }
for rep in 1 2; do
for B in 16 32 64; do run FIXCA_DEBUG_RESIDENT=0; done
B=64 run FIXCA_DEBUG_RESIDENT=0 FIXCA_TH=29
B=64 run FIXCA_DEBUG_RESIDENT=0 FIXCA_TH=41
B=64 run FIXCA_DEBUG_RESIDENT=0 FIXCA_TH=25
done
cat $L; tail -3 gpurun_out/r2_exp8.err
