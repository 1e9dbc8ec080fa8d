#!/bin/bash
# A/B on one box: round-1 kernel vs two-level schedule
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp6.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt|exact count      st_aux=2 :|dynamic th= 32" > $L
run() { echo -n "$* : " >> $L
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --batch ${B:-16} 2>>gpurun_out/r2_exp6.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'])" >> $L 2>&1; }
for rep in 1 2; do
B=16 run FIXCA_LIB=$PWD/tools/bin/r2_head/libfixca_hip.so
B=16 run FIXCA_DEBUG_RESIDENT=0
B=16 run X=1
B=32 run X=1
B=32 run FIXCA_DEBUG_RESIDENT=0
B=16 run FIXCA_DEBUG_RESIDENT=2000
B=16 run FIXCA_DEBUG_RESIDENT=2560
done
cat $L; tail -5 gpurun_out/r2_exp6.err
