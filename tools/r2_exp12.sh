#!/bin/bash
# ring size / rows per tile at 32 images per launch, one box
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp12.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "$* : " >> $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --batch 32 2>>gpurun_out/r2_exp12.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['config']['kernel_plan'])" >> $L 2>&1; }
for rep in 1 2; do
run X=base
run FIXCA_RING=12
run FIXCA_RING=13
run FIXCA_RING=14
run FIXCA_RING=20
run FIXCA_TH=37
run FIXCA_TH=29
run FIXCA_OC=224
done
cat $L
