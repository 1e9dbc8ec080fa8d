#!/bin/bash
# 64-byte aligned ring loads (window start on a multiple of 16 pixels) with 224-column strips vs the default
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp14.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "$* : " >> $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --global-batch 256 2>>gpurun_out/r2_exp14.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'], d['config']['kernel_plan'])" >> $L 2>&1; }
for rep in 1 2; do
run X=base
run FIXCA_LIB=$PWD/tools/bin/r2_u015/libfixca_hip.so
run FIXCA_LIB=$PWD/tools/bin/r2_u015/libfixca_hip.so FIXCA_OC=224
run FIXCA_OC=224
done
cat $L
