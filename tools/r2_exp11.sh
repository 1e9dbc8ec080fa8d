#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp11.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "B=$B $* : " >> $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --batch $B 2>>gpurun_out/r2_exp11.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'])" >> $L 2>&1; }
for rep in 1 2 3; do
for B in 32 16; do run FIXCA_PERSIST=0; run FIXCA_PERSIST=1; done
done
cat $L; tail -3 gpurun_out/r2_exp11.err
FIXCA_PERSIST=1 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "batch or full_size" 2>&1 | tail -3
