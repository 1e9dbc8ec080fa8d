#!/bin/bash
# images per launch: 8 .. 128 on one box (library built with ROLL_BATCH_MAX=128)
cd ${GRAFT_REPO_ROOT:-.}
L=gpurun_out/r2_exp13.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
for rep in 1 2; do
for P in 8 16 24 32 48 64 96 128; do
  echo -n "per-launch $P : " >> $L
  FIXCA_LIB=$PWD/tools/bin/r2_b128/libfixca_hip.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --global-batch 384 --per-launch $P 2>>gpurun_out/r2_exp13.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['config']['launches_per_step'])" >> $L 2>&1
done
done
cat $L
