#!/bin/bash
# does the number depend on how long the GPU has been busy? (warm-up / steps of the bench's timed loop)
cd ${GRAFT_REPO_ROOT:-.}
L=gpurun_out/r2_exp17.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
for rep in 1 2; do
for ws in "0 4" "1 4" "2 10" "5 10" "10 20" "20 40" "2 10"; do set -- $ws
  echo -n "warmup $1 steps $2 : " >> $L
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --warmup $1 --steps $2 --global-batch 256 2>>gpurun_out/r2_exp17.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'])" >> $L 2>&1
done
done
cat $L
