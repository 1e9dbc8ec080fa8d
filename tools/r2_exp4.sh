#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 100 tools/bin/kbench4 5 2>&1 | grep -E "only|exact count      st_aux=2 :|dynamic th" > gpurun_out/r2_exp4.log
for pad in 0 6500 10000; do
  for th in 16 20 24 28 33; do
    echo -n "real kernel pad=$pad th=$th : " >> gpurun_out/r2_exp4.log
    FIXCA_EXP_LDS_PAD=$pad FIXCA_TH=$th timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'])" >> gpurun_out/r2_exp4.log 2>&1
  done
done
cat gpurun_out/r2_exp4.log
