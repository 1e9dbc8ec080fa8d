#!/bin/bash
# real kernel, batch of 16: waves per CU (LDS pad) x rows per tile, plus box classification probes
cd ${GRAFT_REPO_ROOT:-.}
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "only|exact count      st_aux=2 :|dynamic th= 32" > gpurun_out/r2_exp3.log
for rep in 1 2; do
for pad in 0 2000 6500 10000 16000 24000; do
  for th in 33 57; do
    echo -n "pad=$pad th=$th : " >> gpurun_out/r2_exp3.log
    FIXCA_EXP_LDS_PAD=$pad FIXCA_TH=$th timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'])" >> gpurun_out/r2_exp3.log 2>&1
  done
done
done
cat gpurun_out/r2_exp3.log
