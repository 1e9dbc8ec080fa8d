#!/bin/bash
# burst tiles on the real kernel: ring holds the whole tile (th + 6 rows), all rows requested at wave start
cd ${GRAFT_REPO_ROOT:-.}
timeout 100 tools/bin/kbench4 5 2>&1 | grep -E "only|exact count      st_aux=2 :|dynamic th= (12|32)" > gpurun_out/r2_exp5.log
run() { echo -n "$1 ring=$2 th=$3 : " >> gpurun_out/r2_exp5.log
  FIXCA_LIB=$PWD/$1 FIXCA_RING=$2 FIXCA_TH=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'], d['config']['kernel_plan'])" >> gpurun_out/r2_exp5.log 2>&1; }
run gimp-fix-ca_amd/libfixca_hip.so 16 33
for rt in "18 12" "22 16" "26 20" "31 25" "31 33" "39 33" "39 45" "52 46" "16 10"; do set -- $rt; run tools/bin/r2_rd64/libfixca_hip.so $1 $2; done
run gimp-fix-ca_amd/libfixca_hip.so 16 33
cat gpurun_out/r2_exp5.log
