#!/bin/bash
# GPU box: HBM traffic of the RGB8 kernel (rocprofv3 PMC, separate passes), 64 images 6000x4000 per launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_rgb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$set -o pmc -- python3 $ROOT/tools/ab_rgb.py 1 one > $OUT/pmc_$set.log 2>&1
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -E "k_roll" $OUT/summary.txt
