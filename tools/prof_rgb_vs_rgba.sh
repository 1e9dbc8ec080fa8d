#!/bin/bash
# GPU box: instruction counters of the RGBA8 and RGB8 cubic kernels, 64 images 6000x4000 per launch each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_rgbx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for which in ab ab_rgb; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/pmc_$which -o pmc -- python3 $ROOT/tools/$which.py 1 one > $OUT/pmc_$which.log 2>&1
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -E "k_roll" $OUT/summary.txt | grep -v tables
