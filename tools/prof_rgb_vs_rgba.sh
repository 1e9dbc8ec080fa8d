#!/bin/bash
# GPU box: instruction counters of the RGBA8 and RGB8 cubic kernels, 64 images 6000x4000 per launch each
# (one pass of tools/ab.py per pixel format under rocprofv3 --pmc; summary under gpurun_out/prof_rgbx/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_rgbx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for bpp in 4 3; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-24)
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_bpp${bpp}_$name -o pmc -- python3 $ROOT/tools/ab.py --bpp $bpp --images 64 --rounds 1 --reps 1 HEAD,per=64 > $OUT/pmc_bpp${bpp}_$name.log 2>&1
  done
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -E "k_roll" $OUT/summary.txt | grep -v tables
