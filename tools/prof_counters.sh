#!/bin/bash
# GPU box: SQ instruction / wait counters of whatever kernel tools/ab.py's arguments select, one pass per counter set.
#   tools/prof_counters.sh NAME --bpc 2 --interp 2 --images 16 HEAD,per=8
# Summary under gpurun_out/prof_NAME/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$ROOT/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/tools/ab.py --rounds 1 --reps 1 "$@" > $OUT/pmc_$name.log 2>&1
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
grep -E "k_wcol|k_lin|k_cub|k_move|k_roll" $OUT/summary.txt | grep -v tables
