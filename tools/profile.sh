#!/bin/bash
# Profiles bench.py's timed region on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag>
# pass 1: rocprofv3 --kernel-trace --stats over the default workload (512 images, 128 per launch):
#         (4 warmup + 16 timed steps) x 4 launches = 80 launches of k_roll -> per-kernel durations
# pass 2..: rocprofv3 --pmc <counters>, one set per pass (never combined with tracing), on 128 images
#         (1 launch per step, the same 128 images per launch: the counters are per launch)
# Everything lands under gpurun_out/prof_<tag>/ ; copy the summaries worth keeping to profiles/.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras"
export FIXCA_PROF_IMAGES=128      # images per k_roll launch in that command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --global-batch 128"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1 || echo "pmc pass failed: $set" >> $OUT/errors.log
done
python3 $ROOT/tools/summarize_prof.py $OUT "k_roll<4" $OUT/hbm_traffic.json > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
