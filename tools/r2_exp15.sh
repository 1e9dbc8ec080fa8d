#!/bin/bash
# ring windows that start on a 64-byte boundary (272-column window, second 64-byte DMA piece, DPP kept) vs the default
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
L=gpurun_out/r2_exp15.log
timeout 100 tools/bin/kbench4 4 2>&1 | grep -E "write-only nt" > $L
run() { echo -n "$* : " >> $L
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --global-batch 256 2>>gpurun_out/r2_exp15.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['us_per_image'], d['roofline']['frac'], d['ok'], d['config']['kernel_plan'])" >> $L 2>&1; }
for rep in 1 2 3; do
run X=base
run FIXCA_LIB=$PWD/tools/bin/r2_al64/libfixca_hip.so
done
cat $L
FIXCA_LIB=$PWD/tools/bin/r2_al64/libfixca_hip.so python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fuzz_auto or variants or full_size or batch or crops" 2>&1 | tail -3
