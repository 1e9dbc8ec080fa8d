#!/bin/bash
# GPU box: engine clock under k_wcol (GRBM_GUI_ACTIVE over the kernel's duration, same run) and VALU issue share
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_wide_clk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/run -o p -- python3 $ROOT/tools/wide_bench.py "RGBA u16" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
d = "$OUT/run"
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
pc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "k_wcol" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(pc[0])):
    if "k_wcol" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(acc, key=int)[:6]:
    c = acc[k]; us = dur.get(k, 0)
    print("dispatch %s: %.1f us  GRBM_GUI_ACTIVE %.3g (%.2f GHz if summed over 8 XCDs)  VALU %.3g  WAVE_CYCLES %.3g WAIT %.3g BUSY_CU %.3g" % (k, us, c.get("GRBM_GUI_ACTIVE", 0), c.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3 if us else 0, c.get("SQ_INSTS_VALU", 0), c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_WAIT_INST_ANY", 0), c.get("SQ_BUSY_CU_CYCLES", 0)))
PY
