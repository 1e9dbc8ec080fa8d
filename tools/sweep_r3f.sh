mkdir -p gpurun_out/r3f
python -m pytest tests -m gpu -x -q -k "move or wide_samples or fuzz_wide" > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest.log; tail -3 gpurun_out/r3f/pytest.log
L=gpurun_out/r3f/sweep.log
for cfg in "2 4" "4 4" "-8 4" "2 3" "-4 3" "-8 3"; do set -- $cfg
  python tools/ab.py --single --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD >> $L 2>&1
  python tools/ab.py --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD,per=6 >> $L 2>&1
done
grep -v amdgpu.ids $L | cut -c1-230
FIXCA_LIB=tools/bin/stamps/libfixca_hip.so python tools/stamps.py 2 > gpurun_out/r3f/stamps.log 2>&1; cat gpurun_out/r3f/stamps.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f/prof -o single -- python3 $GRAFT_REPO_ROOT/tools/single_launch.py 200 > $GRAFT_REPO_ROOT/gpurun_out/r3f/single.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3f/single.log | tail -3
python3 - <<'PY'
import csv, glob, os, statistics
root = os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(root + "/gpurun_out/r3f/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_roll" in r["Kernel_Name"] and "tables" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
    print(f.split("/")[-1], "launches", len(d), "kernel us: median %.2f mean %.2f min %.2f" % (statistics.median(d), statistics.mean(d), min(d)))
    print("  gap between consecutive kernels us: median %.2f mean %.2f p90 %.2f" % (statistics.median(gaps), statistics.mean(gaps), sorted(gaps)[int(len(gaps) * 0.9)]))
PY
