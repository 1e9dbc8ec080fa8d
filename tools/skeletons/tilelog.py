#!/usr/bin/env python3
"""Summarise gpurun_out/tilelog_th*.bin (kbench3 KB_LOG=1): per-tile start/end stamps and hardware ids."""
import sys, numpy as np
for fn in sys.argv[1:]:
    a = np.fromfile(fn, dtype=np.uint64).reshape(-1, 4)
    t0, t1, t2 = (a[:, i].astype(np.int64) for i in range(3))
    hw = (a[:, 3] & 0xffffffff).astype(np.int64); xcc = (a[:, 3] >> 32).astype(np.int64) & 0xf
    base = t0.min()
    dur = (t2 - t0) / 100.0          # us (100 MHz)
    drain = (t2 - t1) / 100.0
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
    print(fn, "tiles", len(a), "span us", (t2.max() - base) / 100.0)
    print("  tile duration us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f ; store drain mean %.2f p90 %.2f" % (
        dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max(), drain.mean(), np.percentile(drain, 90)))
    # finishing time of the last tile per XCD, and tiles per XCD
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  xcc %d: tiles %6d  mean dur %.2f  last end %.1f us  first start %.1f" % (x, m.sum(), dur[m].mean(), (t2[m].max() - base) / 100.0, (t0[m].min() - base) / 100.0))
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    ks = np.unique(key)
    per = np.array([dur[key == k].sum() for k in ks]); cnt = np.array([(key == k).sum() for k in ks])
    print("  distinct (xcc,se,sh,cu): %d ; tiles per CU min %d max %d ; busy-sum per CU min %.1f mean %.1f max %.1f" % (len(ks), cnt.min(), cnt.max(), per.min(), per.mean(), per.max()))
    # progress: tiles finished over time, in 10 slices
    end = (t2 - base) / 100.0
    edges = np.linspace(0, end.max(), 11)
    print("  finished per tenth of the span:", np.histogram(end, edges)[0].tolist())
    print("  mean duration by start-time tenth:", [round(float(dur[(t0 - base) / 100.0 // (end.max() / 10 + 1e-9) == i].mean()), 2) if ((t0 - base) / 100.0 // (end.max() / 10 + 1e-9) == i).any() else None for i in range(10)])
