"""GPU box: the two allocation modes of the metric's kernel.  Allocates NP image pools (128 images 6000x4000 RGBA8 each,
source + result) one after the other, keeps them all, and times the 128-image launch on each (HIP events).  Under
    rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY -- python3 tools/alloc_modes.py
the k_roll dispatches come in the same order (REPS + 1 per pool), so tools/alloc_modes.py --join <counter csv> puts the
counters next to the times.  Pools of other sizes: ALLOC_IMAGES (default 128); ALLOC_POOLS (default 6)."""
import os, sys, csv, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 4
if len(sys.argv) > 2 and sys.argv[1] == "--join":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_roll<" in r["Kernel_Name"] and "tables" not in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    n = min(len(v) for v in by.values()) // (REPS + 1)
    print("pool " + " ".join("%28s" % k for k in sorted(by)))
    for p in range(n):
        print("%4d " % p + " ".join("%28.4g" % statistics.mean(by[k][p * (REPS + 1) + 1:(p + 1) * (REPS + 1)]) for k in sorted(by)))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C = 6000, 4000, 4
N = int(os.environ.get("ALLOC_IMAGES", "128"))
NP = int(os.environ.get("ALLOC_POOLS", "6"))
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
pools = []
for k in range(NP):
    src = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for i in range(N):
        fixca.noise_fill(src[i], 0xF1CA1000 + i)
    pools.append((src, dst))
    srcs, dsts = [src[i] for i in range(N)], [dst[i] for i in range(N)]
    fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    print("pool %d: src %#x dst %#x  %.1f us per launch  %.1f %% of 8 TB/s" % (k, src.data_ptr(), dst.data_ptr(), us, 2.0 * N * W * H * C / us / 1e3 / 80.0), flush=True)
