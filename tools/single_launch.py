"""GPU box: one image per launch (what fix_ca_region does for the plug-in), N launches back to back on one stream:
prints the HIP-event time per launch.  Run under `rocprofv3 --kernel-trace --stats` to get the kernels' own durations
next to it: the difference is what the launch boundary costs.
    python tools/single_launch.py [launches] [bpp] [bpc] [interp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bpc = int(sys.argv[3]) if len(sys.argv) > 3 else 1
interp = int(sys.argv[4]) if len(sys.argv) > 4 else 2
W, H = 6000, 4000
pb = C * abs(bpc)
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=interp)
src = torch.empty((8, H, W, pb), dtype=torch.uint8, device="cuda")
for i in range(8):
    fixca.noise_fill(src[i], 0xF1CA5000 + i)
dst = torch.empty_like(src)
srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
fixca.time_launches(srcs, dsts, 16, W, H, pb, bpc, p)
ms = fixca.time_launches(srcs, dsts, N, W, H, pb, bpc, p)
print("%s: %d launches, %.2f us per launch by HIP events = %.1f %% of 8 TB/s" % (fixca.plan(W, H, pb, bpc, p)[1], N, ms * 1e3, 2.0 * W * H * pb / (ms * 1e-3) / 8e10), flush=True)
