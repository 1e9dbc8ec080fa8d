"""GPU box: k_wcol on RGBA u16 6000x4000 cubic restricted to the first n tile rows (th fixed by FIXCA_TH): how the time
grows with the number of waves per SIMD (1000 waves = 1 per SIMD)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"] = "1"
th = int(os.environ.setdefault("FIXCA_TH", "66"))
import torch, fixca
W, H, nch, sb, bpc = 6000, 4000, 4, 2, 2
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
B = 4
raw = torch.empty((B, H, W, nch * sb), dtype=torch.uint8, device="cuda")
for i in range(B):
    fixca.noise_fill(raw[i], 0xF1CA4000 + i)
dst = torch.empty_like(raw)
for chunks in [int(a) for a in sys.argv[1:]] or [5, 10, 20, 30, 40, 50, 60]:
    y2 = min(H, chunks * th)
    for i in range(B):
        fixca.fix_ca_region(raw[i], dst[i], W, H, nch * sb, bpc, p, 0, W, 0, y2, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 12
    e0.record()
    for k in range(N):
        fixca.fix_ca_region(raw[k % B], dst[k % B], W, H, nch * sb, bpc, p, 0, W, 0, y2, True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    print("chunks %3d  waves %5d (%.2f per SIMD)  %8.1f us   %.3f us per tile row" % (chunks, chunks * 50, chunks * 50 / 1024, us, us / th), flush=True)
