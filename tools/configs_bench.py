"""GPU box: per-image kernel time of every BASELINE.json GPU configuration through the batch entry
(fixca_batch_device), HIP events on the launch stream, buffers rotated so the Infinity Cache cannot
hold them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca

CONFIGS = [
    ("RGB8 cubic off-centre 6000x4000", 6000, 4000, dict(channels=3, blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)),
    ("RGB8 none centre 6000x4000", 6000, 4000, dict(channels=3, blue=1.5, red=-1.5, lens_x=3000, lens_y=2000, interpolation=0)),
    ("config2 linear centre 6000x4000", 6000, 4000, dict(blue=2.0, red=-2.0, lens_x=3000, lens_y=2000, interpolation=1)),
    ("config3 cubic off-centre 6000x4000", 6000, 4000, dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)),
    ("none centre 6000x4000", 6000, 4000, dict(blue=1.5, red=-1.5, lens_x=3000, lens_y=2000, interpolation=0)),
    ("config4 directional cubic 8192x8192", 8192, 8192, dict(lens_x=4096, lens_y=4096, x_blue=1.0, y_blue=-7.5, y_red=4.0, interpolation=2)),
]
for name, W, H, kw in CONFIGS:
    kw = dict(kw)
    C = kw.pop("channels", 4)
    p = fixca.make_params(**kw)
    kid, desc = fixca.plan(W, H, C, 1, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)
    B = 12 if W == 6000 else 4
    srcs = [torch.randint(0, 256, (H, W, C), dtype=torch.uint8, device="cuda") for _ in range(B)]
    dsts = [torch.empty_like(s) for s in srcs]
    for _ in range(3):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 10
    e0.record()
    for _ in range(N):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (N * B)
    gbs = 2.0 * C * W * H / us / 1e3
    print("%-40s %8.2f us/image  %7.1f GB/s (%.1f %% of 8 TB/s)  %9.0f Mpix/s   [%s]" % (name, us, gbs, gbs / 80, W * H / us, desc))
    del srcs, dsts
    torch.cuda.empty_cache()
