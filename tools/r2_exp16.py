"""GPU box: why is config 4 (8192x8192 directional) steadier and faster than config 3?  Swap sizes and parameters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
P3 = dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
P4 = dict(x_blue=1.0, x_red=0.0, y_blue=-7.5, y_red=4.0, interpolation=2)
CASES = [("6000x4000 config3", 6000, 4000, P3), ("8192x4000 config3 params", 8192, 4000, P3), ("6144x4000 config3 params", 6144, 4000, P3),
         ("5760x4000 config3 params", 5760, 4000, P3), ("6000x4000 config4 params", 6000, 4000, dict(P4, lens_x=3000, lens_y=2000)),
         ("8192x8192 config4", 8192, 8192, dict(P4, lens_x=4096, lens_y=4096)), ("6000x4000 config3", 6000, 4000, P3)]
for name, W, H, kw in CASES:
    p = fixca.make_params(**kw)
    B = 64 if H <= 4000 else 32
    src = torch.empty((B, H, W, 4), dtype=torch.uint8, device="cuda")
    for i in range(B):
        fixca.noise_fill(src[i], 0xF1CA6000 + i)
    dst = torch.empty_like(src)
    srcs, dsts = [src[i] for i in range(B)], [dst[i] for i in range(B)]
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, 4, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 12
    e0.record()
    for _ in range(N):
        fixca.fix_ca_batch(srcs, dsts, W, H, 4, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (N * B)
    print("%-28s %8.2f us/image  %5.1f %%   [%s]" % (name, us, 8.0 * W * H / us / 1e3 / 80, fixca.plan(W, H, 4, 1, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)[1]), flush=True)
    del src, dst, srcs, dsts
    torch.cuda.empty_cache()
