"""GPU box: RGB8 vs RGBA8 cubic 6000x4000 per-image time through the batch entry, same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H = 6000, 4000
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
for C in (3, 4, 3, 4):
    B = 32
    src = torch.empty((B, H, W, C), dtype=torch.uint8, device="cuda")
    for i in range(B):
        fixca.noise_fill(src[i], 0xF1CA3000 + i)
    dst = torch.empty_like(src)
    srcs, dsts = [src[i] for i in range(B)], [dst[i] for i in range(B)]
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 30
    e0.record()
    for _ in range(N):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (N * B)
    print("C=%d  %.2f us/image  %.1f %% of 8 TB/s   [%s]" % (C, us, 2.0 * C * W * H / us / 1e3 / 80, fixca.plan(W, H, C, 1, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)[1]), flush=True)
    del src, dst, srcs, dsts
