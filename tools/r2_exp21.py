"""GPU box: do the two speed modes of an allocation also show in a plain nt copy of the same buffers?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C, N = 6000, 4000, 4, 64
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
def k_roll(srcs, dsts, reps=5):
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    return 8.0 * W * H / (e0.elapsed_time(e1) * 1e3 / (reps * N)) / 1e3 / 80
hold = []
for k in range(10):
    src = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    srcs, dsts = [src[i] for i in range(N)], [dst[i] for i in range(N)]
    for i in range(N):
        fixca.noise_fill(src[i], 0xF1CAB000 + i)
    a = k_roll(srcs, dsts)
    fixca.time_copy(srcs, dsts, N, W * H * C)
    cp = 2.0 * W * H * C / (fixca.time_copy(srcs, dsts, 4 * N, W * H * C) * 1e-3) / 1e9 / 80
    b = k_roll(srcs, dsts)
    # the same copy with source and destination swapped (reads what was the write side)
    cp2 = 2.0 * W * H * C / (fixca.time_copy(dsts, srcs, 4 * N, W * H * C) * 1e-3) / 1e9 / 80
    print("allocation %d: k_roll %5.1f / %5.1f %%   nt copy src->dst %5.1f %%   dst->src %5.1f %%" % (k, a, b, cp, cp2), flush=True)
    if k in (1, 4):
        hold.append((src, dst))
    del src, dst, srcs, dsts
    torch.cuda.empty_cache()
