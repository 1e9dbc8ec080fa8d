"""GPU box: is the two-mode behaviour (68 % / 71.6 % for the same binary on the same box) decided by the allocation?
Re-allocate the images several times inside one process and measure each time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C, N = 6000, 4000, 4, 64
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
def measure(src, dst, reps=6):
    srcs, dsts = [src[i] for i in range(N)], [dst[i] for i in range(N)]
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * N)
keep = []
for k in range(8):
    src = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for i in range(N):
        fixca.noise_fill(src[i], 0xF1CA8000 + i)
    a = measure(src, dst)
    b = measure(src, dst)
    print("allocation %d: src %x dst %x : %6.2f / %6.2f us/image = %5.1f / %5.1f %%" % (k, src.data_ptr(), dst.data_ptr(), a, b, 8.0 * W * H / a / 1e3 / 80, 8.0 * W * H / b / 1e3 / 80), flush=True)
    if k % 2 == 0:
        keep.append((src, dst))          # hold some allocations so that the next ones land elsewhere
    del src, dst
    torch.cuda.empty_cache()
