"""GPU box: does the kernel's speed depend on where dst sits relative to src?  One allocation, src images at
base + i * S, dst images at base + N * S + delta + i * S; the same process (same physical pages) for every delta."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C, N = 6000, 4000, 4, 32
S = W * H * C
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
slack = 9 << 30
big = torch.empty(2 * N * S + 2 * slack, dtype=torch.uint8, device="cuda")
base_addr = big.data_ptr()
print("base %x (mod 2 MiB: %x, mod 1 GiB: %x)" % (base_addr, base_addr % (2 << 20), base_addr % (1 << 30)))
def views(off):
    return [big[off + i * S: off + (i + 1) * S].view(H, W, C) for i in range(N)]
srcs = views(0)
for i, s in enumerate(srcs):
    fixca.noise_fill(s, 0xF1CA7000 + i)
def run(dsts, reps=6):
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
deltas = [0, 1 << 21, 1 << 24, 1 << 26, 1 << 27, 1 << 28, 1 << 29, 1 << 30, 3 << 29, 1 << 31, 3 << 30, 1 << 32, 5 << 30, 3 << 31, 7 << 30, 1 << 33, 0, 1 << 30]
for d in deltas:
    us = run(views(N * S + d))
    print("delta %9d : %6.2f us/image  %5.1f %%   (dst - src) mod 4096 = %4d, mod 64 KiB = %6d, mod 2 MiB = %8d" % (d, us, 8.0 * W * H / us / 1e3 / 80, (N * S + d) % 4096, (N * S + d) % 65536, (N * S + d) % (2 << 20)), flush=True)
