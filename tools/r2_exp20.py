"""GPU box: allocation strategies vs the two speed modes.  Each strategy allocated and measured 4 times in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C, N = 6000, 4000, 4, 64
S = W * H * C
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
def measure(srcs, dsts, reps=5):
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * N)
    return 8.0 * W * H / us / 1e3 / 80
def two_blocks():
    src = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    return [src[i] for i in range(N)], [dst[i] for i in range(N)], (src, dst)
def per_image():
    s = [torch.empty((H, W, C), dtype=torch.uint8, device="cuda") for _ in range(N)]
    d = [torch.empty((H, W, C), dtype=torch.uint8, device="cuda") for _ in range(N)]
    return s, d, (s, d)
def per_image_interleaved():
    s, d = [], []
    for _ in range(N):
        s.append(torch.empty((H, W, C), dtype=torch.uint8, device="cuda")); d.append(torch.empty((H, W, C), dtype=torch.uint8, device="cuda"))
    return s, d, (s, d)
def padded_2mb():
    P = ((S + (2 << 20) - 1) // (2 << 20)) * (2 << 20)
    a = torch.empty(N * P, dtype=torch.uint8, device="cuda"); b = torch.empty(N * P, dtype=torch.uint8, device="cuda")
    return [a[i * P: i * P + S].view(H, W, C) for i in range(N)], [b[i * P: i * P + S].view(H, W, C) for i in range(N)], (a, b)
def one_block_src_then_dst():
    a = torch.empty(2 * N * S, dtype=torch.uint8, device="cuda")
    return [a[i * S:(i + 1) * S].view(H, W, C) for i in range(N)], [a[(N + i) * S:(N + i + 1) * S].view(H, W, C) for i in range(N)], (a,)
for name, fn in (("two blocks (bench.py)", two_blocks), ("one tensor per image", per_image), ("per image, src/dst interleaved", per_image_interleaved),
                 ("images padded to 2 MiB", padded_2mb), ("one block: all src then all dst", one_block_src_then_dst)):
    res = []
    for k in range(4):
        srcs, dsts, hold = fn()
        for i, s in enumerate(srcs):
            fixca.noise_fill(s, 0xF1CA9000 + i)
        res.append(measure(srcs, dsts))
        del srcs, dsts, hold
        torch.cuda.empty_cache()
    print("%-34s : %s" % (name, "  ".join("%5.1f" % r for r in res)), flush=True)
