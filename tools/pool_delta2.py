"""GPU box: the metric's kernel against the placement of its 128 source and 128 result images inside ONE 200-GB buffer:
(a) sources at the start, results at distance D, D = 12 .. 180 GiB; (b) distance fixed, both moved through the buffer.

    python tools/pool_delta2.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca

W, H, C, P = 6000, 4000, 4, 128
IMG = W * H * C
G = 1 << 30
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
big = torch.empty(200 * G, dtype=torch.uint8, device="cuda")
print("buffer at %#x" % big.data_ptr(), flush=True)
seed = torch.empty(IMG * 4, dtype=torch.uint8, device="cuda")
for i in range(4):
    fixca.noise_fill(seed[i * IMG:(i + 1) * IMG].view(H, W, C), 0xF1CA1000 + i)


def t(so, do):
    for i in range(P):                      # the source images: four different ones, repeated (placement is the subject, not content)
        big[so + i * IMG:so + (i + 1) * IMG].copy_(seed[(i % 4) * IMG:(i % 4 + 1) * IMG])
    srcs = [big[so + i * IMG:so + (i + 1) * IMG].view(H, W, C) for i in range(P)]
    dsts = [big[do + i * IMG:do + (i + 1) * IMG].view(H, W, C) for i in range(P)]
    for _ in range(2):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * IMG / (e0.elapsed_time(e1) * 1e3 / (4 * P)) / 1e3 / 80


print("(a) sources at 0, results at D GiB:", flush=True)
print("    " + "  ".join("%d:%.1f" % (d, t(0, d * G)) for d in range(12, 184, 4)), flush=True)
for dist in (12, 24, 48, 96):
    print("(b) results %d GiB after the sources, sources at S GiB:" % dist, flush=True)
    print("    " + "  ".join("%d:%.1f" % (s, t(s * G, (s + dist) * G)) for s in range(0, 188 - dist, 8)), flush=True)
