"""GPU box: the bench's job (512 source + 512 result images 6000x4000 RGBA8, 128 per launch) with the two pools placed
(a) as two tensors allocated one after the other (what bench.py did until now), (b) inside ONE buffer, the result pool D GiB
after the start of the source pool, D = 46 (adjacent) .. 128.  Per placement: the four groups' fractions and their mean.

    python tools/pool_place.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca

W, H, C, N, P = 6000, 4000, 4, 512, 128
IMG = W * H * C
G = 1 << 30
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)


def groups(srcs, dsts):
    fr = []
    for k in range(0, N, P):
        s, d = srcs[k:k + P], dsts[k:k + P]
        for _ in range(2):
            fixca.fix_ca_batch(s, d, W, H, C, 1, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fixca.fix_ca_batch(s, d, W, H, C, 1, p)
        e1.record()
        torch.cuda.synchronize()
        fr.append(2.0 * IMG / (e0.elapsed_time(e1) * 1e3 / (5 * P)) / 1e3 / 80)
    return fr


def show(name, fr):
    print("%-58s groups %s   mean %.1f %%" % (name, "  ".join("%.1f" % f for f in fr), sum(fr) / len(fr)), flush=True)


order = sys.argv[1] if len(sys.argv) > 1 else "two-first"
if order == "two-first":
    a = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for i in range(N):
        fixca.noise_fill(a[i], 0xF1CA1000 + i)
    show("two tensors, allocated one after the other", groups([a[i] for i in range(N)], [b[i] for i in range(N)]))
    del a, b
    torch.cuda.empty_cache()
big = torch.empty(176 * G, dtype=torch.uint8, device="cuda")
srcs = [big[i * IMG:(i + 1) * IMG].view(H, W, C) for i in range(N)]
for i in range(N):
    fixca.noise_fill(srcs[i], 0xF1CA1000 + i)
for d in (46, 56, 64, 72, 80, 96, 128):
    o = d * G
    show("one buffer, results %3d GiB after the sources' start" % d, groups(srcs, [big[o + i * IMG:o + (i + 1) * IMG].view(H, W, C) for i in range(N)]))
