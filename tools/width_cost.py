"""GPU box: what an image width costs the main 8-bit kernel, with every width's images carved from ONE pool (the
physical pages of an allocation move the kernel by +-2.5 points, so widths timed in separate processes do not compare).

    python tools/width_cost.py [--bpp 4] [--interp 2] [--images 128] WIDTH ...        (FIXCA_EXP_FORCE_ODD=1 6000: see ab.py)
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"] = "1"
import torch, fixca

ap = argparse.ArgumentParser()
ap.add_argument("--bpp", type=int, default=4)
ap.add_argument("--interp", type=int, default=2)
ap.add_argument("--images", type=int, default=128)
ap.add_argument("--height", type=int, default=4000)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("widths", nargs="+", type=int)
a = ap.parse_args()
H, N, C = a.height, a.images, a.bpp
wmax = max(a.widths)
pool_s = torch.empty(N * wmax * H * C + 4096, dtype=torch.uint8, device="cuda")
pool_d = torch.empty_like(pool_s)
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=a.interp)
res = {w: [] for w in a.widths}
for rnd in range(a.rounds):
    for w in a.widths:
        n = w * H * C
        srcs = [pool_s[i * n:(i + 1) * n].view(H, w, C) for i in range(N)]
        dsts = [pool_d[i * n:(i + 1) * n].view(H, w, C) for i in range(N)]
        for i in range(N):
            fixca.noise_fill(srcs[i], 0xABC0 + i)
        for _ in range(3):
            fixca.fix_ca_batch(srcs, dsts, w, H, C, 1, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fixca.fix_ca_batch(srcs, dsts, w, H, C, 1, p)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (6 * N)
        res[w].append(2.0 * n / us / 1e3 / 80)
for w in a.widths:
    r = sorted(res[w])
    print("bpp=%d interp=%d width %5d (pitch %% 64 = %2d, %% 16 = %2d): median %5.1f %%   rounds: %s   [%s]" % (
        C, a.interp, w, (w * C) % 64, (w * C) % 16, r[len(r) // 2], "  ".join("%.1f" % v for v in res[w]),
        fixca.plan(w, H, C, 1, p, kernel=fixca.PLAN_BATCH)[1]), flush=True)
