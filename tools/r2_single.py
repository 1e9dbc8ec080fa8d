"""GPU box: one image per launch (the plug-in's call), 8 rotated pairs, 200 launches: FIXCA_TH sweep."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
if len(sys.argv) > 1:
    import torch, fixca
    W, H, C = 6000, 4000, 4
    p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
    src = torch.empty((8, H, W, C), dtype=torch.uint8, device="cuda")
    for i in range(8):
        fixca.noise_fill(src[i], 0xF1CA5000 + i)
    dst = torch.empty_like(src)
    srcs, dsts = [src[i] for i in range(8)], [dst[i] for i in range(8)]
    fixca.time_launches(srcs, dsts, 16, W, H, C, 1, p)
    ms = fixca.time_launches(srcs, dsts, 200, W, H, C, 1, p)
    print("%-70s %7.2f us  %5.1f %%" % (fixca.plan(W, H, C, 1, p)[1], ms * 1e3, 2.0 * W * H * C / (ms * 1e-3) / 8e10), flush=True)
else:
    for th in (0, 29, 33, 41, 49, 57, 67, 80, 100, 134):
        env = dict(os.environ, FIXCA_DEBUG="1")
        if th:
            env["FIXCA_TH"] = str(th)
        subprocess.run([sys.executable, __file__, "x"], env=env)
