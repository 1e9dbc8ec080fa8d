"""Debug helper (GPU box): where does the fast kernel differ from the fp64 kernel?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
from fixca import synth
W, H = int(sys.argv[1]), int(sys.argv[2])
interp = int(sys.argv[3])
kw = dict(blue=2.0, red=-2.0, lens_x=float(sys.argv[4]), lens_y=float(sys.argv[5]), interpolation=interp)
p = fixca.make_params(**kw)
print(fixca.plan(W, H, 4, 1, p))
img = synth.noise_image(0xF1CA0001, W, H, 4)
src = torch.from_numpy(img).cuda()
a = torch.empty_like(src); b = torch.empty_like(src)
fixca.fix_ca_region(src, a, W, H, 4, 1, p, 0, W, 0, H, True, fixca.KERNEL_EXACT)
fixca.fix_ca_region(src, b, W, H, 4, 1, p, 0, W, 0, H, True, fixca.KERNEL_FAST)
torch.cuda.synchronize()
d = (a.to(torch.int16) - b.to(torch.int16)).abs().cpu().numpy()
bad = d > 1
print("bad samples", bad.sum(), "of", bad.size)
if bad.any():
    ys, xs, cs = np.nonzero(bad)
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "channels", np.unique(cs))
    rows = np.unique(ys); print("n bad rows", len(rows), rows[:40])
    cols = np.unique(xs); print("n bad cols", len(cols), cols[:40])
    print("bad per row (first rows)", [(int(r), int((ys == r).sum())) for r in rows[:10]])
