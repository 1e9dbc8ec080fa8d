import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
from fixca import synth
W, H, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
interp = int(sys.argv[4])
kw = dict(blue=float(sys.argv[5]), red=float(sys.argv[6]), lens_x=float(sys.argv[7]), lens_y=float(sys.argv[8]), interpolation=interp)
p = fixca.make_params(**kw)
print(fixca.plan(W, H, C, 1, p))
img = synth.noise_image(1, W, H, C)
src = torch.from_numpy(img).cuda()
a = torch.empty_like(src); b = torch.empty_like(src)
fixca.fix_ca_region(src, a, W, H, C, 1, p, 0, W, 0, H, True, fixca.KERNEL_EXACT)
fixca.fix_ca_region(src, b, W, H, C, 1, p, 0, W, 0, H, True, fixca.KERNEL_AUTO)
torch.cuda.synchronize()
d = (a.to(torch.int16) - b.to(torch.int16)).abs().cpu().numpy()
bad = d > (0 if interp == 0 else 1)
print("bad samples", bad.sum(), "of", bad.size)
if bad.any():
    ys, xs, cs = np.nonzero(bad)
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "channels", np.unique(cs))
    print("bad cols first", np.unique(xs)[:48])
    print("bad rows first", np.unique(ys)[:20])
    y, x = ys[0], xs[0]
    print("at", y, x, "exact", a.cpu().numpy()[y, x-1:x+3].tolist(), "fast", b.cpu().numpy()[y, x-1:x+3].tolist())
