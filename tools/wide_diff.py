"""GPU box: where does k_wcol differ from the one-thread-per-pixel kernel?  usage: wide_diff.py bpc nch interp [W H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
bpc, nch, interp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (6000, 4000)
sb = abs(bpc)
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=interp)
raw = torch.empty((H, W, nch * sb), dtype=torch.uint8, device="cuda")
fixca.noise_fill(raw, 0xF1CA4000)
if bpc == -4:
    raw = raw.view(torch.int32).to(torch.float32).abs_().mul_(1.0 / 2147483648.0).view(torch.uint8).reshape(H, W, nch * sb)
if bpc == -8:
    raw = raw.view(torch.int64).to(torch.float64).abs_().mul_(1.0 / 9223372036854775808.0).view(torch.uint8).reshape(H, W, nch * sb)
a = torch.zeros_like(raw); b = torch.zeros_like(raw)
print(fixca.plan(W, H, nch * sb, bpc, p, kernel=fixca.KERNEL_AUTO)[1])
fixca.fix_ca_region(raw, a, W, H, nch * sb, bpc, p, 0, W, 0, H, True, fixca.KERNEL_AUTO)
fixca.fix_ca_region(raw, b, W, H, nch * sb, bpc, p, 0, W, 0, H, True, fixca.KERNEL_EXACT)
torch.cuda.synchronize()
d = (a != b).reshape(H, W, nch, sb).any(dim=3)
n = int(d.sum())
print("differing samples:", n)
if n:
    idx = d.nonzero()
    print("rows", int(idx[:, 0].min()), "..", int(idx[:, 0].max()), " cols", int(idx[:, 1].min()), "..", int(idx[:, 1].max()), " channels", sorted(set(idx[:, 2].tolist())))
    rows = torch.unique(idx[:, 0]); cols = torch.unique(idx[:, 1])
    print("distinct rows", len(rows), rows[:20].tolist(), " distinct cols", len(cols), cols[:20].tolist())
    y, x, c = idx[0].tolist()
    dt = {2: torch.int16, 4: torch.int32, -4: torch.float32, -8: torch.float64}[bpc]
    print("first:", y, x, c, a.view(dt).reshape(H, W, nch)[y, x].tolist(), b.view(dt).reshape(H, W, nch)[y, x].tolist())
