"""GPU box: soak of the 16-bit weight forms (k_wcol_weights: CUBIC; k_lin<u16, FAST>: LINEAR with SOAK_INTERP=1) against the one-thread-per-pixel fp64 kernel: N random parameter
sets (lateral amounts up to the plug-in's +-30, lens anywhere, some with directional shifts) on 6000x4000 RGBA / RGB u16
images of four kinds (noise, smooth ramps, ramps with a little noise, random 0 / 65535 steps).  Every sample must be identical.
SOAK_BATCH=1: through the batch entry (two result images per call), where k_wcol runs several images per launch and 16-bit
LINEAR takes k_lin's weight form by itself.
usage: wide_soak.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import numpy as np, torch, fixca
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
W, H = 6000, 4000
rng = np.random.default_rng(0x50AC)
total = bad = weights_runs = 0
for it in range(N):
    nch = 3 if it % 4 == 3 else 4
    kind = it % 4 if N > 12 else it % 3
    if kind == 3:
        img = (torch.randint(0, 2, (H, W, nch), device="cuda", dtype=torch.int32) * 65535).to(torch.int16)      # extreme steps
        raw = img.contiguous().view(torch.uint8).reshape(H, W, nch * 2)
    elif kind == 0:
        raw = torch.empty((H, W, nch * 2), dtype=torch.uint8, device="cuda"); fixca.noise_fill(raw, 0x50AC0000 + it)
        img = raw.view(torch.int16).reshape(H, W, nch)
    else:
        yy = torch.arange(H, device="cuda", dtype=torch.float64)[:, None, None]; xx = torch.arange(W, device="cuda", dtype=torch.float64)[None, :, None]
        cc = torch.arange(nch, device="cuda", dtype=torch.float64)[None, None, :]
        v = (xx * (5.0 + it) + yy * (3.0 + 0.5 * it) + cc * 4099.0 + (xx * yy) / 977.0) % 65536.0
        if kind == 2:
            v = (v + torch.randint(0, 5, (H, W, nch), device="cuda").to(torch.float64)) % 65536.0
        img = v.to(torch.int32).to(torch.int16)       # bit pattern of the u16 value
        raw = img.contiguous().view(torch.uint8).reshape(H, W, nch * 2)
    mag = [0.5, 2.0, 8.0, 30.0][it % 4]
    kw = dict(blue=float(rng.uniform(-mag, mag)), red=float(rng.uniform(-mag, mag)), lens_x=float(rng.integers(-1, W + 1)), lens_y=float(rng.integers(-1, H + 1)))
    if it % 5 == 0:
        kw.update(x_blue=float(rng.uniform(-mag, mag)), y_red=float(rng.uniform(-mag, mag)))
    p = fixca.make_params(interpolation=int(os.environ.get("SOAK_INTERP", "2")), **kw)
    batch = os.environ.get("SOAK_BATCH", "0") == "1"
    desc = fixca.plan(W, H, nch * 2, 2, p, kernel=fixca.KERNEL_AUTO | (fixca.PLAN_BATCH if batch else 0))[1]
    weights_runs += "weights+recheck" in desc
    a = torch.zeros_like(raw); b = torch.zeros_like(raw)
    if batch:
        a2 = torch.zeros_like(raw)
        fixca.fix_ca_batch([raw, raw], [a, a2], W, H, nch * 2, 2, p)
    else:
        fixca.fix_ca_region(raw, a, W, H, nch * 2, 2, p, 0, W, 0, H, True, fixca.KERNEL_AUTO)
    fixca.fix_ca_region(raw, b, W, H, nch * 2, 2, p, 0, W, 0, H, True, fixca.KERNEL_EXACT)
    torch.cuda.synchronize()
    d = int((a.view(torch.int16) != b.view(torch.int16)).sum())
    if batch:
        d += int((a2.view(torch.int16) != b.view(torch.int16)).sum())
    total += H * W * 2; bad += d
    print("%2d nch=%d kind=%d %-60s differing %d  [%s]" % (it, nch, kind, str({k: round(v, 3) for k, v in kw.items()})[:60], d, desc[-40:]), flush=True)
print("red/blue samples compared: %d, differing: %d, runs on the weight form: %d of %d" % (total, bad, weights_runs, N))
sys.exit(1 if bad else 0)
