"""GPU box: per-image time of the >8-bit sample formats (6000x4000, cubic and linear), wide-roll vs the one-thread-per-pixel kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H = 6000, 4000
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
for name, bpc, nch in (("RGBA u16", 2, 4), ("RGBA float", -4, 4), ("RGBA u32", 4, 4), ("RGBA double", -8, 4), ("RGB u16", 2, 3), ("RGB float", -4, 3)):
    if ONLY and ONLY not in name:
        continue
    for interp in ((2, 1) if ONLY else (2, 1, 0)):
        p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=interp)
        sb = abs(bpc)
        B = 4
        raw = torch.empty((B, H, W, nch * sb), dtype=torch.uint8, device="cuda")
        for i in range(B):
            fixca.noise_fill(raw[i], 0xF1CA4000 + i)
        if bpc == -4:
            raw = raw.view(torch.int32).to(torch.float32).abs_().mul_(1.0 / 2147483648.0).view(torch.uint8).reshape(B, H, W, nch * sb)
        if bpc == -8:
            raw = raw.view(torch.int64).to(torch.float64).abs_().mul_(1.0 / 9223372036854775808.0).view(torch.uint8).reshape(B, H, W, nch * sb)
        dst = torch.empty_like(raw)
        for kern, kname in (((fixca.KERNEL_AUTO, "auto"),) if ONLY else ((fixca.KERNEL_AUTO, "auto"), (fixca.KERNEL_EXACT, "exact"))):
            desc = fixca.plan(W, H, nch * sb, bpc, p, kernel=kern)[1]
            for i in range(B):
                fixca.fix_ca_region(raw[i], dst[i], W, H, nch * sb, bpc, p, 0, W, 0, H, True, kern)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            N = 12 if kern == fixca.KERNEL_AUTO else 4
            e0.record()
            for k in range(N):
                fixca.fix_ca_region(raw[k % B], dst[k % B], W, H, nch * sb, bpc, p, 0, W, 0, H, True, kern)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / N
            gbs = 2.0 * nch * sb * W * H / us / 1e3
            print("%-12s interp=%d %-5s %9.1f us/image %7.1f GB/s %5.1f%%  [%s]" % (name, interp, kname, us, gbs, gbs / 80, desc), flush=True)
            if kern == fixca.KERNEL_AUTO:
                keep = dst[0].clone()
            else:
                print("      auto == exact on image 0:", bool(torch.equal(keep, dst[0])))
        del raw, dst
        torch.cuda.empty_cache()
