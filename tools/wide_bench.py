"""GPU box: kernel time of the fp64 kernel for 16/32-bit and float samples (6000x4000 RGBA)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H = 6000, 4000
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
for name, bpc, sb in (("u8 (k_exact)", 1, 1), ("u16", 2, 2), ("u32", 4, 4), ("float", -4, 4), ("double", -8, 8)):
    src = torch.randint(0, 256, (H, W, 4 * sb), dtype=torch.uint8, device="cuda")
    if bpc < 0:
        src = (torch.rand((H, W, 4), device="cuda", dtype=torch.float32 if bpc == -4 else torch.float64)).view(torch.uint8).reshape(H, W, 4 * sb)
    dst = torch.empty_like(src)
    kern = fixca.KERNEL_EXACT if bpc == 1 else fixca.KERNEL_AUTO
    for _ in range(2):
        fixca.fix_ca_region(src, dst, W, H, 4 * sb, bpc, p, 0, W, 0, H, True, kern)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fixca.fix_ca_region(src, dst, W, H, 4 * sb, bpc, p, 0, W, 0, H, True, kern)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print("%-14s %9.1f us  %8.0f Mpix/s  %6.0f GB/s algorithmic" % (name, us, W * H / us, 2.0 * W * H * 4 * sb / us / 1e3))
