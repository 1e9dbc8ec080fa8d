"""GPU box: does a balanced static partition (every CU the same number of equal tiles, all resident) beat
dynamically dispatched short tiles on the REAL kernel?  W = 7680 = 32 strips of 240, 16 images ->
512 columns; th = 1000 -> 2048 tiles = 8 per CU."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
if len(sys.argv) > 1:
    import torch, fixca
    W, H, C, B = 7680, 4000, 4, 16
    p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
    kid, desc = fixca.plan(W, H, C, 1, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)
    srcs = [torch.randint(0, 256, (H, W, C), dtype=torch.uint8, device="cuda") for _ in range(B)]
    dsts = [torch.empty_like(s) for s in srcs]
    for _ in range(3):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 20
    e0.record()
    for _ in range(N):
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, p)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (N * B)
    gbs = 2.0 * C * W * H / us / 1e3
    print("%-70s %8.2f us/image  %5.1f %% of 8 TB/s" % (desc, us, gbs / 80), flush=True)
else:
    for rep in range(2):
        for th in (33, 1000, 1336, 800, 667, 57):
            env = dict(os.environ, FIXCA_TH=str(th))
            subprocess.run([sys.executable, __file__, "x"], env=env)
