import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"]="1"
import torch, fixca
W,H=6000,4000
def run(c, kw, n=128, th=None):
    p=fixca.make_params(interpolation=2, **kw)
    src=torch.empty((n,H,W,c),dtype=torch.uint8,device="cuda"); dst=torch.empty_like(src)
    for i in range(n): fixca.noise_fill(src[i], 0xABC0+i)
    s=[src[i] for i in range(n)]; d=[dst[i] for i in range(n)]
    for _ in range(4): fixca.fix_ca_batch(s,d,W,H,c,1,p)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): fixca.fix_ca_batch(s,d,W,H,c,1,p)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/(8*n)
    print("bpp=%d %-60s %.2f us %.1f %%  [%s]"%(c,str(kw),us,2.0*W*H*c/us/1e3/80, fixca.plan(W,H,c,1,p,kernel=fixca.PLAN_BATCH)[1]), flush=True)
    del src,dst
for c in (4,3):
    run(c, dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280))
    run(c, dict(lens_x=3000, lens_y=2000, x_blue=1.0, x_red=0.0, y_blue=-2.5, y_red=2.0))       # no scale: no steps, T-tap path everywhere but the edges
    run(c, dict(blue=0.5, red=-0.5, lens_x=3000, lens_y=2000))                                   # few steps
    run(c, dict(blue=6.0, red=-6.0, lens_x=3000, lens_y=2000))                                   # many steps
