"""Timing experiment (GPU box): run the instrumented build of k_roll (tools/build_variant.sh stamps -DFIXCA_EXPERIMENTS)
on ONE image per launch and print where a wave's cycles go and how the waves of a launch spread over its duration.
    FIXCA_LIB=tools/bin/stamps/libfixca_hip.so python tools/stamps.py [interp]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H = 6000, 4000
interp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=interp)
kid, desc = fixca.plan(W, H, 4, 1, p)
print(desc)
nw = int(desc.split("waves=")[1])
src = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    fixca.fix_ca_region(src, dst, W, H, 4, 1, p, 0, W, 0, H, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for _ in range(N):
    fixca.fix_ca_region(src, dst, W, H, 4, 1, p, 0, W, 0, H, True)
e1.record()
torch.cuda.synchronize()
print("event time per launch (back to back): %.2f us" % (e0.elapsed_time(e1) * 1e3 / N))
a = dst.view(-1)[: nw * 64].cpu().numpy().view(np.uint64).reshape(nw, 8).astype(np.float64)
names = ["outside blocks", "fetch (DMA issue + vm wait)", "LDS reads issued+landed", "filter arithmetic"]
tot = a[:, 6]
print("waves", nw, " mean total cycles (memtime ticks) %.0f  min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i in range(4):
    print("  %-30s %9.0f  %5.1f %%" % (names[i], a[:, i].mean(), 100 * a[:, i].mean() / tot.mean()))
print("  gen-path waves: %.1f %%" % (100 * a[:, 7].mean()))
rt = a[:, 5]
print("  wave lifetime in 100 MHz ticks: mean %.0f (%.1f us)  -> shader clock %.2f GHz" % (rt.mean(), rt.mean() / 100, tot.mean() / rt.mean() * 0.1))
st = a[:, 4]
t0 = st.min()
print("  wave start spread: %.1f us, last wave ends %.1f us after the first starts" % ((st.max() - t0) / 100, ((st + rt).max() - t0) / 100))
start, end = (st - t0) / 100, (st + rt - t0) / 100
print("  starts: p50 %.1f p90 %.1f p99 %.1f max %.1f us;  ends: p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (
    *np.percentile(start, [50, 90, 99, 100]), *np.percentile(end, [1, 10, 50, 90, 99, 100])))
print("  lifetime: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(rt / 100, [10, 50, 90, 100])))
edges = np.arange(0, end.max() + 2, 2.0)
alive = [(int(((start <= e) & (end > e)).sum())) for e in edges]
print("  waves alive every 2 us:", alive)
g = a[:, 7] > 0
if g.any() and (~g).any():
    print("  total fast-path waves %.0f, general-path waves %.0f" % (tot[~g].mean(), tot[g].mean()))
