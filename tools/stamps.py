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
# NPAIRS image pairs in rotation: with one pair the 192 MB stay in the 256-MB Infinity Cache and a launch takes 36-40 us
# instead of the 43-45 us it takes on images that come from HBM (STAMPS_PAIRS=1 shows that regime)
NPAIRS = int(os.environ.get("STAMPS_PAIRS", "64"))
srcs = [torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(NPAIRS)]
dsts = [torch.empty_like(srcs[0]) for _ in range(NPAIRS)]
for k in range(min(NPAIRS, 8)):
    fixca.fix_ca_region(srcs[k], dsts[k], W, H, 4, 1, p, 0, W, 0, H, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = max(64, 2 * NPAIRS)
e0.record()
for k in range(N):
    fixca.fix_ca_region(srcs[k % NPAIRS], dsts[k % NPAIRS], W, H, 4, 1, p, 0, W, 0, H, True)
e1.record()
torch.cuda.synchronize()
print("event time per launch (back to back, %d image pairs in rotation): %.2f us" % (NPAIRS, e0.elapsed_time(e1) * 1e3 / N))
src, dst = srcs[-1], dsts[-1]
for k in range(NPAIRS - 1):                          # the measured launch comes after launches on the other pairs
    fixca.fix_ca_region(srcs[k], dsts[k], W, H, 4, 1, p, 0, W, 0, H, True)
dst.zero_()
fixca.fix_ca_region(src, dst, W, H, 4, 1, p, 0, W, 0, H, True)
torch.cuda.synchronize()
nb = (nw + 7) // 8 * 8                               # workgroups of the launch; those without a tile record nothing
rec = dst.view(-1)[: nb * 64].cpu().numpy().view(np.uint64).reshape(nb, 8)
valid = (rec[:, 6] != 0) & (rec[:, 5] < 1000000) & (rec[:, 6] < 100000000)     # a workgroup without a tile leaves image bytes there
rec = rec[valid]
nw = len(rec)
a = rec.astype(np.float64)
names = ["outside blocks", "fetch (DMA issue + vm wait)", "LDS reads issued+landed", "filter arithmetic"]
tot = a[:, 6]
print("waves", nw, " mean total cycles (memtime ticks) %.0f  min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i in range(4):
    print("  %-30s %9.0f  %5.1f %%" % (names[i], a[:, i].mean(), 100 * a[:, i].mean() / tot.mean()))
tag = rec[:, 7]
a[:, 7] = (tag & np.uint64(0xff)).astype(np.float64)
strip, row0 = ((tag >> np.uint64(8)) & np.uint64(0xfff)).astype(int), ((tag >> np.uint64(20)) & np.uint64(0xfffff)).astype(int)
xcc, hwid = ((tag >> np.uint64(40)) & np.uint64(15)).astype(int), ((tag >> np.uint64(44)) & np.uint64(0xfff)).astype(int)
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

# where the slow waves are: lifetime by XCD, by strip (x position), by tile row (y position), by CU; and whether the same
# tiles are slow in another launch (systematic) or not (the memory system's noise)
life = rt / 100
def table(name, key):
    ks = sorted(set(key.tolist()))
    print("  lifetime by %s:" % name, " ".join("%s:%.1f" % (k, life[key == k].mean()) for k in ks[:40]))
table("XCD", xcc)
table("strip", strip)
table("tile row / 8", (row0 // max(1, (H // 8))))
cu = hwid >> 4                                   # CU, SH, SE bits of HW_ID
print("  lifetime by CU (SE/SH/CU id): min %.1f max %.1f over %d ids; the 5 slowest: %s" % (
    min(life[cu == k].mean() for k in set(cu.tolist())), max(life[cu == k].mean() for k in set(cu.tolist())), len(set(cu.tolist())),
    sorted(((round(life[cu == k].mean(), 1), int(k)) for k in set(cu.tolist())), reverse=True)[:5]))
dst.zero_()
for k in range(NPAIRS - 1):
    fixca.fix_ca_region(srcs[k], dsts[k], W, H, 4, 1, p, 0, W, 0, H, True)
fixca.fix_ca_region(src, dst, W, H, 4, 1, p, 0, W, 0, H, True)
torch.cuda.synchronize()
b = dst.view(-1)[: nb * 64].cpu().numpy().view(np.uint64).reshape(nb, 8)[valid]
life2 = b[:, 5].astype(np.float64) / 100
xcc2 = ((b[:, 7] >> np.uint64(40)) & np.uint64(15)).astype(int)
print("  correlation of a tile's lifetime between two launches: %.2f; same XCD in both: %.0f %% of the tiles" % (np.corrcoef(life, life2)[0, 1], 100 * (xcc == xcc2).mean()))
chunk = row0 // int(desc.split("th=")[1].split()[0])
print("  lifetime by chunk (mean over its strips):", " ".join("%.1f" % life[chunk == k].mean() for k in sorted(set(chunk.tolist()))))
print("  spread inside a chunk (std over its strips): mean %.2f us; spread of the chunk means: %.2f us; inside a strip over chunks: %.2f us" % (
    np.mean([life[chunk == k].std() for k in set(chunk.tolist())]), np.std([life[chunk == k].mean() for k in set(chunk.tolist())]),
    np.mean([life[strip == k].std() for k in set(strip.tolist())])))
endt = (st + rt - t0) / 100
print("  end time by chunk:", " ".join("%.1f" % endt[chunk == k].mean() for k in sorted(set(chunk.tolist()))))
simd = hwid & 3
import collections
key = (xcc.astype(np.int64) << 20) | (hwid.astype(np.int64) >> 4 << 4) | simd          # XCD, SE/SH/CU, SIMD
cnt = collections.Counter(key.tolist())
share = np.array([cnt[k] for k in key.tolist()])
print("  lifetime by SIMD id:", " ".join("%d:%.1f(n=%d)" % (k, life[simd == k].mean(), (simd == k).sum()) for k in range(4)))
print("  lifetime by waves sharing the wave's SIMD:", " ".join("%d:%.1f(n=%d)" % (k, life[share == k].mean(), (share == k).sum()) for k in sorted(set(share.tolist()))))
cukey = (xcc.astype(np.int64) << 20) | (hwid.astype(np.int64) >> 4)
cucnt = collections.Counter(cukey.tolist())
percu = np.array([cucnt[k] for k in cukey.tolist()])
print("  waves per CU:", dict(collections.Counter(cucnt.values())), " lifetime by waves on the wave's CU:", " ".join("%d:%.1f" % (k, life[percu == k].mean()) for k in sorted(set(percu.tolist()))))
j = np.zeros(nw, int)                                # dispatch order inside the XCD: tile = xcd * (nblocks / 8) + j
nb8 = ((nw + 7) // 8)
tile = chunk * int(desc.split("strips=")[1].split()[0]) + strip
j = tile % nb8
print("  lifetime by dispatch order inside the XCD (32 at a time):", " ".join("%.1f" % life[(j // 32) == k].mean() for k in range((nb8 + 31) // 32)))
print("  SIMD id by dispatch order (32 at a time, most common):", " ".join("%d" % collections.Counter(simd[(j // 32) == k].tolist()).most_common(1)[0][0] for k in range((nb8 + 31) // 32)))
