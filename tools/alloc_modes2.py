"""GPU box: is the metric kernel's speed tied to HOW a pool of images was allocated?  (bench.py's `configs` leg reads the
metric's own configuration 3 points under the headline of the same run: 69.5 against 72.3 %.)  One process:
  A  128 images allocated first thing
  B  512 images (the bench's pool), timed over its first 128 and over all 512
  C  128 images allocated after B was freed (torch.cuda.empty_cache: hipFree, then hipMalloc again)
  D  128 more while C is alive
Every measurement: 8 untimed launches of 128 images, then 24 timed (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
import torch, fixca
W, H, C = 6000, 4000, 4
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)

def pool(n):
    src = torch.empty((n, H, W, C), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for i in range(n):
        fixca.noise_fill(src[i], 0xF1CA1000 + i)
    return src, dst

def measure(name, src, dst, lo, hi):
    groups = [([src[i] for i in range(k, k + 128)], [dst[i] for i in range(k, k + 128)]) for k in range(lo, hi, 128)]
    def launch(j):
        s, d = groups[j % len(groups)]
        fixca.fix_ca_batch(s, d, W, H, C, 1, p)
    for j in range(8):
        launch(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(24):
        launch(j)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 24
    print("%-44s src %#x dst %#x  %.1f us per launch  %.1f %%" % (name, src.data_ptr(), dst.data_ptr(), us, 2.0 * 128 * W * H * C / us / 1e3 / 80.0), flush=True)

a = pool(128); measure("A: 128 images, first allocation", *a, 0, 128)
b = pool(512); measure("B: 512 images, its first 128", *b, 0, 128); measure("B: 512 images, all in rotation", *b, 0, 512)
measure("A again", *a, 0, 128)
del b; torch.cuda.empty_cache()
c = pool(128); measure("C: 128 images after B was freed", *c, 0, 128)
d = pool(128); measure("D: 128 more, C alive", *d, 0, 128)
measure("A again", *a, 0, 128)
del a, c, d; torch.cuda.empty_cache()
e = pool(128); measure("E: 128 images after everything was freed", *e, 0, 128)
