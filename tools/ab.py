"""GPU box: A/B of library builds and tuning knobs ON ONE ALLOCATION.

The kernels' speed moves by +-2.5 points with the physical pages a buffer lands on (profiles/NOTES.md), so variants
are only comparable when they share the buffers: all libraries are loaded into this process (each has its own
planner / table caches; they share the HIP runtime), the source and result images are allocated once, and the
variants are timed in rotation, `--rounds` times.  Prints per variant the per-round percentages of the HBM peak
(8 TB/s over 2 x bytes per image) and their median, and checks once that every variant computes HEAD's pixels.

    python tools/ab.py [--bpp 4] [--bpc 1] [--interp 2] [--images 128] [--rounds 4] [--single] [--blue 2 --red -2] VARIANT ...

    VARIANT = name[@lib][,ENV=value ...][,per=N]
        lib   directory under tools/bin/ holding a libfixca_hip.so built by tools/build_variant.sh (default: the
              in-tree library)
        ENV   FIXCA_* tuning variables for that variant (FIXCA_DEBUG=1 is set)
        per   images per fixca_batch_device call (default 128; --single: one fixca_region_device call per image,
              what fix_ca_region does for the plug-in)

    python tools/ab.py --bpp 3 HEAD th21,FIXCA_TH=21 cap168@rgb_cap168 HEAD2
"""
import argparse, ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"] = "1"
import torch, fixca

ap = argparse.ArgumentParser()
ap.add_argument("--bpp", type=int, default=4, help="samples per pixel (3 RGB, 4 RGBA)")
ap.add_argument("--bpc", type=int, default=1, help="color_size code: 1 u8, 2 u16, 4 u32, -4 float, -8 double")
ap.add_argument("--interp", type=int, default=2)
ap.add_argument("--images", type=int, default=128)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--width", type=int, default=6000)
ap.add_argument("--height", type=int, default=4000)
ap.add_argument("--blue", type=float, default=2.0, help="lateral amounts (the headline workload: blue 2, red -2)")
ap.add_argument("--red", type=float, default=-2.0)
ap.add_argument("--single", action="store_true", help="one image per launch through fixca_region_device")
ap.add_argument("--grid", default="", help="cartesian product of variants: 'lib=HEAD,d10;FIXCA_TH=0,64,160;per=8' (0 / HEAD = unset / in-tree library)")
ap.add_argument("variants", nargs="*")
a = ap.parse_args()
W, H, N = a.width, a.height, a.images
SB = abs(a.bpc)
PB = a.bpp * SB
BIN = os.path.join(ROOT, "tools", "bin")

def parse(v):
    parts = v.split(",")
    name, lib = (parts[0].split("@") + [None])[:2]
    env, per = {}, 128
    for kv in parts[1:]:
        k, val = kv.split("=", 1)
        if k == "per":
            per = int(val)
        else:
            env[k] = val
    return v.replace("FIXCA_", ""), fixca.LIB_PATH if lib is None else os.path.join(BIN, lib, "libfixca_hip.so"), env, per
def grid(spec):
    import itertools
    axes = [ax.split("=", 1) for ax in spec.split(";") if ax]
    out = []
    for combo in itertools.product(*[vals.split(",") for _, vals in axes]):
        lib = ""
        parts = []
        for (k, _), v in zip(axes, combo):
            if k == "lib":
                lib = "" if v == "HEAD" else "@" + v
            elif v != "0" or k == "per":
                parts.append("%s=%s" % (k, v))
        out.append(",".join(["g" + lib] + parts))
    return out
VARIANTS = [parse(v) for v in a.variants + (grid(a.grid) if a.grid else [])]

libs = {}
def lib_for(path):
    if path not in libs:
        l = ctypes.CDLL(path)
        vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
        P = ctypes.POINTER(fixca.FixCaParams)
        l.fixca_batch_device.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, P, cu, vp]
        l.fixca_batch_device.restype = ci
        l.fixca_region_device.argtypes = [vp, vp, ci, ci, ci, ci, P, ci, ci, ci, ci, ci, cu, vp]
        l.fixca_region_device.restype = ci
        l.fixca_plan_describe.argtypes = [ci, ci, ci, ci, P, ci, ci, ci, ci, ci, cu, ctypes.c_char_p, ctypes.c_size_t]
        l.fixca_last_error.restype = ctypes.c_char_p
        libs[path] = l
    return libs[path]

p = fixca.make_params(blue=a.blue, red=a.red, lens_x=658, lens_y=1280, interpolation=a.interp)
src = torch.empty((N, H, W, PB), dtype=torch.uint8, device="cuda")
for i in range(N):
    fixca.noise_fill(src[i], 0xF1CAA000 + i)
if a.bpc == -4:
    src = src.view(torch.int32).to(torch.float32).abs_().mul_(1.0 / 2147483648.0).view(torch.uint8).reshape(N, H, W, PB)
elif a.bpc == -8:
    src = src.view(torch.int64).to(torch.float64).abs_().mul_(1.0 / 9223372036854775808.0).view(torch.uint8).reshape(N, H, W, PB)
dst = torch.empty_like(src)
vp = ctypes.c_void_p
sp = (vp * N)(*[src[i].data_ptr() for i in range(N)])
dp = (vp * N)(*[dst[i].data_ptr() for i in range(N)])
stream = vp(torch.cuda.current_stream().cuda_stream)

def one_pass(lib, per):
    if a.single:
        for k in range(N):
            rc = lib.fixca_region_device(sp[k], dp[k], W, H, PB, a.bpc, ctypes.byref(p), 0, W, 0, H, 1, 0, stream)
            if rc < 0:
                raise RuntimeError(lib.fixca_last_error().decode())
        return
    for k in range(0, N, per):
        m = min(per, N - k)
        rc = lib.fixca_batch_device(ctypes.cast(ctypes.byref(sp, k * 8), ctypes.POINTER(vp)), ctypes.cast(ctypes.byref(dp, k * 8), ctypes.POINTER(vp)),
                                    m, W, H, PB, a.bpc, ctypes.byref(p), 0, stream)
        if rc < 0:
            raise RuntimeError(lib.fixca_last_error().decode())

class Env:
    def __init__(self, env): self.env = env
    def __enter__(self):
        self.saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
    def __exit__(self, *x):
        for k, v in self.saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

host_us = {}     # what the host spends issuing one image's work (the queue is never empty when this is below the device time)
def timed(lib, env, per):
    # short timed regions read slow (the clocks ramp up after every pause: 16 images per pass measured 3-8 points low
    # in the first variant after a comparison): at least 64 images untimed, then at least 256 timed
    warm, reps = max(1, 64 // N), max(a.reps, 256 // N)
    with Env(env):
        for _ in range(warm):
            one_pass(lib, per)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(reps):
            one_pass(lib, per)
        host_us.setdefault(id(lib), {})[tuple(sorted(env.items())) + (per,)] = (time.perf_counter() - t0) * 1e6 / (reps * N)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * N)
    return us

def plan_of(lib, env, per):
    buf = ctypes.create_string_buffer(256)
    with Env(env):
        lib.fixca_plan_describe(W, H, PB, a.bpc, ctypes.byref(p), 0, W, 0, H, 1, 0 if (a.single or per == 1) else 4, buf, 256)
    return buf.value.decode()

res = {v[0]: [] for v in VARIANTS}
# one untimed pass of every variant first: the first timed variant otherwise pays for clocks that ramp up after the idle
# time of the allocation and for first-launch costs (code objects, tables)
for name, path, env, per in VARIANTS:
    if os.path.exists(path):
        with Env(env):
            one_pass(lib_for(path), per)
torch.cuda.synchronize()
for r in range(a.rounds):
    for name, path, env, per in VARIANTS:
        if os.path.exists(path):
            res[name].append(timed(lib_for(path), env, per))
    if r == 0:
        name0, path0, env0, per0 = VARIANTS[0]
        with Env(env0):
            one_pass(lib_for(path0), per0)
        torch.cuda.synchronize()
        ref = dst[N - 1].clone()
        for name, path, env, per in VARIANTS[1:]:
            if os.path.exists(path):
                dst[N - 1].zero_()
                with Env(env):
                    one_pass(lib_for(path), per)
                torch.cuda.synchronize()
                if not torch.equal(dst[N - 1], ref):
                    print("!! %s: output differs from %s in %d bytes" % (name, name0, int((dst[N - 1] != ref).sum())))
        # the comparison left the device idle: one untimed measurement brings the clocks back, or the next round's first
        # variant reads 2-4 points low (seen on every first variant of round 2 before this line existed)
        timed(lib_for(path0), env0, per0)
algo = 2.0 * PB * W * H
# the box's own yardstick on the same buffers: the library's 16-byte-per-lane nt copy kernel
copy_ms = fixca.time_copy([src[i] for i in range(N)], [dst[i] for i in range(N)], max(64, 4 * N), W * H * PB)
print("%dx%d bpp=%d bpc=%d interp=%d, %d images, %s; nt copy of the same buffers: %.1f %%" % (
    W, H, a.bpp, a.bpc, a.interp, N, "one image per launch" if a.single else "batch entry", algo / (copy_ms * 1e3) / 1e3 / 80))
for name, path, env, per in VARIANTS:
    v = res[name]
    if not v:
        print("%-44s (library missing: %s)" % (name, path))
        continue
    pct = [algo / us / 1e3 / 80 for us in v]
    hu = host_us.get(id(lib_for(path)), {}).get(tuple(sorted(env.items())) + (per,), float("nan"))
    print("%-44s median %6.2f us %5.1f %%   rounds: %s   host %.1f us/image   [%s]" % (name, statistics.median(v), statistics.median(pct), " ".join("%5.1f" % x for x in pct),
                                                                 hu, plan_of(lib_for(path), env, per)), flush=True)
