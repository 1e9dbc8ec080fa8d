"""RGB8 copy of tools/ab.py (3 bytes per pixel).  GPU box: A/B of library builds and tuning knobs ON ONE ALLOCATION (the kernel's speed moves by +-2.5 points
with the physical pages a buffer lands on, DESIGN.md §4, so variants must share the buffers).

    python tools/ab.py [rounds] [structure|knobs|builds]

Every variant = (library path, environment overrides, images per launch).  All libraries are loaded into this
process (each has its own planner / table caches; they share the HIP runtime), the 128 source and result images
are allocated once, and the variants are timed in rotation, `rounds` times.  Prints per variant the per-round
percentages of the HBM peak and their median.
"""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"] = "1"
import torch, fixca

W, H, C, N = 6000, 4000, 3, 128
BIN = os.path.join(ROOT, "tools", "bin")
HEAD = fixca.LIB_PATH
def V(name, lib=None, env=None, per=64):
    return (name, HEAD if lib is None else os.path.join(BIN, lib, "libfixca_hip.so"), env or {}, per)
SETS = {
 "one": [V("HEAD")],
 "builds": [V("HEAD"), V("RDEPTH=8 ring=18 th=25", "ab_rd8", {"FIXCA_RING": "18", "FIXCA_TH": "25"}), V("RDEPTH=8 ring=20 th=25", "ab_rd8", {"FIXCA_RING": "20", "FIXCA_TH": "25"}),
            V("RDEPTH=10 ring=20 th=25", "ab_rd10", {"FIXCA_RING": "20", "FIXCA_TH": "25"}), V("RDEPTH=4 th=25", "ab_rd4", {"FIXCA_TH": "25"}),
            V("RDEPTH=4 ring=12 th=25", "ab_rd4", {"FIXCA_RING": "12", "FIXCA_TH": "25"}), V("ring=14 th=25", env={"FIXCA_RING": "14", "FIXCA_TH": "25"}),
            V("ring=12 th=21", env={"FIXCA_RING": "12", "FIXCA_TH": "21"}), V("th=17", env={"FIXCA_TH": "17"}), V("th=21", env={"FIXCA_TH": "21"}), V("HEAD (again)")],
 "nrb": [V("HEAD"), V("NRB=8", "ab_nrb8"), V("NRB=6", "ab_nrb6"), V("NRB=2", "ab_nrb2"), V("NRB=2 ring=14", "ab_nrb2", {"FIXCA_RING": "14"}), V("NRB=8 ring=20", "ab_nrb8", {"FIXCA_RING": "20"}), V("HEAD (again)")],
 "knobs": [V("HEAD"), V("th=21", env={"FIXCA_TH": "21"}), V("th=25", env={"FIXCA_TH": "25"}), V("th=29", env={"FIXCA_TH": "29"}), V("th=41", env={"FIXCA_TH": "41"}),
           V("th=57", env={"FIXCA_TH": "57"}), V("ring=14", env={"FIXCA_RING": "14"}), V("ring=18", env={"FIXCA_RING": "18"}), V("ring=20", env={"FIXCA_RING": "20"}),
           V("ring=24", env={"FIXCA_RING": "24"}), V("ring=20 th=25", env={"FIXCA_RING": "20", "FIXCA_TH": "25"}), V("oc=248", env={"FIXCA_OC": "248"}), V("oc=224", env={"FIXCA_OC": "224"}),
           V("32 per launch", per=32), V("128 per launch (b128 build)", "ab_b128", per=128), V("HEAD (again)")],
}
VARIANTS = SETS[sys.argv[2]] if len(sys.argv) > 2 and sys.argv[2] in SETS else SETS["knobs"]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4

libs = {}
def lib_for(path):
    if path not in libs:
        l = ctypes.CDLL(path)
        vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
        l.fixca_batch_device.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ctypes.POINTER(fixca.FixCaParams), cu, vp]
        l.fixca_batch_device.restype = ci
        l.fixca_last_error.restype = ctypes.c_char_p
        libs[path] = l
    return libs[path]

p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
src = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
for i in range(N):
    fixca.noise_fill(src[i], 0xF1CAA000 + i)
vp = ctypes.c_void_p
sp = (vp * N)(*[src[i].data_ptr() for i in range(N)])
dp = (vp * N)(*[dst[i].data_ptr() for i in range(N)])
stream = vp(torch.cuda.current_stream().cuda_stream)

def one_pass(lib, per):
    for k in range(0, N, per):
        m = min(per, N - k)
        rc = lib.fixca_batch_device(ctypes.cast(ctypes.byref(sp, k * 8), ctypes.POINTER(vp)), ctypes.cast(ctypes.byref(dp, k * 8), ctypes.POINTER(vp)),
                                    m, W, H, C, 1, ctypes.byref(p), 0, stream)
        if rc < 0:
            raise RuntimeError(lib.fixca_last_error().decode())

def timed(lib, env, per, reps=4):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        one_pass(lib, per)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            one_pass(lib, per)
        e1.record()
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    us = e0.elapsed_time(e1) * 1e3 / (reps * N)
    return 2.0 * C * W * H / us / 1e3 / 80

res = {v[0]: [] for v in VARIANTS}
ref = None
for r in range(rounds):
    for name, path, env, per in VARIANTS:
        if not os.path.exists(path):
            continue
        res[name].append(timed(lib_for(path), env, per))
    if r == 0:
        # every variant must still compute the same pixels as HEAD (checked once, on the last image)
        one_pass(lib_for(HEAD), 64); torch.cuda.synchronize(); ref = dst[N - 1].clone()
        for name, path, env, per in VARIANTS:
            if os.path.exists(path):
                saved = {k: os.environ.get(k) for k in env}; os.environ.update(env)
                dst[N - 1].zero_(); one_pass(lib_for(path), per); torch.cuda.synchronize()
                for k, v in saved.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
                if not torch.equal(dst[N - 1], ref):
                    print("!! %s: output differs from HEAD" % name)
for name, _, _, _ in VARIANTS:
    v = res[name]
    if v:
        print("%-52s median %5.1f %%   rounds: %s" % (name, statistics.median(v), " ".join("%5.1f" % x for x in v)), flush=True)
