"""GPU box: A/B of library builds and tuning knobs ON ONE ALLOCATION (the kernel's speed moves by +-2.5 points
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

W, H, C, N = 6000, 4000, 4, 128
BIN = os.path.join(ROOT, "tools", "bin")
HEAD = fixca.LIB_PATH
def V(name, lib=None, env=None, per=64):
    return (name, HEAD if lib is None else os.path.join(BIN, lib, "libfixca_hip.so"), env or {}, per)
SETS = {
 "one": [V("HEAD")],
 "structure": [V("HEAD, 64 per launch"), V("HEAD, 32 per launch", per=32), V("HEAD, 16 per launch", per=16),
               V("HEAD, 128 per launch (ROLL_BATCH_MAX=128 build)", "ab_b128", per=128),
               V("64-byte aligned 272-column window, DPP kept", "ab_al64"), V("nt ring loads", "ab_ldnt"),
               V("persistent, per-XCD tile queues (601aab9)", "ab_persist", {"FIXCA_PERSIST": "1"}),
               V("big pieces + pool (d0cd322)", "ab_bigpieces", {"FIXCA_DEBUG_RESIDENT": "-1"}), V("HEAD, 64 per launch (again)")],
 "knobs": [V("HEAD"), V("th=25", env={"FIXCA_TH": "25"}), V("th=27", env={"FIXCA_TH": "27"}), V("th=29", env={"FIXCA_TH": "29"}), V("th=31", env={"FIXCA_TH": "31"}),
           V("th=35", env={"FIXCA_TH": "35"}), V("th=37", env={"FIXCA_TH": "37"}), V("ring=14", env={"FIXCA_RING": "14"}), V("ring=15", env={"FIXCA_RING": "15"}),
           V("ring=17", env={"FIXCA_RING": "17"}), V("ring=18", env={"FIXCA_RING": "18"}), V("oc=248", env={"FIXCA_OC": "248"}), V("oc=224", env={"FIXCA_OC": "224"}),
           V("lds pad 1500 (8 waves/CU)", env={"FIXCA_EXP_LDS_PAD": "3000"}), V("HEAD (again)")],
 "combo": [V("HEAD"), V("ring=18 th=27", env={"FIXCA_RING": "18", "FIXCA_TH": "27"}), V("ring=18 th=31", env={"FIXCA_RING": "18", "FIXCA_TH": "31"}),
           V("RDEPTH=10 ring=20", "ab_rd10", {"FIXCA_RING": "20"}), V("RDEPTH=10 ring=20 th=27", "ab_rd10", {"FIXCA_RING": "20", "FIXCA_TH": "27"}),
           V("RDEPTH=10 ring=20 th=37", "ab_rd10", {"FIXCA_RING": "20", "FIXCA_TH": "37"}),
           V("RDEPTH=10 ring=22", "ab_rd10", {"FIXCA_RING": "22"}), V("RDEPTH=12 ring=22", "ab_rd12", {"FIXCA_RING": "22"}), V("RDEPTH=12 ring=24", "ab_rd12", {"FIXCA_RING": "24"}),
           V("RDEPTH=14 ring=24", "ab_rd14", {"FIXCA_RING": "24"}), V("RDEPTH=14 ring=26", "ab_rd14", {"FIXCA_RING": "26"}),
           V("RDEPTH=10 ring=20, 128 per launch", "ab_rd10b128", {"FIXCA_RING": "20"}, 128), V("HEAD (again)")],
 "builds": [V("HEAD"), V("RDEPTH=4", "ab_rd4"), V("RDEPTH=8", "ab_rd8"), V("RDEPTH=10", "ab_rd10"), V("RDEPTH=8 ring=18", "ab_rd8", {"FIXCA_RING": "18"}),
            V("RDEPTH=10 ring=20", "ab_rd10", {"FIXCA_RING": "20"}), V("stores default policy", "ab_st0"), V("stores sc0 sc1... aux 3", "ab_st3"), V("loads aux 1 (sc0)", "ab_ldsc1"), V("HEAD (again)")],
}
VARIANTS = SETS[sys.argv[2]] if len(sys.argv) > 2 and sys.argv[2] in SETS else SETS["structure"]
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
    return 8.0 * W * H / us / 1e3 / 80

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
