"""GPU box: the data-movement skeleton (tools/skel_lib.hip) and the real kernel on the SAME buffers, for both
ways of holding 32 images (one tensor / one allocation per image).  Answers: does the skeleton's gain from
static big pieces (kbench2: +4.5 points) exist on the buffers where the real kernel lost 5 with them?"""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
os.environ["FIXCA_DEBUG"] = "1"
import torch, fixca
W, H, C, N = 6000, 4000, 4, 32
BIN = os.path.join(ROOT, "tools", "bin")
vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
skel = ctypes.CDLL(os.path.join(BIN, "libskel.so"))
skel.skel_strip.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp)] + [ci] * 9 + [vp]
skel.skel_copy.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, vp]
def real(path):
    l = ctypes.CDLL(path)
    l.fixca_batch_device.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ctypes.POINTER(fixca.FixCaParams), cu, vp]
    l.fixca_last_error.restype = ctypes.c_char_p
    return l
head = real(fixca.LIB_PATH)
bigp = real(os.path.join(BIN, "ab_bigpieces", "libfixca_hip.so")) if os.path.exists(os.path.join(BIN, "ab_bigpieces", "libfixca_hip.so")) else None
p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
stream = vp(torch.cuda.current_stream().cuda_stream)

def run_set(label, srcs, dsts, rounds=3):
    sp = (vp * N)(*[t.data_ptr() for t in srcs]); dp = (vp * N)(*[t.data_ptr() for t in dsts])
    def rl(lib, env):
        def f():
            saved = {k: os.environ.get(k) for k in env}; os.environ.update(env)
            rc = lib.fixca_batch_device(sp, dp, N, W, H, C, 1, ctypes.byref(p), 0, stream)
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            assert rc >= 0, lib.fixca_last_error()
        return f
    variants = [("skeleton dynamic th=33 (batch order)", lambda: skel.skel_strip(sp, dp, N, W, H, 240, 16, 0, 0, 33, 2, stream)),
                ("skeleton dynamic th=33 + XCD remap", lambda: skel.skel_strip(sp, dp, N, W, H, 240, 16, 0, 0, 33, 3, stream)),
                ("skeleton big 5x14% + pool th=32", lambda: skel.skel_strip(sp, dp, N, W, H, 240, 16, 5, 700, 32, 2, stream)),
                ("skeleton big 4x15% + pool th=32", lambda: skel.skel_strip(sp, dp, N, W, H, 240, 16, 4, 600, 32, 2, stream)),
                ("real HEAD", rl(head, {})),
                ("copy_lin nt", lambda: skel.skel_copy(sp, dp, N, W, H, 1, stream))]
    if bigp is not None:
        variants.insert(5, ("real big pieces + pool (d0cd322)", rl(bigp, {"FIXCA_DEBUG_RESIDENT": "-1"})))
    res = {n: [] for n, _ in variants}
    for r in range(rounds):
        for n, f in variants:
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (6 * N)
            res[n].append(8.0 * W * H / us / 1e3 / 80)
    print("== %s" % label)
    for n, _ in variants:
        print("  %-40s median %5.1f %%   %s" % (n, statistics.median(res[n]), " ".join("%5.1f" % x for x in res[n])), flush=True)

for trial in range(2):
    big_s = torch.empty((N, H, W, C), dtype=torch.uint8, device="cuda"); big_d = torch.empty_like(big_s)
    for i in range(N):
        fixca.noise_fill(big_s[i], 0xF1CAB000 + i)
    run_set("one tensor of 32 images (trial %d)" % trial, [big_s[i] for i in range(N)], [big_d[i] for i in range(N)])
    del big_s, big_d; torch.cuda.empty_cache()
    ss = [torch.empty((H, W, C), dtype=torch.uint8, device="cuda") for _ in range(N)]
    dd = [torch.empty((H, W, C), dtype=torch.uint8, device="cuda") for _ in range(N)]
    for i in range(N):
        fixca.noise_fill(ss[i], 0xF1CAB000 + i)
    run_set("one allocation per image (trial %d)" % trial, ss, dd)
    del ss, dd; torch.cuda.empty_cache()
