#!/usr/bin/env python3
"""bench.py — Mpixels/s of the lateral-CA cubic pixel loop on 6000x4000 RGBA8.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--global-batch G] [--per-launch P]

Workload (BASELINE.json): config 3's image and parameters (6000x4000 RGBA8, lateral blue +2 /
red -2, lens (658,1280), cubic) as config 5's batch: G = 512 independent images, image i =
noise(0xF1CA1000 + i) generated on the device, image i owned by rank i mod N.
One step = one pass of the hot path over the rank's images: fixca_batch_device -> the row-rolling
kernel, 128 images per launch.  Nothing is re-used between launches (98 GB of pixels at N = 1), so
no cache level can serve a repeat.  N > 1 (one rank per GPU, torch.distributed.run): rank 0
broadcasts the 80-byte parameter block over RCCL, no pixel crosses GPUs, the total work is fixed
(strong scaling).  Prints ONE JSON line on rank 0.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: the
parent (which never touches the GPU, not even through an import) runs `python -m
torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`
as a child process, relays rank 0's JSON line and exits with the child's code.  Launched under
torch.distributed.run by somebody else (WORLD_SIZE set) it is a rank.

Untimed extras on rank 0 at N = 1 (skipped with --no-extras): `single_image_launch` (one image per
launch, what fix_ca_region does for the plug-in), `configs` (every BASELINE configuration through the
same timing method), the on-box copy yardstick, the host entry (PCIe inclusive), `cold_call_ms`
(a fresh process from start to exit), and `cpu_baseline` (the CPU oracle timed on this host).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))

W, H, C = 6000, 4000, 4
PARAMS = dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
SEED0 = 0xF1CA1000               # image i of the batch = noise(SEED0 + i), SURVEY §8d config 5
METRIC = "Mpixels/s lateral-CA cubic on 6000×4000 RGBA; % of HBM-read roofline"     # BASELINE.json's metric
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# every GPU configuration of BASELINE.json (+ the formats SURVEY §8f row 1 names), timed the same way
CONFIGS = [
    dict(name="config2 linear, centre lens, 6000x4000 RGBA8", w=6000, h=4000, c=4, bpc=1,
         params=dict(blue=2.0, red=-2.0, lens_x=3000, lens_y=2000, interpolation=1)),
    dict(name="config3 cubic, lens (658,1280), 6000x4000 RGBA8", w=6000, h=4000, c=4, bpc=1, params=PARAMS),
    dict(name="none (bit-exact gather), centre lens, 6000x4000 RGBA8", w=6000, h=4000, c=4, bpc=1,
         params=dict(blue=1.5, red=-1.5, lens_x=3000, lens_y=2000, interpolation=0)),
    dict(name="config4 directional cubic, 8192x8192 RGBA8", w=8192, h=8192, c=4, bpc=1,
         params=dict(lens_x=4096, lens_y=4096, x_blue=1.0, x_red=0.0, y_blue=-7.5, y_red=4.0, interpolation=2)),
    dict(name="RGB8 cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=1, params=PARAMS),
    dict(name="RGBA u16 cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=2, params=PARAMS, images=64),
    dict(name="RGBA float cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-4, params=PARAMS, images=32),
    dict(name="RGBA u16 linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=2, params=dict(PARAMS, interpolation=1), images=64),
    dict(name="RGBA u16 none, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=2, params=dict(PARAMS, interpolation=0), images=64),
    dict(name="RGBA double cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-8, params=PARAMS, images=16),
    dict(name="RGBA u32 none, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=4, params=dict(PARAMS, interpolation=0), images=32),
    dict(name="RGBA float none, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-4, params=dict(PARAMS, interpolation=0), images=32),
    dict(name="RGBA double none, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-8, params=dict(PARAMS, interpolation=0), images=16),
    dict(name="RGB u16 none, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=2, params=dict(PARAMS, interpolation=0), images=64),
    dict(name="RGBA float linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-4, params=dict(PARAMS, interpolation=1), images=32),
    dict(name="RGBA double linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=-8, params=dict(PARAMS, interpolation=1), images=16),
    dict(name="RGB float linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=-4, params=dict(PARAMS, interpolation=1), images=32),
    # the weakest kernels belong in the line too (16-bit RGB is what a 16-bit TIFF / PNG photograph is in GIMP)
    dict(name="RGB8 linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=1, params=dict(PARAMS, interpolation=1)),
    dict(name="RGB8 none, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=1, params=dict(PARAMS, interpolation=0)),
    dict(name="RGB u16 cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=2, params=PARAMS, images=64),
    dict(name="RGB u16 linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=2, params=dict(PARAMS, interpolation=1), images=64),
    dict(name="RGB float cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=-4, params=PARAMS, images=32),
    dict(name="RGB u32 linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=4, params=dict(PARAMS, interpolation=1), images=32),
    dict(name="RGBA u32 cubic, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=4, params=PARAMS, images=32),
    dict(name="RGBA u32 linear, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=4, params=dict(PARAMS, interpolation=1), images=32),
    # widths that are not multiples of 4 (three of four arbitrary crops): the main kernel since round 4, before that the
    # register-window kernels at 19-25 %
    dict(name="RGBA8 cubic, 5999 wide, lens (658,1280), 5999x4000", w=5999, h=4000, c=4, bpc=1, params=PARAMS),
    dict(name="RGB8 cubic, 5999 wide, lens (658,1280), 5999x4000", w=5999, h=4000, c=3, bpc=1, params=PARAMS),
    dict(name="RGB u16 cubic, 5999 wide, lens (658,1280), 5999x4000", w=5999, h=4000, c=3, bpc=2, params=PARAMS, images=64),
    # stronger corrections (outputs clamped to the row's edge pixels lie out of a lane window's reach): the main kernel's
    # edge-pixel instantiation since round 4, before that k_scan (44 %) / k_interp (22 %)
    dict(name="RGBA8 cubic, blue 5 red -5, lens (658,1280), 6000x4000", w=6000, h=4000, c=4, bpc=1, params=dict(PARAMS, blue=5.0, red=-5.0)),
    dict(name="RGB8 cubic, blue 5 red -5, lens (658,1280), 6000x4000", w=6000, h=4000, c=3, bpc=1, params=dict(PARAMS, blue=5.0, red=-5.0)),
]


def oracle_image0(params_dict, threads=None):
    """The CPU oracle's result for image 0 of the batch (noise(SEED0)): the checker of the bench line's parity.
    The row-band OpenMP build computes the same bytes as the single thread (tests/test_oracle_golden.py)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pyoracle
    from fixca import synth
    img = synth.noise_image(SEED0, W, H, C)
    dst = np.empty_like(img)
    nthr = threads or max(1, min(len(os.sched_getaffinity(0)), 64))
    pyoracle.region(img, pyoracle.make_params(**params_dict), dst=dst, threads=nthr)
    return dst


def oracle_diff(np, got, ref):
    """{max_abs_diff, differing} of a GPU result against the oracle's (uint8 images)"""
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    return {"image": 0, "max_abs_diff": int(d.max()), "differing": int(np.count_nonzero(d)), "samples": int(d.size),
            "green_alpha_equal": bool(np.array_equal(got[..., 1::2], ref[..., 1::2])),
            "bar": "<= 1 LSB on red/blue (north_star), green/alpha exact"}


def cpu_baseline(params_dict, preview=None):
    """The CPU oracle (the reference's algorithm restated in C, oracle/) timed on this
    host: 1 thread like the reference, plus an all-cores row-band run as context.
    preview = (GPU window, its parameters): the same preview call on the oracle, timed and compared."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pyoracle
    from fixca import synth
    img = synth.noise_image(SEED0, W, H, C)
    dst = np.empty_like(img)
    p = pyoracle.make_params(**params_dict)
    best = float("inf")
    for _ in range(3):
        t = time.perf_counter()
        pyoracle.region(img, p, dst=dst)
        best = min(best, time.perf_counter() - t)
    ncpu = len(os.sched_getaffinity(0))
    nthr = max(1, min(ncpu, 64))
    best_mt = float("inf")
    for _ in range(2):
        t = time.perf_counter()
        pyoracle.region(img, p, dst=dst, threads=nthr)
        best_mt = min(best_mt, time.perf_counter() - t)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {
        "value": round(W * H / best / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": "3 passes over image 0 of the batch (6000x4000 RGBA8 noise, 24 Mpx each), best pass, gcc -O2 no -march",
        "all_cores": {"value": round(W * H / best_mt / 1e6, 1), "threads": nthr},
        "host": {"cpu": model, "nproc": ncpu},
    }
    if preview is not None:
        got, pdict = preview
        x, y, w, h = PREVIEW_WINDOW
        pp = pyoracle.make_params(**pdict)
        t = time.perf_counter()
        full = pyoracle.region(img, pp, y1=y, y2=y + h, show_progress=0, dst=dst)     # what preview_update asks of fix_ca_region (fix-ca.c:656)
        want = pyoracle.preview(full, 1, x, y, w, h)
        ms = (time.perf_counter() - t) * 1e3
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        out["preview_call"] = {"ms": round(ms, 1), "cores": 1, "gpu_window_max_abs_diff": int(d.max()),
                               "green_alpha_equal": bool(np.array_equal(got[..., 1], want[..., 1]) and np.array_equal(got[..., 3], want[..., 3]))}
    return out, dst


PREVIEW_WINDOW = (2744, 1744, 512, 512)     # x, y, width, height of the preview window in the 6000x4000 image


def preview_calls(np, fixca, himg):
    """preview_update()'s pixel work (fix-ca.c:647-671) through fixca_preview_host, as a slider drag makes it: every
    call has a new parameter block (new tables), uploads the rows the window's rows reach, corrects full-width rows
    [y, y + height) with show_progress FALSE (centre lines), crops the window and downloads it.  Host wall time."""
    x, y, w, h = PREVIEW_WINDOW
    times, out, last = [], None, None
    for k in range(6):
        last = dict(PARAMS, blue=2.0 + 0.25 * k, red=-2.0 - 0.125 * k)
        p = fixca.make_params(**last)
        t = time.perf_counter()
        out = fixca.preview(himg, p, 1, x, y, w, h)
        times.append((time.perf_counter() - t) * 1e3)
    times = sorted(times[1:])                                   # the first call allocates
    return {"ms": round(times[len(times) // 2], 3), "window": list(PREVIEW_WINDOW), "calls": 5,
            "what": "fixca_preview_host on the host image, a new parameter block per call (slider drag): upload of the rows in reach, "
                    "rows y..y+height corrected at full width, centre lines, crop, download; median host wall time"}, out, last


def sample_bytes(bpc):
    return abs(bpc)


KID_ROLL = 6                     # fixca_plan_describe's id of the row-rolling kernel (include/fixca_hip.h)
KID_WCOL = 7                     # ... of the fp64 rolling-window kernel of the wide sample types: up to 128 images per launch
KID_MOVE = 8                     # ... of the NONE byte mover of the wide sample types: likewise
KID_LIN = 9                      # ... of the LDS-ring LINEAR kernel of the wide RGBA formats: likewise
KID_CUB = 10                     # ... of the LDS-ring CUBIC kernel of the wide RGBA formats: likewise


def kernel_name(kid, plan_text, width=0):
    """the device function a plan runs, from the planner's own description (width: the image's, for the row-rolling
    kernel's third template argument: 1 = its instantiation for widths that are not multiples of 4)"""
    words = plan_text.split()
    if kid == KID_ROLL:
        taps = {"none": 1, "linear": 2, "cubic": 4}[words[1]]
        ring = int([w for w in words if w.startswith("ring=")][0][5:])
        rgb = "bpp=3" in words
        return "k_roll%s<%d,%s,%d>" % ("_rgb" if rgb else "", taps, "true" if ring & (ring - 1) == 0 else "false", 1 if width % 4 else 0)
    return {"exact-fp64": "k_exact/k_wide", "fast-none": "k_none", "fast-interp": "k_interp", "scan": "k_scan", "scanw": "k_scanw",
            "wide-roll-fp64": "k_wcol", "move-none": "k_move", "ring-linear-fp64": "k_lin", "ring-cubic-fp64": "k_cub"}.get(words[0], words[0])


def make_images(torch, fixca, n, h, w, c, bpc, seed0, dev, into=None):
    """n device images of SURVEY noise; float formats get values in [0,1) so that clip01 is not the only path.
    into: a uint8 device tensor to carve the images from (they are then a view of it) instead of a new allocation"""
    sb = sample_bytes(bpc)
    nbytes = n * h * w * c * sb
    raw = (torch.empty(nbytes, dtype=torch.uint8, device=dev) if into is None else into[:nbytes]).view(n, h, w, c * sb)
    for i in range(n):
        fixca.noise_fill(raw[i], seed0 + i)
        if bpc == -4:          # image by image, in place: no second copy of the pool
            f = raw[i].view(torch.int32).to(torch.float32).abs_().mul_(1.0 / 2147483648.0)
            raw[i].copy_(f.view(torch.uint8).reshape(h, w, c * sb))
        elif bpc == -8:
            f = raw[i].view(torch.int64).to(torch.float64).abs_().mul_(1.0 / 9223372036854775808.0)
            raw[i].copy_(f.view(torch.uint8).reshape(h, w, c * sb))
    return raw


def time_batch(torch, fixca, srcs, dsts, w, h, bytes_pp, bpc, params, per_launch, reps):
    """mean device time of one batch launch (HIP events on the launch stream), launches of per_launch images"""
    n = len(srcs)
    groups = [(srcs[i:i + per_launch], dsts[i:i + per_launch]) for i in range(0, n, per_launch)]

    def one_pass():
        for s, d in groups:
            fixca.fix_ca_batch(s, d, w, h, bytes_pp, bpc, params)
    # untimed passes first: the configuration's images were just allocated and filled (short kernels, pauses), and the
    # clocks take a few milliseconds of this kernel's load to settle: one pass read config 3 three points under the
    # headline of the same run (68.8-69.8 against 71.8-72.8 %, same kernel, same parameters)
    for _ in range(max(2, 512 // n)):
        one_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one_pass()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)          # us per image




def run_configs(torch, fixca, dev, per_launch, scratch=None):
    """Every configuration through the same timing.  scratch: (source pool, result pool), device memory allocated at the
    START of the process from which every configuration carves its images: the timed job's own two image pools, so that a
    configuration and the headline are measured on the same physical pages.  Memory that torch has freed and allocated again
    reads 3 points slower with this kernel than memory allocated for the first time (tools/alloc_modes2.py,
    profiles/r04_alloc_modes2.txt: 68.7-68.8 against 70.1-71.8 % on the same virtual addresses), and a bench that frees
    its 98-GB pool and then allocates pool after pool measured every configuration in that state (config 3: 69.5 % next
    to the headline's 72.3 % in the same run).  A PDB call is a fresh process: its images are first allocations."""
    out = []
    for cfg in CONFIGS:
        w, h, c, bpc = cfg["w"], cfg["h"], cfg["c"], cfg["bpc"]
        bytes_pp = c * sample_bytes(bpc)
        n = cfg.get("images", min(per_launch, 128 if cfg["w"] * cfg["h"] <= 6000 * 4000 else 32))     # 8-bit configs: as many per launch as the headline
        p = fixca.make_params(**cfg["params"])
        entry = {"name": cfg["name"], "width": w, "height": h, "bytes_per_pixel": bytes_pp, "bpc": bpc}
        try:
            kid, desc = fixca.plan(w, h, bytes_pp, bpc, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)
            pool_bytes = n * h * w * bytes_pp
            carve = scratch is not None and pool_bytes <= min(scratch[0].numel(), scratch[1].numel())
            raw = make_images(torch, fixca, n, h, w, c, bpc, 0xF1CA2000, dev, into=scratch[0] if carve else None)
            srcs = [raw[i] for i in range(n)]
            dstt = scratch[1][:pool_bytes].view(raw.shape) if carve else torch.empty_like(raw)
            entry["fresh_memory"] = bool(carve)
            dsts = [dstt[i] for i in range(n)]
            reps = max(4, 1024 // n)             # >= 1024 image passes: short runs read a few percent slow (clock ramp)
            us = time_batch(torch, fixca, srcs, dsts, w, h, bytes_pp, bpc, p, per_launch, reps)
            algo = 2.0 * w * h * bytes_pp
            # sanity next to the number: the fast path against the fp64 kernel on image 0 (integer formats: <= 1 LSB)
            exact = torch.empty_like(srcs[0])
            fixca.fix_ca_region(srcs[0], exact, w, h, bytes_pp, bpc, p, 0, w, 0, h, True, fixca.KERNEL_EXACT)
            torch.cuda.synchronize()
            if bpc > 0:
                dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[bpc]
                a, b = dsts[0].view(dt).to(torch.int64), exact.view(dt).to(torch.int64)
                if bpc == 2:
                    a, b = a & 0xFFFF, b & 0xFFFF
                if bpc == 4:
                    a, b = a & 0xFFFFFFFF, b & 0xFFFFFFFF
                md = int((a - b).abs().max())
            else:
                dt = torch.float32 if bpc == -4 else torch.float64
                md = float((dsts[0].view(dt).to(torch.float64) - exact.view(dt).to(torch.float64)).abs().max())
            # the row-rolling kernel (plan id 6) takes up to per_launch images in one launch, the kernels of the wide sample types (7 - 10) up to 128; every other kernel is one launch per image
            ipl = min(n, per_launch, 128) if kid in (KID_ROLL, KID_WCOL, KID_MOVE, KID_LIN, KID_CUB) else 1
            entry.update({"kernel_plan": desc, "kernel": kernel_name(kid, desc, w), "images_per_launch": ipl, "launches_timed": reps * ((n + per_launch - 1) // per_launch if ipl > 1 else n),
                          "us_per_image": round(us, 2), "GBs": round(algo / us / 1e3, 1),
                          "frac": round(algo / us / 1e3 / HBM_PEAK_GBS, 4), "mpix_s": round(w * h / us, 1),
                          "max_abs_diff_vs_fp64_kernel": md})
            del raw, dstt, srcs, dsts, exact
        except Exception as e:                      # a configuration the library refuses is reported, not hidden
            entry["error"] = str(e)[:200]
        torch.cuda.empty_cache()
        out.append(entry)
    return out


def cold_call(torch, hsrc):
    """wall time of one fresh process doing what a PDB call does: start, HIP init, read the image,
    fixca_region_host (upload + kernel + download), write the result, exit; with the phases the process
    stamps itself (FIXCA_PHASES: ms since this process read the clock just before spawning it)"""
    import numpy as np
    cli = os.path.join(ROOT, "gimp-fix-ca_amd", "fixca_cli")
    if not os.path.exists(cli):
        return None
    tmp = os.environ.get("TMPDIR", "/tmp")
    src, dst = os.path.join(tmp, "fixca_cold_in.pam"), os.path.join(tmp, "fixca_cold_out.pam")
    with open(src, "wb") as f:
        f.write(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" % (W, H))
        f.write(np.ascontiguousarray(hsrc).tobytes())
    base = [cli, "--lens-as-float", src, dst, str(PARAMS["blue"]), str(PARAMS["red"]), str(PARAMS["lens_x"]),
            str(PARAMS["lens_y"]), str(PARAMS["interpolation"])]

    def run(extra, reps):
        best, phases = None, None
        for _ in range(reps):                       # always a fresh child process; the later runs have the page cache and the code objects' files warm
            time.sleep(0.5)                         # the kernel driver is still tearing the previous child down: its successor's HIP init then takes 170-190 ms instead of 55
            env = dict(os.environ, FIXCA_PHASES="1", FIXCA_T0_NS=str(time.clock_gettime_ns(time.CLOCK_MONOTONIC)))
            t = time.perf_counter()
            r = subprocess.run(base[:1] + extra + base[1:], capture_output=True, env=env)
            dt = (time.perf_counter() - t) * 1e3
            if r.returncode != 0:
                return None, {"error": (r.stderr or r.stdout).decode("utf-8", "replace")[:200]}
            if best is None or dt < best:
                best = dt
                phases = {}
                for line in r.stderr.decode("utf-8", "replace").splitlines():
                    w = line.split()
                    if len(w) == 3 and w[0] == "fixca-phase":
                        phases[w[1]] = round(float(w[2]), 1)
                phases["parent:child_reaped"] = round(dt, 1)
        return best, phases

    best, phases = run([], 3)
    if best is None:
        return phases
    out = np.fromfile(dst, dtype=np.uint8)[-W * H * 4:].reshape(H, W, 4)
    pinned_best, pinned_phases = run(["--pinned"], 2)
    fast_best, fast_phases = run(["--fast-exit"], 3)           # _exit(0) once the result file is closed: no runtime teardown
    fast_out = np.fromfile(dst, dtype=np.uint8)[-W * H * 4:].reshape(H, W, 4) if fast_best is not None else None
    for p_ in (src, dst):
        try:
            os.remove(p_)
        except OSError:
            pass
    return {"ms": round(best, 1), "result_ready_ms": phases.get("cli:image_written"),
            "what": "fixca_cli (C host layer) on a 6000x4000 RGBA PAM in tmpfs/page cache: process start, HIP init, "
            "file read, fixca_region_host, file write, exit; a fresh process each time, best of 3; phases = ms since the parent spawned it; "
            "ms = until the parent has reaped the child (the HIP runtime's teardown is the last ~90 ms), result_ready_ms = until the result file is written; "
            "ms_with_fast_exit = the same with fixca_cli --fast-exit (_exit(0) after the result file is closed)",
            "phases": phases, "with_page_locked_images_ms": None if pinned_best is None else round(pinned_best, 1),
            "with_page_locked_images_phases": pinned_phases,
            "ms_with_fast_exit": None if fast_best is None else round(fast_best, 1), "fast_exit_phases": fast_phases,
            "fast_exit_output_equal": None if fast_out is None else bool(np.array_equal(fast_out, out))}, out


def compact_line(full):
    """The ONE line rank 0 prints.  The driver's record keeps the fixed keys (with `roofline` whole, scalars only) and the
    last few KB of output, so the line stays under ~6 KB: no prose, no plan strings per configuration, no second set of
    cold-call phases (those are in gpurun_out/bench_full.json), and the figures a reader needs first sit inside `roofline`."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "roofline", "ok", "checks") if k in full}
    roof = dict(full["roofline"])
    one = full.get("single_image_launch")
    if one:
        roof["single_image_us"] = one["us_per_image"]
        roof["single_image_frac"] = one["frac"]
        line["single_image_launch"] = {k: one[k] for k in ("us_per_image", "frac", "launches", "warmup_launches", "rotated_pairs") if k in one}
        if "parity_vs_oracle" in one:
            line["single_image_launch"]["oracle_max_abs_diff"] = one["parity_vs_oracle"]["max_abs_diff"]
    if "copy_kernel_GBs" in full:
        roof["copy_frac"] = round(full["copy_kernel_GBs"] / HBM_PEAK_GBS, 4)
    po = full.get("parity_vs_oracle")
    if po:
        roof["oracle_max_abs_diff"] = po["max_abs_diff"]
        roof["oracle_differing"] = po["differing"]
        roof["oracle_samples"] = po["samples"]
        roof["oracle_green_alpha_equal"] = po["green_alpha_equal"]
    pf = full.get("parity_vs_fp64_kernel")
    if pf:
        roof["fp64_kernel_max_abs_diff"] = pf["max_abs_diff"]
    if "second_allocation" in full:
        roof["second_allocation_frac"] = full["second_allocation"]["frac"]
    if "launch_groups" in full and full["launch_groups"]["frac"]:
        roof["launch_frac_min"] = min(full["launch_groups"]["frac"])
        roof["launch_frac_max"] = max(full["launch_groups"]["frac"])
    cfgs = [c for c in full.get("configs", []) if "frac" in c]
    if cfgs:
        rgb8 = [c for c in cfgs if c["name"].startswith("RGB8 cubic")]
        if rgb8:
            roof["rgb8_frac"] = rgb8[0]["frac"]
        worst = min(cfgs, key=lambda c: c["frac"])
        roof["min_config_frac"] = worst["frac"]
        roof["min_config"] = worst["name"].split(",")[0]
        line["configs_columns"] = ["name", "us_per_image", "frac_of_8TBs", "kernel", "images_per_launch", "max_abs_diff_vs_fp64_kernel"]
        line["configs"] = [[c["name"].replace(", lens (658,1280), 6000x4000", "").replace(", lens (658,1280), 5999x4000", "").replace(", 6000x4000", ""), c["us_per_image"], c["frac"],
                            c["kernel"], c["images_per_launch"], c["max_abs_diff_vs_fp64_kernel"]] for c in cfgs]
        line["configs"] += [[c["name"], "error: " + c["error"][:80]] for c in full["configs"] if "error" in c]
    line["roofline"] = roof
    if "cpu_baseline" in full:
        line["cpu_baseline"] = full["cpu_baseline"]
    for k in ("host_entry_pcie_inclusive_mpix_s", "host_entry_pinned_pcie_inclusive_mpix_s"):
        if k in full:
            line[k] = full[k]
    pv = full.get("preview_call")
    if pv:
        line["preview_call"] = {k: pv[k] for k in ("ms", "calls", "cpu_oracle_ms", "vs_oracle") if k in pv}
    cc = full.get("cold_call_ms")
    if cc:
        line["cold_call_ms"] = {k: cc[k] for k in ("ms", "result_ready_ms", "ms_with_fast_exit", "phases", "with_page_locked_images_ms", "error") if k in cc}
    line["full_record"] = "gpurun_out/bench_full.json"
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=512, help="images of the whole job (BASELINE config 5: 512)")
    ap.add_argument("--per-launch", type=int, default=128, help="images per kernel launch (the library takes up to 128)")
    ap.add_argument("--batch", type=int, default=0, help="shorthand: global batch = this many images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras that launch other kernels or single images; tools/profile.sh uses "
                         "it so that rocprof's per-kernel average is over the timed launches only")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) with no rendezvous in the environment: start the N ranks.
    The parent has imported nothing that initialises HIP (no torch, no fixca) and starts
    torch.distributed.run as a CHILD process (never exec: SURVEY/driver rule for this pool), relays
    rank 0's JSON line on stdout and everything else on stderr, and returns the child's exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line_out = None
    for line in child.stdout:
        t = line.strip()
        if t.startswith("{") and '"metric"' in t:
            line_out = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = child.wait()
    if line_out:
        print(line_out, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    # dmabuf IPC: RCCL between processes needs it on this pool.  Set before anything touches the GPU, also when a driver
    # started the ranks under torch.distributed.run itself (launch_ranks sets it for the ranks it starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a driver that asks for N GPUs must not get a number for another N with exit code 0
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: run `python bench.py --gpus N` (it starts its own ranks) or "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import fixca                      # raises if libfixca_hip.so is missing: no fallback
    from fixca import dist as fdist
    from fixca import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    # FIXCA_BENCH_REHEARSE=1: several ranks share the one GPU of a test box and talk over gloo, to
    # rehearse the N > 1 control flow (tests/test_gpu_parity.py::test_bench_two_ranks_rehearsal);
    # the real run is one rank per GPU over RCCL
    rehearse = os.environ.get("FIXCA_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    if not rehearse and world > ndev:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (FIXCA_BENCH_REHEARSE=1 shares one GPU between ranks)" % (world, ndev))
    local = local % ndev if rehearse else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearse else dev          # where the few bytes of the collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            if dist.get_backend() != "nccl":
                raise SystemExit("bench.py: the process group came up as %r, not nccl (RCCL)" % dist.get_backend())
    transport = "none"
    if world > 1:
        if rehearse:
            transport = "gloo, %d ranks sharing GPU 0 (FIXCA_BENCH_REHEARSE: control-flow rehearsal, not a measurement)" % world
        else:
            transport = "rccl %s (torch.distributed nccl backend), world %d, one rank per GPU" % (
                ".".join(str(v) for v in torch.cuda.nccl.version()), world)

    # rank 0 owns the parameters; everyone else receives them over RCCL
    params = fdist.broadcast_params(fixca.make_params(**PARAMS) if rank == 0 else None, src=0, device=cdev)
    kid, plan_text = fixca.plan(W, H, C, 1, params, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)   # what fix_ca_batch will run

    G = args.batch * world if args.batch else args.global_batch
    if G < world:
        raise SystemExit("bench.py: a batch of %d images cannot give every one of %d ranks an image" % (G, world))
    own = fdist.shard_indices(G, rank, world)               # image i -> rank i mod N
    n_own = len(own)
    P = max(1, min(args.per_launch, 128))
    src_all = torch.empty((n_own, H, W, C), dtype=torch.uint8, device=dev)
    dst_all = torch.empty_like(src_all)
    for j, i in enumerate(own):
        fixca.noise_fill(src_all[j], SEED0 + i)
    srcs = [src_all[j] for j in range(n_own)]
    dsts = [dst_all[j] for j in range(n_own)]
    # the C-ABI calls of one step: the rank's share in launches of P images
    vp = ctypes.c_void_p
    groups = []
    for k in range(0, n_own, P):
        m = min(P, n_own - k)
        groups.append(((vp * m)(*[t.data_ptr() for t in srcs[k:k + m]]), (vp * m)(*[t.data_ptr() for t in dsts[k:k + m]]), m))
    launches_per_step = len(groups)
    stream = vp(torch.cuda.current_stream().cuda_stream)

    def step():
        for s, d, m in groups:
            rc = fixca.lib.fixca_batch_device(s, d, m, W, H, C, 1, ctypes.byref(params), fixca.KERNEL_AUTO, stream)
            if rc < 0:
                raise fixca.FixCaError(rc, fixca.lib.fixca_last_error().decode())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # torch's current stream == the stream the kernels are launched on
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    if world > 1:
        t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = t.tolist()

    # ---- not timed: sanity of what was just computed (every rank), yardsticks, CPU baseline (rank 0) ----
    ok = True
    extra = {}
    pv_img = pv_params = None
    checks = {}
    exact = torch.empty_like(srcs[0])
    worst, differing = 0, 0
    checked = sorted(set([0, min(31, n_own - 1), min(32, n_own - 1), n_own - 1]))
    seeds_ok = True
    for j in checked:
        # the image this rank holds at slot j IS image own[j] of the batch: its first bytes are noise(SEED0 + own[j])'s
        seeds_ok = seeds_ok and bool(np.array_equal(srcs[j].reshape(-1)[:64].cpu().numpy(), synth.noise_bytes(SEED0 + own[j], 64)))
        fixca.fix_ca_region(srcs[j], exact, W, H, C, 1, params, 0, W, 0, H, True, fixca.KERNEL_EXACT)
        torch.cuda.synchronize()
        diff = (dsts[j].to(torch.int16) - exact.to(torch.int16)).abs()
        worst = max(worst, int(diff.max()))
        differing = max(differing, int((diff != 0).sum()))
        ok = ok and torch.equal(dsts[j][..., 1], srcs[j][..., 1]) and torch.equal(dsts[j][..., 3], srcs[j][..., 3])
    checks["green_alpha_copied_exactly"] = bool(ok)
    checks["within_1_lsb_of_fp64_kernel"] = worst <= 1
    checks["held_images_are_the_owned_ones"] = seeds_ok
    ok = ok and worst <= 1 and seeds_ok
    if rehearse and os.environ.get("FIXCA_BENCH_REHEARSE_FAIL_RANK") == str(rank):
        ok = False                    # rehearsal only: what the job does when one rank's parity check fails (tests)
        checks["rehearsal_injected_failure"] = True
    del exact, diff
    # the oracle's verdict on the timed batch itself: image 0 (rank 0 owns it) against oracle/ on the host,
    # taken before any extra touches dsts
    single0 = None
    if rank == 0 and world == 1 and not args.no_extras:
        # (1) one image per launch: what fix_ca_region does for the plug-in (64 rotated pairs = 12 GB, so no launch finds
        # its image in the 256-MB Infinity Cache; 32 untimed + 1000 timed launches).  Timed HERE, while the device is
        # still busy from the timed steps: the oracle check below keeps the host busy and the GPU idle for half a second
        batch0 = dsts[0].cpu().numpy()                        # image 0 as the timed batch left it
        nrot = min(64, n_own)                                 # 64 pairs: the physical pages decide between modes 3 points apart, 8 pairs are one draw
        one_ms = fixca.time_launches(srcs[:nrot], dsts[:nrot], 1000, W, H, C, 1, params)
        kid1, plan1 = fixca.plan(W, H, C, 1, params)
        extra["single_image_launch"] = {"us_per_image": round(one_ms * 1e3, 2), "launches": 1000, "warmup_launches": 32, "rotated_pairs": nrot,
                                        "frac": round(2.0 * W * H * C / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "kernel_plan": plan1}
        single0 = dsts[0].cpu().numpy()
    if rank == 0 and not args.no_cpu_baseline:
        ref0 = oracle_image0(PARAMS)
        if single0 is None:
            batch0 = dsts[0].cpu().numpy()
        extra["parity_vs_oracle"] = oracle_diff(np, batch0, ref0)
        checks["within_1_lsb_of_cpu_oracle"] = extra["parity_vs_oracle"]["max_abs_diff"] <= 1 and extra["parity_vs_oracle"]["green_alpha_equal"]
        ok = ok and checks["within_1_lsb_of_cpu_oracle"]
    owned = [own]
    if world > 1:
        owned = [None] * world
        dist.all_gather_object(owned, own)
    ok = fdist.all_ranks_ok(ok, device=cdev)
    if rank == 0:
        extra["parity_vs_fp64_kernel"] = {"images_checked": [own[j] for j in checked], "max_abs_diff": worst,
                                          "max_differing_samples_per_image": differing, "samples_per_image": W * H * C}
        if world == 1 and not args.no_extras:
            if "parity_vs_oracle" in extra:              # one image per launch against the oracle too
                extra["single_image_launch"]["parity_vs_oracle"] = oracle_diff(np, single0, ref0)
                ok = ok and extra["single_image_launch"]["parity_vs_oracle"]["max_abs_diff"] <= 1
            batch0 = single0
            # (2) the on-box copy yardstick over the same buffers (it overwrites dsts)
            copy_ms = fixca.time_copy(srcs[:64], dsts[:64], 320, W * H * C)
            extra["copy_kernel_GBs"] = round(2 * W * H * C / (copy_ms * 1e-3) / 1e9, 1)
            # (3) the host entry, PCIe inclusive (never the bench value)
            hsrc = srcs[0].cpu().numpy()
            hdst = np.empty_like(hsrc)
            fixca.fix_ca_region(hsrc, hdst, W, H, C, 1, params, 0, W, 0, H, True)     # warm (allocs, context)
            t = time.perf_counter()
            fixca.fix_ca_region(hsrc, hdst, W, H, C, 1, params, 0, W, 0, H, True)
            extra["host_entry_pcie_inclusive_mpix_s"] = round(W * H / (time.perf_counter() - t) / 1e6, 1)
            psrc, pdst = fixca.pinned_empty(hsrc.shape), fixca.pinned_empty(hsrc.shape)
            psrc[...] = hsrc
            fixca.fix_ca_region(psrc, pdst, W, H, C, 1, params, 0, W, 0, H, True)
            t = time.perf_counter()
            fixca.fix_ca_region(psrc, pdst, W, H, C, 1, params, 0, W, 0, H, True)
            extra["host_entry_pinned_pcie_inclusive_mpix_s"] = round(W * H / (time.perf_counter() - t) / 1e6, 1)
            checks["host_entry_pinned_equals_pageable"] = bool(np.array_equal(pdst, hdst))
            d0 = np.abs(hdst.astype(np.int16) - batch0.astype(np.int16))
            checks["host_entry_vs_batch_entry_max_abs_diff"] = int(d0.max())      # different tilings of the same arithmetic
            ok = ok and checks["host_entry_pinned_equals_pageable"] and int(d0.max()) <= 1
            del d0, batch0
            fixca.pinned_free(psrc)
            fixca.pinned_free(pdst)
            # (3b) the second caller: preview_update()'s calls while a slider is dragged
            pv, pv_img, pv_params = preview_calls(np, fixca, hsrc.reshape(H, W, C))
            extra["preview_call"] = pv
            # (4) a PDB call is a fresh process: what that costs from start to exit
            cc = cold_call(torch, hsrc)
            if isinstance(cc, tuple):
                extra["cold_call_ms"] = cc[0]
                checks["cold_call_output_equals_host_entry"] = bool(np.array_equal(cc[1], hdst))
                ok = ok and checks["cold_call_output_equals_host_entry"]
            elif cc:
                extra["cold_call_ms"] = cc
            del hsrc, hdst
            # (4b) the timed job again, launch by launch.  A pool is not uniform: a launch whose sources and results lie in the
            # same region of device memory runs up to 4 points slower than one whose sources and results lie in different
            # regions (profiles/NOTES.md, round 4 §12), so the groups of one job differ and `value` is their mix
            gfr = []
            for _ in range(2):                                # the device's clocks drop during the host-side legs above: two untimed steps first
                step()
            for s_, d_, m in groups:
                def one():
                    rc = fixca.lib.fixca_batch_device(s_, d_, m, W, H, C, 1, ctypes.byref(params), fixca.KERNEL_AUTO, stream)
                    if rc < 0:
                        raise fixca.FixCaError(rc, fixca.lib.fixca_last_error().decode())
                one()
                torch.cuda.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(4):
                    one()
                g1.record()
                torch.cuda.synchronize()
                gfr.append(round(2.0 * C * W * H * m / (g0.elapsed_time(g1) / 4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            extra["launch_groups"] = {"frac": gfr, "what": "the job's launches one by one (4 timed launches each, not the bench value): where a launch's sources and results lie in device memory decides between two modes"}
            # (5) the same job on a second allocation of the image pool.  Which physical pages a buffer lands on decides
            # between two modes of this kernel about 3 points apart (profiles/NOTES.md); `value` is this run's allocation as
            # it came, the second one is context for whoever reads a 69 or a 72 in the line.
            free_b, _ = torch.cuda.mem_get_info()
            if free_b > 2.2 * src_all.numel():
                src2 = torch.empty_like(src_all)
                dst2 = torch.empty_like(dst_all)
                for j, i in enumerate(own):
                    fixca.noise_fill(src2[j], SEED0 + i)
                groups2 = []
                for k in range(0, n_own, P):
                    m = min(P, n_own - k)
                    groups2.append(((vp * m)(*[src2[j].data_ptr() for j in range(k, k + m)]), (vp * m)(*[dst2[j].data_ptr() for j in range(k, k + m)]), m))

                def step2():
                    for s2, d2, m in groups2:
                        rc = fixca.lib.fixca_batch_device(s2, d2, m, W, H, C, 1, ctypes.byref(params), fixca.KERNEL_AUTO, stream)
                        if rc < 0:
                            raise fixca.FixCaError(rc, fixca.lib.fixca_last_error().decode())
                for _ in range(2):
                    step2()
                torch.cuda.synchronize()
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(6):
                    step2()
                f1.record()
                torch.cuda.synchronize()
                ms2 = f0.elapsed_time(f1) / (6 * len(groups2))
                extra["second_allocation"] = {"frac": round(2.0 * C * W * H * (n_own / len(groups2)) / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "launches_timed": 6 * len(groups2),
                                              "what": "the timed job again on a freshly allocated image pool (not the bench value): the allocation decides between two modes"}
                if n_own > 64:                            # the single-image leg and the copy yardstick overwrote the first 64 results; the last one is intact
                    checks["second_allocation_last_image_equal"] = bool(torch.equal(dst2[n_own - 1], dsts[n_own - 1]))
                    ok = ok and checks["second_allocation_last_image_equal"]
                del src2, dst2, groups2
                torch.cuda.empty_cache()
        kernel_ms = ev_ms / (args.steps * launches_per_step)
        imgs_per_launch = n_own / launches_per_step
        algo_bytes = 2 * C * W * H * imgs_per_launch           # every source byte read once, every result byte written once
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            # recorded per launch of the profiled run; scaled if this run batches differently
            traffic = round(t.get("bytes_per_launch") * imgs_per_launch / max(1, t.get("images_per_launch", 1)))
        out = {
            "metric": METRIC,
            "value": round(args.steps * G * W * H / wall / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] at this N: %d images 6000x4000 RGBA8 (u8 samples, image i = noise(0xF1CA1000+i)), "
                                   "configs[2]'s parameters: lateral blue +2 / red -2, lens (658,1280), cubic" % G,
                       "global_batch": G, "images_per_gpu_per_step": n_own, "images_per_launch": imgs_per_launch,
                       "launches_per_step": launches_per_step,
                       "parallelism": "image i -> rank i mod %d, params broadcast from rank 0, no pixel traffic" % world,
                       "images_by_rank": [o if len(o) <= 16 else o[:4] + ["..."] + o[-2:] for o in owned],
                       "transport": transport,
                       "kernel_plan": plan_text},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "recorded (profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, tools/profile.sh), not measured in this run" if traffic else None,
                         "kernel": kernel_name(kid, plan_text, W), "kernel_us": round(kernel_ms * 1e3, 2),
                         "launches_timed": args.steps * launches_per_step,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "images_per_launch": imgs_per_launch, "us_per_image": round(kernel_ms * 1e3 / imgs_per_launch, 2)},
            "ok": bool(ok), "checks": checks,
        }
        out.update(extra)
        if world == 1 and not args.no_extras:
            # the configurations' images are carved from the timed job's own pools (run_configs says why); nothing below
            # needs the job's images any more
            del srcs, dsts, groups
            out["configs"] = run_configs(torch, fixca, dev, P, (src_all.view(-1), dst_all.view(-1)))
            del src_all, dst_all
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            cb, ref = cpu_baseline(PARAMS, (pv_img, pv_params) if pv_img is not None else None)
            out["cpu_baseline"] = cb
            if "preview_call" in cb:
                pc = cb.pop("preview_call")
                out["preview_call"]["cpu_oracle_ms"] = pc["ms"]
                out["preview_call"]["vs_oracle"] = {k: pc[k] for k in ("gpu_window_max_abs_diff", "green_alpha_equal")}
                good = pc["gpu_window_max_abs_diff"] <= 1 and pc["green_alpha_equal"]
                out["checks"]["preview_within_1_lsb_of_cpu_oracle"] = bool(good)
                out["ok"] = bool(out["ok"] and good)
                ok = ok and good
        line = compact_line(out)
        try:                                   # the whole record (prose, plans, every phase) next to the line
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        if line["ok"]:
            print(json.dumps(line), flush=True)
        else:                                  # a job whose check failed on any rank has no result line, only a non-zero exit
            print("bench.py: parity check failed; the record without a result line: " + json.dumps(line), file=sys.stderr, flush=True)

    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("parity check failed")


if __name__ == "__main__":
    main()
