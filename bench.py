#!/usr/bin/env python3
"""bench.py — Mpixels/s of the lateral-CA cubic pixel loop on 6000x4000 RGBA8.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One step = one pass of the hot path (fixca_batch_device -> the row-rolling kernel,
which takes up to 32 images of a batch per launch) over a batch of B independent
6000x4000 RGBA8 synthetic images already resident in HBM (B src + B dst buffers;
B >= 4 so the working set exceeds the 256 MB Infinity Cache).  Workload = BASELINE.json config "lateral cubic + off-centre lens
(658,1280)", blue +2 / red -2.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank owns B images of a global batch of N*B (image i -> rank
i mod N), rank 0 broadcasts the parameter block over RCCL, and `value` is the
whole-job rate (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))

W, H, C = 6000, 4000, 4
PARAMS = dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
METRIC = "Mpixels/s lateral-CA cubic on 6000\u00d74000 RGBA; % of HBM-read roofline"     # BASELINE.json's metric
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_PX = 2 * C        # every source byte read once, every result byte written once


def cpu_baseline(params_dict):
    """The CPU oracle (the reference's algorithm restated in C, oracle/) timed on this
    host: 1 thread like the reference, plus an all-cores row-band run as context."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pyoracle
    rng = np.random.default_rng(0xF1CA)
    img = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
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
    return {
        "value": round(W * H / best / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": "3 passes over one 6000x4000 RGBA8 noise image (24 Mpx each), best pass, gcc -O2 no -march",
        "all_cores": {"value": round(W * H / best_mt / 1e6, 1), "threads": nthr},
        "host": {"cpu": model, "nproc": ncpu},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras that launch other kernels or single images (copy yardstick, "
                         "host-entry run); tools/profile.sh uses it so that rocprof's per-kernel average is over "
                         "the timed launches only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fixca                      # raises if libfixca_hip.so is missing: no fallback
    from fixca import dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    # FIXCA_BENCH_REHEARSE=1: several ranks share the one GPU of a test box and talk over gloo, to
    # rehearse the N > 1 control flow; the real run is one rank per GPU over RCCL
    rehearse = os.environ.get("FIXCA_BENCH_REHEARSE") == "1"
    local = local % torch.cuda.device_count() if rehearse else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == max(1, args.gpus) or world == 1, "launch N>1 with torch.distributed.run"

    # rank 0 owns the parameters; everyone else receives them over RCCL
    params = fdist.broadcast_params(fixca.make_params(**PARAMS) if rank == 0 else None, src=0, device=dev)
    kid, plan_text = fixca.plan(W, H, C, 1, params, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)   # what fix_ca_batch will run

    B = max(4, args.batch)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xF1CA1000 + rank)
    srcs = [torch.randint(0, 256, (H, W, C), dtype=torch.uint8, device=dev, generator=gen) for _ in range(B)]
    dsts = [torch.empty_like(s) for s in srcs]

    def step():
        fixca.fix_ca_batch(srcs, dsts, W, H, C, 1, params)

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
        t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = t.tolist()

    # ---- not timed: sanity of what was just computed, yardsticks, CPU baseline (rank 0) ----
    ok = True
    extra = {}
    if rank == 0:
        exact = torch.empty_like(srcs[0])
        fixca.fix_ca_region(srcs[0], exact, W, H, C, 1, params, 0, W, 0, H, True, fixca.KERNEL_EXACT)
        torch.cuda.synchronize()
        diff = (dsts[0].to(torch.int16) - exact.to(torch.int16)).abs()
        extra["parity_vs_fp64_kernel"] = {"max_abs_diff": int(diff.max()), "differing_samples": int((diff != 0).sum()),
                                          "samples": int(diff.numel())}
        ok = extra["parity_vs_fp64_kernel"]["max_abs_diff"] <= 1
        del exact, diff
        if not args.no_extras:
            copy_ms = fixca.time_copy(srcs, dsts, 10 * B, W * H * C)
            extra["copy_kernel_GBs"] = round(2 * W * H * C / (copy_ms * 1e-3) / 1e9, 1)
            import numpy as np
            hsrc = srcs[0].cpu().numpy()
            hdst = np.empty_like(hsrc)
            fixca.fix_ca_region(hsrc, hdst, W, H, C, 1, params, 0, W, 0, H, True)     # warm (allocs, context)
            t = time.perf_counter()
            fixca.fix_ca_region(hsrc, hdst, W, H, C, 1, params, 0, W, 0, H, True)
            extra["host_entry_pcie_inclusive_mpix_s"] = round(W * H / (time.perf_counter() - t) / 1e6, 1)
            # the same call on page-locked buffers (fixca_host_alloc), as INTEGRATION.md suggests for fix_ca()
            psrc, pdst = fixca.pinned_empty(hsrc.shape), fixca.pinned_empty(hsrc.shape)
            psrc[...] = hsrc
            fixca.fix_ca_region(psrc, pdst, W, H, C, 1, params, 0, W, 0, H, True)
            t = time.perf_counter()
            fixca.fix_ca_region(psrc, pdst, W, H, C, 1, params, 0, W, 0, H, True)
            extra["host_entry_pinned_pcie_inclusive_mpix_s"] = round(W * H / (time.perf_counter() - t) / 1e6, 1)
            ok = ok and bool(np.array_equal(pdst, hdst))
            fixca.pinned_free(psrc)
            fixca.pinned_free(pdst)
            fixca.lib.fixca_release_cache()
    ok = fdist.all_ranks_ok(ok, device=dev)

    if rank == 0:
        # the row-rolling kernel (plan id 6) takes up to 32 images of a batch per launch; every
        # other kernel is one launch per image
        imgs_per_launch = min(B, 32) if kid == 6 else 1
        launches = args.steps * ((B + imgs_per_launch - 1) // imgs_per_launch)
        kernel_ms = ev_ms / launches
        algo_bytes = ALGO_BYTES_PER_PX * W * H * imgs_per_launch
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            # recorded per launch of the profiled run; scaled if this run batches differently
            traffic = round(t.get("bytes_per_launch") * imgs_per_launch / max(1, t.get("images_per_launch", 1)))
        out = {
            "metric": METRIC,
            "value": round(world * args.steps * B * W * H / wall / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "6000x4000 RGBA8, lateral blue +2 / red -2, lens (658,1280), cubic (BASELINE configs[2])",
                       "images_per_gpu_per_step": B, "global_batch": B * world,
                       "parallelism": "image i -> rank i mod %d, params broadcast from rank 0" % world,
                       "kernel_plan": plan_text},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_us": round(kernel_ms * 1e3, 2), "algorithmic_bytes_per_launch": algo_bytes,
                         "images_per_launch": imgs_per_launch, "us_per_image": round(kernel_ms * 1e3 / imgs_per_launch, 2)},
            "ok": ok,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(PARAMS)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("parity check failed")


if __name__ == "__main__":
    main()
