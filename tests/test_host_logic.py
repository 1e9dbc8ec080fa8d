"""CPU tests of the host-side logic: the planner's strip widths against a numpy
model of the row-scan kernel, and the model against the oracle (<= 1 LSB)."""
import os
import re

import numpy as np
import pytest

import fixca
import kernel_model
import pyoracle
from fixca import synth
from helpers import diff_stats

PSETS = [
    dict(blue=1.5, red=-1.5, lens_x=0.5, lens_y=0.5),                 # lens as a fraction of the size
    dict(blue=6.0, red=-2.4, lens_x=0, lens_y=0),
    dict(lens_x=0.5, lens_y=0.5, x_blue=1.0, y_blue=-7.5, y_red=4.0),
    dict(blue=-3.3, red=4.7, lens_x=-1, lens_y=-1, x_blue=2.5, x_red=-1.25, y_blue=.75, y_red=-3),
    dict(blue=-20, red=25, lens_x=0.33, lens_y=0.5, x_blue=-29, x_red=30, y_blue=30, y_red=-30),
]


def _params(kw, w, h, interp):
    kw = dict(kw)
    for k, n in (("lens_x", w), ("lens_y", h)):
        if 0 < kw[k] < 1:
            kw[k] = int(kw[k] * n)
    return kw


@pytest.mark.parametrize("size", [(257, 193, 4), (300, 70, 3), (64, 40, 4), (1000, 50, 4)])
@pytest.mark.parametrize("pset", range(len(PSETS)))
@pytest.mark.parametrize("interp", [1, 2])
def test_model_within_one_lsb(size, pset, interp, monkeypatch):
    monkeypatch.setenv("FIXCA_NO_SCAN", "1")       # the model describes the register-window kernel
    monkeypatch.setenv("FIXCA_NO_ROLL", "1")
    w, h, c = size
    kw = _params(PSETS[pset], w, h, interp)
    p = fixca.make_params(interpolation=interp, **kw)
    k, desc = fixca.plan(w, h, c, 1, p)
    assert k == 3, desc
    oc = int(re.search(r"oc=(\d+)", desc).group(1))
    img = synth.noise_image(5, w, h, c)
    ref = pyoracle.region(img, pyoracle.make_params(interpolation=interp, **kw))
    out = kernel_model.run(img, p, oc)
    mx, n = diff_stats(out, ref)
    assert mx <= 1
    assert n <= 0.02 * ref.size                   # ties at .5 are the only disagreements


def test_model_smooth_image_mostly_exact():
    img = synth.smooth_image(257, 193, 4)
    kw = dict(blue=2.0, red=-2.0, lens_x=28, lens_y=61)
    p = fixca.make_params(interpolation=2, **kw)
    ref = pyoracle.region(img, pyoracle.make_params(interpolation=2, **kw))
    out = kernel_model.run(img, p, 252)
    mx, n = diff_stats(out, ref)
    assert mx <= 1 and n <= 0.001 * ref.size


# ---- k_roll's tile schedule (RollSched, csrc/fixca_roll.h) ---------------------------------------------

@pytest.mark.parametrize("shape", [(6000, 4000, 4, 32), (6000, 4000, 3, 33), (8192, 8192, 4, 4), (768, 400, 4, 8), (1024, 257, 4, 1)])
def test_roll_schedule_covers_every_row_once(shape):
    """what k_roll's workgroups do with a RollSched: every row of every column exactly once"""
    w, h, c, n = shape
    p = fixca.make_params(blue=2.0, red=-2.0, lens_x=w * 0.11, lens_y=h * 0.32, interpolation=2)
    kid, desc = fixca.plan(w, h, c, 1, p, kernel=fixca.KERNEL_AUTO | (fixca.PLAN_BATCH if n > 1 else 0))
    assert kid == 6, desc
    nstrips = int(re.search(r"strips=(\d+)", desc).group(1))
    sc = fixca.debug_sched(w, h, c, p, n)
    assert sc["ncols"] == n * nstrips and sc["rows"] == h and sc["pool_blocks"] % 8 == 0 and sc["pool_blocks"] >= sc["pool_waves"]
    cover = np.zeros((sc["ncols"], h), np.int32)
    for img in range(n):
        for bid in range(sc["pool_blocks"]):
            tile = (bid & 7) * (sc["pool_blocks"] >> 3) + (bid >> 3)          # xcd_remap
            if tile >= sc["pool_waves"]:
                continue
            strip, chunk = tile % nstrips, tile // nstrips
            y0 = chunk * sc["th"]
            cover[img * nstrips + strip, y0:min(y0 + sc["th"], h)] += 1
    assert np.all(cover == 1)


def test_tuning_knobs_need_debug(monkeypatch):
    """FIXCA_TH and friends change the plan only under FIXCA_DEBUG (a stray variable in a GIMP
    session's environment must not reshape the kernels)"""
    p = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2)
    monkeypatch.delenv("FIXCA_DEBUG", raising=False)
    base = fixca.plan(6000, 4000, 4, 1, p)[1]
    monkeypatch.setenv("FIXCA_TH", "7")
    monkeypatch.setenv("FIXCA_NO_ROLL", "1")
    assert fixca.plan(6000, 4000, 4, 1, p)[1] == base
    monkeypatch.setenv("FIXCA_DEBUG", "1")
    k, desc = fixca.plan(6000, 4000, 4, 1, p)
    assert k != 6 and desc != base
    monkeypatch.delenv("FIXCA_NO_ROLL")
    assert "th=7 " in fixca.plan(6000, 4000, 4, 1, p)[1]
    monkeypatch.setenv("FIXCA_DEBUG", "0")
    assert fixca.plan(6000, 4000, 4, 1, p)[1] == base


def test_host_cache_is_per_thread():
    """fixca_region_host's cached device images belong to the calling thread: nothing is
    shared through a process-wide static, so one thread can never free what another is using"""
    import threading
    seen = {}

    def worker(name):
        seen[name] = fixca.debug_host_cache()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen[0] == (None, None, 0) and seen[1] == (None, None, 0)


def test_no_store_data_hazard_in_kernels(tmp_path):
    """A 16-byte buffer store must not be followed directly by a VALU write of its data registers: on MI355X the
    overwrite reaches the store when the store's scalar offset is a register (tools/store_hazard_scan.py, profiles/NOTES.md).
    Compiles the kernels that use buffer stores to assembly with the library's flags and scans them."""
    import shutil, subprocess, sys, os
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import store_hazard_scan
    # every translation unit of the library (the Makefile's HIPSRCS): the RGB8 compilation of the rolling kernel stores
    # 12 bytes per lane, the NONE mover 16, and a compiler change could bring the hazard back in any of them
    mk = open(os.path.join(root, "gimp-fix-ca_amd", "csrc", "Makefile")).read()
    srcs = re.search(r"^HIPSRCS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    assert {"fixca_wide.hip", "fixca_roll.hip", "fixca_roll_rgb.hip", "fixca_move.hip"} <= set(srcs)
    flags = re.search(r"^CXXFLAGS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    nstores = 0
    for src in srcs:
        out = tmp_path / (src + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950"] + flags + ["-S", "--cuda-device-only",
                        os.path.join(root, "gimp-fix-ca_amd", "csrc", src), "-o", str(out)], check=True, capture_output=True)
        nstores += sum(1 for l in open(out) if re.match(r"\s+(buffer|global)_store_dwordx[34]\s", l))
        direct = [h for h in store_hazard_scan.scan(str(out)) if "(+1)" in h]
        assert not direct, direct[:3]
    assert nstores > 50                                  # the scan saw the stores it is about


def test_bench_launcher_relays_the_ranks_exit_code(repo_root):
    """`python bench.py --gpus 2` without WORLD_SIZE starts its own ranks as child processes and returns
    their exit code.  On a machine without a GPU the ranks refuse to run: the launcher must say so with a
    non-zero code and no result line (a driver must never read a number out of a failed run)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the command would run the real bench (tests/test_gpu_parity.py covers that)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--global-batch", "4",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "no GPU visible" in r.stderr
    # asking for N ranks while a different world is already set up is refused, not silently re-interpreted
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


# ---- planner of the LDS-ring kernels of the wide sample types (k_move: NONE, k_lin: LINEAR); no GPU needed ----------

def _plan_fields(desc):
    f = dict(kv.split("=") for kv in desc.split() if "=" in kv)
    ring, nu = f["ring"].rstrip("K").split("x")
    return dict(oc=int(f["oc"]), th=int(f["th"]), ring=int(ring), nu=int(nu), depth=int(f["depth"]), strips=int(f["strips"]),
                chunks=int(f["chunks"]), waves=int(f["waves"]), bpp=int(f["bpp"]))


@pytest.mark.parametrize("bpc,nch", [(2, 4), (2, 3), (4, 4), (-4, 4), (-4, 3), (-8, 4), (-8, 3)], ids=str)
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_ring_kernel_plans_of_the_wide_types(bpc, nch, interp):
    """k_move takes NONE on every wide format whose rows are whole dwords, k_lin / k_lin_rgb LINEAR on u32 / float /
    double (RGBA and RGB), k_cub CUBIC on RGBA double and batched RGBA float; strips are whole 16-byte units and whole pixels; a one-image launch is ONE generation (never more
    waves than the ring's LDS footprint leaves slots); the batch hint changes the shape, not the kernel"""
    sb = abs(bpc)
    pb = nch * sb
    for w, h, kw in ((6000, 4000, dict(blue=2.0, red=-2.0, lens_x=658, lens_y=1280)), (1202, 300, dict(blue=0.4, red=-0.3, lens_x=1000, lens_y=30, x_blue=-3.0, y_red=2.0)),
                     (8192, 8192, dict(lens_x=4096, lens_y=4096, x_blue=1.0, y_blue=-7.5, y_red=4.0))):
        p = fixca.make_params(interpolation=interp, **kw)
        kid, desc = fixca.plan(w, h, pb, bpc, p)
        kidb, descb = fixca.plan(w, h, pb, bpc, p, kernel=fixca.KERNEL_AUTO | fixca.PLAN_BATCH)
        if interp == 0:
            assert kid == 8 and kidb == 8, desc
        elif interp == 2 and nch == 4 and bpc in (-4, -8):
            assert kidb == 10 and kid == (10 if bpc == -8 else 7), desc      # floats: batches only
            if kid != 10:
                desc = descb
        elif interp == 1 and bpc == 2 and nch == 4:
            assert kid == 7 and kidb in (7, 9), desc             # 16-bit RGBA: the ring kernel's weight form in batches, except on shifts by halves
            if kidb == 7:
                continue
            assert "weights+recheck" in descb
            desc = descb
        elif interp == 1 and bpc != 2:
            assert kid == 9 and kidb == 9, desc
        else:
            assert kid == 7 and kidb == 7, desc              # 16-bit and RGB 4-byte LINEAR stay with the rolling kernel
            continue
        for d in (desc, descb):
            f = _plan_fields(d)
            assert f["bpp"] == pb and f["nu"] in (1, 2, 4) and 1 <= f["depth"] <= 14
            assert (f["oc"] * pb) % 16 == 0                          # whole units and whole pixels
            assert f["oc"] * pb <= (1536 if (kid == 9 and nch == 3) else 64 * f["nu"] * 16)    # 64 lanes x nu units (RGB LINEAR: x 24 bytes)
            assert f["strips"] == -(-w // f["oc"]) and f["waves"] == f["strips"] * f["chunks"]
            assert f["ring"] * f["nu"] * 1024 <= 64 * 1024
            assert f["ring"] >= f["depth"] + 2
        f = _plan_fields(desc)
        slots = min(16, (160 * 1024) // (f["ring"] * f["nu"] * 1024 + 64)) * 256
        if interp == 0 or (interp == 1 and nch == 4 and bpc != 2):
            assert f["waves"] <= slots, desc                         # the movers and RGBA LINEAR, one image per launch: one generation of fat waves
        # (RGB LINEAR and CUBIC are bound by their arithmetic: many short tiles, several generations)
        assert _plan_fields(descb)["th"] in (32, 64)                 # many images per launch: short tiles


def test_ring_kernel_plans_fall_back():
    """what the ring kernels refuse goes to the kernels that can: shifts beyond a ring row's reach fall back to the
    one-thread-per-pixel kernel, RGB u16 rows of odd width (not whole dwords: the last halfword of a tile's source rows
    cannot be fetched by a dword that stays inside the buffer) go to k_wcol's one-tap instantiation, and the debug
    switches select the round-2 paths for A/B runs"""
    p = fixca.make_params(blue=2.0, red=-2.0, lens_x=100, lens_y=50, interpolation=0)
    assert fixca.plan(257, 193, 6, 2, p)[0] == 7                     # RGB u16, odd width
    assert fixca.plan(258, 193, 6, 2, p)[0] == 8
    # a scale far from 1: the ring kernel narrows its strips until every strip's window fits a ring row (k_wcol, whose
    # per-channel windows are 128 columns, gives up on this one: tests/test_abi_symbols.py)
    far = fixca.make_params(blue=-29.0, red=29.0, lens_x=600, lens_y=8, interpolation=1)
    kid, desc = fixca.plan(1200, 16, 16, -4, far)
    assert kid == 9 and _plan_fields(desc)["oc"] < 124, desc        # 2-KiB rows hold 128 columns of RGBA float: 80-column strips here
    import os
    os.environ["FIXCA_NO_MOVE"] = "1"
    try:
        assert fixca.plan(6000, 4000, 8, 2, p)[0] == 7 and fixca.plan(6000, 4000, 16, -4, p)[0] == 1
    finally:
        del os.environ["FIXCA_NO_MOVE"]
    os.environ["FIXCA_NO_LIN"] = "1"
    try:
        assert fixca.plan(6000, 4000, 16, -4, fixca.make_params(blue=2.0, red=-2.0, lens_x=100, lens_y=50, interpolation=1))[0] == 7
    finally:
        del os.environ["FIXCA_NO_LIN"]


def test_bench_line_stays_small_and_carries_the_evidence():
    """the driver's record keeps the fixed keys (roofline whole, scalars only) and about 8 KB of output: the line bench.py
    prints must stay under 6 KB and hold, inside `roofline`, the figures a reader needs first (a full record of a real run,
    profiles/r04x_bench_full.json, through bench.compact_line)"""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r04x_bench_full.json")))
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < 6144
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "single_image_us", "single_image_frac", "copy_frac",
              "oracle_max_abs_diff", "oracle_differing", "rgb8_frac", "min_config_frac", "min_config", "second_allocation_frac",
              "launch_frac_min", "launch_frac_max", "traffic_source"):
        assert k in roof, k
    assert all(not isinstance(v, (dict, list)) for v in roof.values())      # scalars only: nested values are dropped from the record
    assert line["metric"] == full["metric"] and line["cpu_baseline"]["kind"] == "port"
    assert len(line["configs"]) == len(full["configs"]) >= 28 and len(line["configs_columns"]) == len(line["configs"][0])
    assert len({c["name"] for c in bench.CONFIGS}) == len(bench.CONFIGS) <= 32     # (more rows would not fit the line)
