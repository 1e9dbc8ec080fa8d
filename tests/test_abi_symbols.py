"""CPU checks of the C-ABI: the library loads, exports every symbol the header
declares, refuses bad calls with the documented codes, and never computes
without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import fixca


def _declared(repo_root):
    text = open(os.path.join(repo_root, "include", "fixca_hip.h")).read()
    return sorted(set(re.findall(r"FIXCA_API\s+[\w\s\*]+?\b(fixca_\w+)\s*\(", text)))


def test_header_symbols_exported(repo_root):
    names = _declared(repo_root)
    assert len(names) >= 17
    raw = ctypes.CDLL(fixca.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "missing export " + n
    assert sorted(fixca.ABI_SYMBOLS) == names       # the binding covers exactly the header


def test_no_internal_symbols_leak():
    raw = ctypes.CDLL(fixca.LIB_PATH)
    for n in ("fixca_launch", "fixca_make_plan", "fixca_fail"):
        assert not hasattr(raw, n)


def test_params_layout():
    assert ctypes.sizeof(fixca.FixCaParams) == 80
    offs = {n: getattr(fixca.FixCaParams, n).offset for n, _ in fixca.FixCaParams._fields_}
    assert offs == dict(blue=0, red=8, lens_x=16, lens_y=24, update_preview=32, interpolation=36,
                        saturation=40, x_blue=48, x_red=56, y_blue=64, y_red=72)
    d = fixca.default_params()                      # fix-ca.c:85-97
    assert (d.lens_x, d.lens_y, d.interpolation, d.update_preview) == (-1.0, -1.0, 1, 1)
    assert fixca.lib.fixca_abi_version() == 1


def test_params_check_matches_run_range():
    ok = fixca.make_params(blue=30, red=-30, interpolation=2, x_blue=-30, y_red=30)
    assert fixca.params_check(ok)
    for bad in (dict(blue=30.01), dict(red=-31), dict(interpolation=3), dict(interpolation=-1),
                dict(x_blue=31), dict(y_red=float("nan"))):
        assert not fixca.params_check(fixca.make_params(**bad)), bad


@pytest.mark.parametrize("kwargs,code", [
    (dict(x1=1), fixca.E_UNSUPPORTED),              # partial columns: undefined in the reference
    (dict(x2=10), fixca.E_UNSUPPORTED),
    (dict(y1=-1), fixca.E_INVALID),
    (dict(y2=99), fixca.E_INVALID),
    (dict(y1=9, y2=3), fixca.E_INVALID),
])
def test_plan_refusals(kwargs, code):
    p = fixca.make_params(blue=1, red=-1, lens_x=16, lens_y=8)
    with pytest.raises(fixca.FixCaError) as e:
        fixca.plan(32, 16, 4, 1, p, **kwargs)
    assert e.value.code == code and str(e.value)


def test_preview_call_is_planned_unless_it_needs_saturation():
    p = fixca.make_params(blue=1, red=-1, lens_x=16, lens_y=8)
    assert fixca.plan(32, 16, 4, 1, p, show_progress=False)[0] >= 1          # preview_update()'s call: centre lines
    q = fixca.make_params(blue=1, red=-1, lens_x=16, lens_y=8, saturation=25.0)
    assert fixca.plan(32, 16, 4, 1, q, show_progress=True)[0] >= 1           # ignored in the final render
    with pytest.raises(fixca.FixCaError) as e:
        fixca.plan(32, 16, 4, 1, q, show_progress=False)                      # HSV boost: libgimpcolor, unpinned
    assert e.value.code == fixca.E_UNSUPPORTED


def test_plan_refuses_formats_and_domain():
    p = fixca.make_params(blue=1, red=-1, lens_x=16, lens_y=8)
    for bytes_, bpc in ((32, 8), (24, 8), (10, 2), (16, -8), (2, 1), (5, 1), (4, 3)):       # u64 needs x87 long double
        with pytest.raises(fixca.FixCaError) as e:
            fixca.plan(32, 16, bytes_, bpc, p)
        assert e.value.code == fixca.E_UNSUPPORTED
    for bytes_, bpc in ((8, 2), (6, 2), (16, 4), (16, -4), (12, -4), (32, -8)):     # u16, u32, float, double
        k, desc = fixca.plan(32, 16, bytes_, bpc, p)
        assert k == 8 and "bpc=%d" % bpc in desc      # NONE (make_params' default): the byte mover on every wide type
        # CUBIC: the LDS-ring kernel on RGBA double, the register-queue kernel elsewhere
        assert fixca.plan(32, 16, bytes_, bpc, fixca.make_params(blue=1, red=-1, lens_x=16, lens_y=8, interpolation=2))[0] == (10 if (bytes_, bpc) == (32, -8) else 7)
        k, desc = fixca.plan(32, 16, bytes_, bpc, p, kernel=fixca.KERNEL_EXACT)
        assert k == 1 and "bpc=%d" % bpc in desc                     # the one-thread-per-pixel fp64 kernel
        # a scale that drifts more than a window over one strip: k_wcol gives up; on RGBA the ring kernel narrows its
        # strips instead, on RGB only the one-thread-per-pixel kernel is left
        far = fixca.make_params(blue=-29.0, red=29.0, lens_x=600, lens_y=8, interpolation=2)
        kid_far = fixca.plan(1200, 16, bytes_, bpc, far)[0]
        assert kid_far in ((1, 10) if (bytes_, bpc) == (32, -8) else (1,))
        if kid_far == 1:                                 # no fast kernel there: KERNEL_FAST says so instead of falling back
            with pytest.raises(fixca.FixCaError) as e:
                fixca.plan(1200, 16, bytes_, bpc, far, kernel=fixca.KERNEL_FAST)
            assert e.value.code == fixca.E_UNSUPPORTED
    with pytest.raises(fixca.FixCaError) as e:      # max_dim 16 + (-20) <= 0
        fixca.plan(32, 16, 4, 1, fixca.make_params(blue=-20, lens_x=16, lens_y=8))
    assert e.value.code == fixca.E_DOMAIN
    with pytest.raises(fixca.FixCaError) as e:
        fixca.plan(32, 16, 4, 1, fixca.make_params(interpolation=7))
    assert e.value.code == fixca.E_DOMAIN


def test_planner_choices():
    cub = fixca.make_params(blue=2, red=-2, lens_x=658, lens_y=1280, interpolation=2)
    k, desc = fixca.plan(6000, 4000, 4, 1, cub)                         # RGBA, W % 4 == 0: LDS ring, rolling window
    assert k == 6 and "roll cubic" in desc and "oc=240" in desc and "ring=16" in desc   # 240 px = 960 B: 64-byte aligned stores
    kb, descb = fixca.plan(6000, 4000, 4, 1, cub, kernel=fixca.PLAN_BATCH)  # one of many equal calls: shorter waves
    assert kb == 6 and int(re.search(r"th=(\d+)", descb).group(1)) < int(re.search(r"th=(\d+)", desc).group(1))
    lin = fixca.make_params(blue=2, red=-2, lens_x=3000, lens_y=2000, interpolation=1)
    k, desc = fixca.plan(6000, 4000, 4, 1, lin)
    assert k == 6 and "roll linear" in desc
    k, desc = fixca.plan(6000, 4000, 4, 1, fixca.make_params(blue=1.5, red=-1.5, lens_x=3000, lens_y=2000))
    assert k == 6 and "roll none" in desc                               # NONE runs as an exact 1-tap filter
    k, desc = fixca.plan(6000, 4000, 3, 1, cub)                         # RGB: the same kernel compiled for 3 bytes per pixel
    assert k == 6 and "roll cubic bpp=3" in desc
    assert fixca.plan(6000, 4000, 3, 1, fixca.make_params(blue=1.5, red=-1.5, lens_x=3000, lens_y=2000))[0] == 6
    k, desc = fixca.plan(6002, 4000, 3, 1, cub)                         # width not a multiple of 4: the main kernel since round 4 (rows at byte addresses)
    assert k == 6 and "roll cubic bpp=3" in desc
    assert fixca.plan(257, 193, 4, 1, cub)[0] == 6                      # ... RGBA8 too (a last lane of one pixel)
    # amounts of about -4 and beyond: outputs clamped to the row's first / last pixel lie out of their neighbours' 5 window
    # pixels' reach: the main kernel overwrites them (scan / register-window kernels until round 4)
    assert "edge-pixels" not in desc and "edge-pixels" not in fixca.plan(6000, 4000, 4, 1, cub)[1]
    for bpp in (4, 3):
        for interp in (0, 1, 2):
            for blue, red in ((5, -5), (-6, -6), (12, -12), (-4, 9)):
                k, desc = fixca.plan(6000, 4000, bpp, 1, fixca.make_params(blue=blue, red=red, lens_x=658, lens_y=1280, interpolation=interp))
                assert k == 6 and ("edge-pixels" in desc or (blue, red) == (-4, 9)), (bpp, interp, blue, red, desc)
    k, desc = fixca.plan(255, 193, 4, 1, cub)                           # narrower than the 256-column window: register-window kernel
    assert k == 3 and "fast-interp cubic" in desc
    assert fixca.plan(6000, 4000, 4, 1, cub, kernel=fixca.KERNEL_EXACT)[0] == 1
    assert fixca.plan(6, 6, 3, 1, fixca.make_params(lens_x=3, lens_y=3, interpolation=1))[0] == 1   # too narrow
    with pytest.raises(fixca.FixCaError):
        fixca.plan(6, 6, 3, 1, fixca.make_params(lens_x=3, lens_y=3, interpolation=1), kernel=fixca.KERNEL_FAST)
    # strong magnification: a strip's union window no longer fits 256 columns at full width
    k, desc = fixca.plan(2000, 100, 4, 1, fixca.make_params(red=-30, lens_x=200, lens_y=50, interpolation=2))
    oc = int(re.search(r"oc=(\d+)", desc).group(1))
    assert k in (3, 4) and oc < 248 and oc % 4 == 0                     # tap offsets drift too far for the rolling kernel
    # the +-30 px directional extreme keeps 60+ rows alive: too many for the ring, register-window kernel instead
    far = fixca.make_params(lens_x=1000, lens_y=500, interpolation=2, y_blue=30, y_red=-30)
    assert fixca.plan(2000, 1000, 4, 1, far)[0] == 3
    assert fixca.plan(2000, 100, 4, 1, cub, y1=10, y2=10)[0] in (3, 4, 6)  # empty row range is legal
    # images beyond 2 GB: the rolling kernel's 32-bit offsets are relative to each wave's own rows
    big = fixca.make_params(blue=2, red=-2, lens_x=15000, lens_y=10000, interpolation=2)
    assert fixca.plan(30000, 20000, 4, 1, big)[0] == 6


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    img = np.zeros((16, 32, 4), np.uint8)
    with pytest.raises(fixca.FixCaError) as e:
        fixca.fix_ca_image(img, fixca.make_params(blue=1, lens_x=16, lens_y=8))
    assert e.value.code in (fixca.E_NODEVICE, fixca.E_HIP)


def test_gimp_box_binding(tmp_path):
    """tools/gimp_box: the binding compiles next to the plug-in's own FixCaParams declaration, and the
    patcher puts it at the two anchors of the reference's fix-ca.c and nowhere else (skipped where the
    reference tree is absent, e.g. on the GPU box)"""
    import subprocess
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(root, "tests", "c", "binding_check.c"), "-o", str(tmp_path / "b.o")])
    ref = "/root/reference/fix-ca.c"
    if not os.path.exists(ref):
        pytest.skip("no reference tree here")
    work = tmp_path / "fix-ca.c"
    shutil.copyfile(ref, work)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gimp_box", "apply_binding.py"), str(work), root])
    old = open(ref, encoding="utf-8").read().split("\n")
    new = open(work, encoding="utf-8").read().split("\n")
    import difflib
    ops = [op for op in difflib.SequenceMatcher(None, old, new, autojunk=False).get_opcodes() if op[0] != "equal"]
    assert [op[0] for op in ops] == ["insert", "insert"] and len(new) == len(old) + 2 + 8   # two insertions, nothing removed or changed
    i_inc = new.index("#include <fixca_hip.h>")
    assert "gimpui.h" in new[i_inc - 2]
    i_call = next(i for i, l in enumerate(new) if "fixca_region_host" in l)
    # right behind the statement that opens the progress bar, inside the definition of fix_ca_region, before its row buffers
    assert any("gimp_progress_init" in l for l in new[i_call - 4:i_call])
    i_def = max(i for i, l in enumerate(new[:i_call]) if "fix_ca_region" in l and l.lstrip().startswith("static"))
    assert "{" in [l.strip() for l in new[i_def:i_def + 6]]                  # the definition, not the prototype
    assert not any(l.startswith("}") for l in new[i_def:i_call])             # still inside that function
    assert "gimp_progress_update (1.0)" in "\n".join(new[i_call:i_call + 6])  # the bar is filled before returning
    assert sum("fixca_region_host" in l for l in new) == 1
    # running it twice changes nothing more
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gimp_box", "apply_binding.py"), str(work), root])
    assert open(work, encoding="utf-8").read().split("\n") == new
