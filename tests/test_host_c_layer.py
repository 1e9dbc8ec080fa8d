"""CPU tests of the plain-C host layer (gimp-fix-ca_amd/host): run()'s
non-interactive argument handling, color_size(), and that fix_ca()/fix_ca_region()
fail loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import fixca

HOST = os.path.join(os.path.dirname(fixca.LIB_PATH), "libfixca_host.so")
CLI = os.path.join(os.path.dirname(fixca.LIB_PATH), "fixca_cli")
SUCCESS, CALLING_ERROR = 3, 1


class Data(ctypes.Union):
    _fields_ = [("d_int32", ctypes.c_int32), ("d_int8", ctypes.c_int8), ("d_float", ctypes.c_double),
                ("pad_", ctypes.c_double * 4)]


class PdbArg(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("data", Data)]


class Drawable(ctypes.Structure):
    _fields_ = [("pixels", ctypes.c_void_p), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("bpp", ctypes.c_int), ("bpc", ctypes.c_int), ("sel_x", ctypes.c_int), ("sel_y", ctypes.c_int),
                ("sel_w", ctypes.c_int), ("sel_h", ctypes.c_int)]


@pytest.fixture(scope="module")
def host():
    lib = ctypes.CDLL(HOST)
    lib.fixca_unmarshal_noninteractive.argtypes = [ctypes.c_int, ctypes.POINTER(PdbArg), ctypes.POINTER(fixca.FixCaParams)]
    lib.fixca_color_size.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.fix_ca.argtypes = [ctypes.POINTER(Drawable), ctypes.POINTER(fixca.FixCaParams)]
    return lib


def _args(vals):
    """vals = PDB args 3.. (blue, red, lens_x, lens_y, interp, xb, xr, yb, yr)"""
    a = (PdbArg * 12)()
    for i, v in enumerate(vals):
        k = i + 3
        if k == 7:
            a[k].data.d_int8 = int(v)
        else:
            a[k].data.d_float = float(v)
    return a, 3 + len(vals)


def test_gimpparam_layout():
    assert ctypes.sizeof(PdbArg) == 40 and PdbArg.data.offset == 8      # libgimp 2.10 GimpParam on LP64


def test_shipped_test_arguments_hit_the_lens_quirk(host):
    """tests/Makefile.am:18 passes lens 658 1280 as FLOATs; run() reads d_int32 (fix-ca.c:254,258),
    i.e. the low 32 bits of the double: 0 for both."""
    a, n = _args([6.0, -2.4, 658, 1280, 1, 0.0, 0.0, 0.0, 0.0])
    p = fixca.FixCaParams()
    assert host.fixca_unmarshal_noninteractive(n, a, ctypes.byref(p)) == SUCCESS
    assert (p.blue, p.red, p.interpolation) == (6.0, -2.4, 1)
    assert (p.lens_x, p.lens_y) == (0.0, 0.0)


def test_defaults_for_omitted_arguments(host):
    a, n = _args([1.5, -1.5])                       # nparams 5: fix-ca.c:251-278
    p = fixca.FixCaParams()
    assert host.fixca_unmarshal_noninteractive(n, a, ctypes.byref(p)) == SUCCESS
    assert (p.lens_x, p.lens_y, p.interpolation) == (-1.0, -1.0, 0)
    assert (p.x_blue, p.x_red, p.y_blue, p.y_red) == (0.0, 0.0, 0.0, 0.0)


@pytest.mark.parametrize("vals", [[30.5, 0], [0, -31], [1, 1, 0, 0, 3], [1, 1, 0, 0, 2, 31], [1, 1, 0, 0, 2, 0, 0, 0, -30.1]])
def test_range_check(host, vals):
    a, n = _args(vals)
    assert host.fixca_unmarshal_noninteractive(n, a, ctypes.byref(fixca.FixCaParams())) == CALLING_ERROR


def test_nparams_bounds(host):
    a, _ = _args([1, 1])
    p = fixca.FixCaParams()
    assert host.fixca_unmarshal_noninteractive(4, a, ctypes.byref(p)) == CALLING_ERROR
    assert host.fixca_unmarshal_noninteractive(13, a, ctypes.byref(p)) == CALLING_ERROR
    assert host.fixca_unmarshal_noninteractive(12, a, ctypes.byref(p)) == SUCCESS


@pytest.mark.parametrize("name,bpp,want", [
    (b"R'G'B' u8", 3, 1), (b"R'G'B'A u8", 4, 1), (b"R'G'B'A u16", 8, 2), (b"RGBA u32", 16, 4),
    (b"R'G'B'A float", 16, -4), (b"RGBA double", 32, -8), (b"RGBA half", 8, -99), (b"R'G'B' u15", 6, -99),
    (b"Y u8", 1, -99), (b"RGBA u64x", 40, -99),
])
def test_color_size(host, name, bpp, want):
    assert host.fixca_color_size(name, bpp) == want          # fix-ca.c:681-711


def test_fix_ca_fails_loudly_without_gpu(host):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    img = np.zeros((16, 32, 4), np.uint8)
    d = Drawable(img.ctypes.data, 32, 16, 4, 1, 0, 0, 32, 16)
    p = fixca.make_params(blue=1, lens_x=16, lens_y=8)
    assert host.fix_ca(ctypes.byref(d), ctypes.byref(p)) == -1
    d.bpc = -99
    assert host.fix_ca(ctypes.byref(d), ctypes.byref(p)) == -1          # "Invalid color type!"
    d.bpc, d.sel_w = 1, 0
    assert host.fix_ca(ctypes.byref(d), ctypes.byref(p)) == -1          # empty selection


def test_preview_pixels_arguments_and_no_gpu(host):
    """fixca_preview_pixels (preview_update's pixel work, fix-ca.c:647-671): argument errors are reported
    without a GPU; with valid arguments and no GPU the call fails loudly (-1, E_NODEVICE)"""
    host.fixca_preview_pixels.argtypes = [ctypes.POINTER(Drawable), ctypes.POINTER(fixca.FixCaParams)] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    img = np.zeros((16, 32, 4), np.uint16)
    prev = np.zeros((8, 10, 4), np.uint8)
    d = Drawable(img.ctypes.data, 32, 16, 8, 2, 0, 0, 32, 16)
    p = fixca.make_params(blue=1, lens_x=16, lens_y=8)
    assert host.fixca_preview_pixels(ctypes.byref(d), ctypes.byref(p), 30, 2, 10, 8, prev.ctypes.data) == -1     # window leaves the image
    assert b"outside" in fixca.lib.fixca_last_error()
    p.saturation = 20.0
    assert host.fixca_preview_pixels(ctypes.byref(d), ctypes.byref(p), 3, 2, 10, 8, prev.ctypes.data) == -1      # saturate(): refused
    assert b"saturation" in fixca.lib.fixca_last_error()
    p.saturation = 0.0
    d.bpc = -99
    assert host.fixca_preview_pixels(ctypes.byref(d), ctypes.byref(p), 3, 2, 10, 8, prev.ctypes.data) == -1
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d.bpc = 2
    assert host.fixca_preview_pixels(ctypes.byref(d), ctypes.byref(p), 3, 2, 10, 8, prev.ctypes.data) == -1
    assert fixca.lib.fixca_preview_host(img.ctypes.data, prev.ctypes.data, 32, 16, 8, 2, ctypes.byref(p), 3, 2, 10, 8) == fixca.E_NODEVICE


def test_cli_usage_and_no_gpu(tmp_path):
    assert subprocess.run([CLI], capture_output=True).returncode == 2
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    src = tmp_path / "in.ppm"
    src.write_bytes(b"P6\n8 8\n255\n" + bytes(range(192)))
    r = subprocess.run([CLI, str(src), str(tmp_path / "out.ppm"), "1.5", "-1.5"], capture_output=True)
    assert r.returncode == 1 and b"fix-ca (HIP)" in r.stderr
    assert not (tmp_path / "out.ppm").exists()
