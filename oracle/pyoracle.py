"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE.  Import this only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  Nothing under gimp-fix-ca_amd/ may import it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class FixCaParams(ctypes.Structure):
    """Mirror of include/fixca_types.h (reference fix-ca.c:70-82), 80 bytes."""
    _fields_ = [
        ("blue", ctypes.c_double), ("red", ctypes.c_double),
        ("lens_x", ctypes.c_double), ("lens_y", ctypes.c_double),
        ("update_preview", ctypes.c_int), ("interpolation", ctypes.c_int),
        ("saturation", ctypes.c_double),
        ("x_blue", ctypes.c_double), ("x_red", ctypes.c_double),
        ("y_blue", ctypes.c_double), ("y_red", ctypes.c_double),
    ]


assert ctypes.sizeof(FixCaParams) == 80


def make_params(blue=0.0, red=0.0, lens_x=-1.0, lens_y=-1.0, interpolation=0,
                x_blue=0.0, x_red=0.0, y_blue=0.0, y_red=0.0, saturation=0.0):
    return FixCaParams(blue, red, lens_x, lens_y, 1, int(interpolation), saturation,
                       x_blue, x_red, y_blue, y_red)


def build(force=False):
    """make -C oracle (gcc only; a second or two)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "fixca_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


_libs = {}


def _load(omp=False):
    key = "omp" if omp else "st"
    if key not in _libs:
        build()
        lib = ctypes.CDLL(os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so"))
        args = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                ctypes.c_int, ctypes.c_int, ctypes.POINTER(FixCaParams),
                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.fixca_oracle_region.argtypes = args
        lib.fixca_oracle_region.restype = ctypes.c_int
        lib.fixca_oracle_region_mt.argtypes = args + [ctypes.c_int]
        lib.fixca_oracle_region_mt.restype = ctypes.c_int
        lib.fixca_oracle_preview.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8
        lib.fixca_oracle_preview.restype = ctypes.c_int
        _libs[key] = lib
    return _libs[key]


def region(src: np.ndarray, params: FixCaParams, bpc=1, y1=0, y2=None,
           show_progress=1, dst=None, threads=1):
    """Run the oracle on a packed H x W x C image (C = bytes per pixel / |bpc|
    samples).  Returns dst (rows outside [y1,y2) keep whatever dst held)."""
    assert src.flags["C_CONTIGUOUS"] and src.ndim == 3
    h, w = src.shape[0], src.shape[1]
    bytes_pp = src.strides[1]
    if y2 is None:
        y2 = h
    if dst is None:
        dst = np.zeros_like(src)
    lib = _load(omp=threads > 1)
    if threads > 1:
        rc = lib.fixca_oracle_region_mt(src.ctypes.data, dst.ctypes.data, w, h, bytes_pp, bpc,
                                        ctypes.byref(params), 0, w, y1, y2, show_progress, threads)
    else:
        rc = lib.fixca_oracle_region(src.ctypes.data, dst.ctypes.data, w, h, bytes_pp, bpc,
                                     ctypes.byref(params), 0, w, y1, y2, show_progress)
    if rc != 0:
        raise ValueError("oracle refused the call: rc=%d" % rc)
    return dst


def preview(img: np.ndarray, bpc, x, y, width, height):
    """The tail of preview_update() (fix-ca.c:659-671): crop the window and convert samples wider than
    8 bits to 8 bits.  `img` is the corrected image (H x W x C samples).  Returns height x width x C uint8
    (bpc == 1: the cropped bytes)."""
    assert img.flags["C_CONTIGUOUS"] and img.ndim == 3
    h, w = img.shape[0], img.shape[1]
    bytes_pp = img.strides[1]
    nch = bytes_pp // abs(bpc)
    out = np.zeros((height, width, nch), np.uint8)
    rc = _load().fixca_oracle_preview(img.ctypes.data, out.ctypes.data, w, h, bytes_pp, bpc, x, y, width, height)
    if rc != 0:
        raise ValueError("oracle refused the preview call: rc=%d" % rc)
    return out
