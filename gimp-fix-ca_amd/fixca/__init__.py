"""fixca — Python mirror of the Fix-CA plug-in's pixel-loop interface on top of
libfixca_hip.so (the C-ABI in include/fixca_hip.h).

The names and argument meaning follow the reference (fix-ca.c:70-82 FixCaParams,
fix-ca.c:998-1001 fix_ca_region): `fix_ca_region(src, dst, orig_width,
orig_height, bytes, bpc, params, x1, x2, y1, y2, show_progress)`.

  * numpy arrays  -> fixca_region_host   (H2D, kernel, D2H of rows [y1,y2))
  * torch CUDA tensors -> fixca_region_device on torch's current stream

There is no CPU implementation here.  If the shared library is missing the import
fails; if no GPU is present every compute call raises FixCaError.
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FIXCA_LIB") or os.path.join(os.path.dirname(_PKG), "libfixca_hip.so")   # FIXCA_LIB: timing-experiment builds

INTERP_NONE, INTERP_LINEAR, INTERP_CUBIC = 0, 1, 2
KERNEL_AUTO, KERNEL_EXACT, KERNEL_FAST = 0, 1, 2
PLAN_BATCH = 4            # planning hint OR-ed into `kernel`: one of many equal calls (fix_ca_batch sets it)
OK, E_INVALID, E_UNSUPPORTED, E_DOMAIN, E_HIP, E_NODEVICE = 0, -1, -2, -3, -4, -5
INPUT_MAX = 30


class FixCaError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("fixca error %d: %s" % (code, text))
        self.code = code


class FixCaParams(ctypes.Structure):
    """The plug-in's parameter block, 80 bytes (fix-ca.c:70-82)."""
    _fields_ = [
        ("blue", ctypes.c_double), ("red", ctypes.c_double),
        ("lens_x", ctypes.c_double), ("lens_y", ctypes.c_double),
        ("update_preview", ctypes.c_int), ("interpolation", ctypes.c_int),
        ("saturation", ctypes.c_double),
        ("x_blue", ctypes.c_double), ("x_red", ctypes.c_double),
        ("y_blue", ctypes.c_double), ("y_red", ctypes.c_double),
    ]

    FIELDS = ("blue", "red", "lens_x", "lens_y", "interpolation", "x_blue", "x_red", "y_blue", "y_red")

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}


assert ctypes.sizeof(FixCaParams) == 80


def default_params():
    """fix_ca_params_default (fix-ca.c:85-97)."""
    return FixCaParams(0.0, 0.0, -1.0, -1.0, 1, INTERP_LINEAR, 0.0, 0.0, 0.0, 0.0, 0.0)


def make_params(blue=0.0, red=0.0, lens_x=-1.0, lens_y=-1.0, interpolation=INTERP_NONE,
                x_blue=0.0, x_red=0.0, y_blue=0.0, y_red=0.0, saturation=0.0):
    """Non-interactive defaults of run() for omitted arguments (fix-ca.c:251-278)."""
    return FixCaParams(blue, red, lens_x, lens_y, 1, int(interpolation), saturation,
                       x_blue, x_red, y_blue, y_red)


def _load():
    # PyTorch (device memory, streams, torch.distributed) ships its own HIP + HSA runtime with
    # the same SONAME as /opt/rocm's.  Two runtimes in one process cannot both own the GPU, so
    # torch's copy is mapped first and libfixca_hip.so's NEEDED libamdhip64.so.7 binds to it.
    # A C host (the plug-in, fixca_cli) has no torch and uses /opt/rocm's runtime.
    import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise ImportError("libfixca_hip.so not built: run `make -C gimp-fix-ca_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    lib = ctypes.CDLL(_LIB_PATH)
    P = ctypes.POINTER(FixCaParams)
    vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
    sigs = {
        "fixca_region_host": ([vp, vp, ci, ci, ci, ci, P, ci, ci, ci, ci, ci], ci),
        "fixca_region_device": ([vp, vp, ci, ci, ci, ci, P, ci, ci, ci, ci, ci, cu, vp], ci),
        "fixca_batch_device": ([ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, P, cu, vp], ci),
        "fixca_plan_describe": ([ci, ci, ci, ci, P, ci, ci, ci, ci, ci, cu, ctypes.c_char_p, ctypes.c_size_t], ci),
        "fixca_params_check": ([P], ci),
        "fixca_device_count": ([], ci),
        "fixca_set_device": ([ci], ci),
        "fixca_dev_alloc": ([ctypes.c_size_t], vp),
        "fixca_dev_free": ([vp], ci),
        "fixca_host_alloc": ([ctypes.c_size_t], vp),
        "fixca_host_free": ([vp], ci),
        "fixca_copy_h2d": ([vp, vp, ctypes.c_size_t, vp], ci),
        "fixca_copy_d2h": ([vp, vp, ctypes.c_size_t, vp], ci),
        "fixca_stream_sync": ([vp], ci),
        "fixca_release_cache": ([], None),
        "fixca_prewarm": ([], None),
        "fixca_time_launches": ([ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ci, P, cu, vp,
                                 ctypes.POINTER(ctypes.c_double)], ci),
        "fixca_time_copy": ([ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ctypes.c_size_t, vp,
                             ctypes.POINTER(ctypes.c_double)], ci),
        "fixca_preview_host": ([vp, vp, ci, ci, ci, ci, P, ci, ci, ci, ci], ci),
        "fixca_preview_device": ([vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp], ci),
        "fixca_noise_device": ([vp, ctypes.c_size_t, ctypes.c_ulonglong, vp], ci),
        "fixca_debug_sched": ([ci, ci, ci, P, ci, ctypes.POINTER(ci)], ci),
        "fixca_debug_host_cache": ([ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)], None),
        "fixca_phase": ([ctypes.c_char_p], None),
        "fixca_last_error": ([], ctypes.c_char_p),
        "fixca_abi_version": ([], ci),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    return lib, tuple(sigs)


lib, ABI_SYMBOLS = _load()
LIB_PATH = _LIB_PATH


def _check(rc):
    if rc < 0:
        raise FixCaError(rc, lib.fixca_last_error().decode("utf-8", "replace"))
    return rc


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def params_check(params):
    """run()'s range check (fix-ca.c:279-295): True if a non-interactive call would be accepted."""
    return lib.fixca_params_check(ctypes.byref(params)) == OK


def plan(orig_width, orig_height, bytes, bpc, params, x1=0, x2=None, y1=0, y2=None,
         show_progress=True, kernel=KERNEL_AUTO):
    """(kernel id, description) the planner would choose; works without a GPU."""
    buf = ctypes.create_string_buffer(256)
    x2 = orig_width if x2 is None else x2
    y2 = orig_height if y2 is None else y2
    k = _check(lib.fixca_plan_describe(orig_width, orig_height, bytes, bpc, ctypes.byref(params),
                                       x1, x2, y1, y2, int(bool(show_progress)), kernel, buf, len(buf)))
    return k, buf.value.decode()


def fix_ca_region(src, dst, orig_width, orig_height, bytes, bpc, params,
                  x1, x2, y1, y2, show_progress=True, kernel=KERNEL_AUTO):
    """The reference's fix_ca_region (fix-ca.c:998-1348) on the GPU.

    src, dst: packed interleaved images of orig_height x orig_width x bytes, either
    both C-contiguous numpy uint8 arrays (host entry) or both torch CUDA uint8
    tensors (device entry, asynchronous on torch's current stream).  Rows of dst
    outside [y1, y2) are left untouched.  `kernel` is honoured by the device
    entry only; the host entry follows the FIXCA_EXACT environment variable.
    """
    nbytes = orig_width * orig_height * bytes
    if _is_torch(src) != _is_torch(dst):
        raise TypeError("src and dst must both be numpy arrays or both be torch tensors")
    if _is_torch(src):
        if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()):
            raise TypeError("torch tensors must be contiguous CUDA tensors")
        if src.numel() * src.element_size() != nbytes or dst.numel() * dst.element_size() != nbytes:
            raise ValueError("tensor size does not match orig_width*orig_height*bytes")
        _check(lib.fixca_region_device(src.data_ptr(), dst.data_ptr(), orig_width, orig_height, bytes, bpc,
                                       ctypes.byref(params), x1, x2, y1, y2, int(bool(show_progress)),
                                       kernel, _current_stream()))
        return dst
    if not (src.flags["C_CONTIGUOUS"] and dst.flags["C_CONTIGUOUS"] and dst.flags["WRITEABLE"]):
        raise TypeError("numpy arrays must be C-contiguous (dst writeable)")
    if src.nbytes != nbytes or dst.nbytes != nbytes:
        raise ValueError("array size does not match orig_width*orig_height*bytes")
    _check(lib.fixca_region_host(src.ctypes.data, dst.ctypes.data, orig_width, orig_height, bytes, bpc,
                                 ctypes.byref(params), x1, x2, y1, y2, int(bool(show_progress))))
    return dst


def fix_ca_image(img, params, kernel=KERNEL_AUTO, out=None):
    """Whole-image convenience: what fix_ca() does around fix_ca_region (fix-ca.c:364-374)
    for an H x W x C uint8 image (numpy or torch CUDA)."""
    h, w, c = img.shape
    if out is None:
        out = np.empty_like(img) if not _is_torch(img) else img.new_empty(img.shape)
    return fix_ca_region(img, out, w, h, c, 1, params, 0, w, 0, h, True, kernel)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def fix_ca_batch(srcs, dsts, orig_width, orig_height, bytes, bpc, params, kernel=KERNEL_AUTO):
    """A batch of independent images with one parameter block (device tensors),
    launched back to back on torch's current stream."""
    assert len(srcs) == len(dsts)
    _check(lib.fixca_batch_device(_ptr_array(srcs), _ptr_array(dsts), len(srcs), orig_width, orig_height,
                                  bytes, bpc, ctypes.byref(params), kernel, _current_stream()))
    return dsts


def preview(img, params, bpc, x, y, width, height):
    """preview_update()'s pixels (fix-ca.c:617-679) for a host image `img` (numpy, H x W x C samples):
    rows [y, y+height) corrected as the preview does (show_progress FALSE), the window cropped and
    samples wider than 8 bits converted to 8 bits.  Returns height x width x C uint8."""
    h, w, c = img.shape
    bytes_pp = img.strides[1]
    out = np.zeros((height, width, c), np.uint8) if abs(bpc) != 1 else np.zeros((height, width, bytes_pp), np.uint8)
    _check(lib.fixca_preview_host(img.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), w, h,
                                  bytes_pp, bpc, ctypes.byref(params), x, y, width, height))
    return out


def noise_fill(t, seed):
    """Fill the device tensor `t` (uint8, contiguous) with SURVEY §8d's noise(seed) on torch's current stream."""
    assert t.is_contiguous() and t.element_size() == 1
    _check(lib.fixca_noise_device(ctypes.c_void_p(t.data_ptr()), t.numel(), seed & 0xFFFFFFFFFFFFFFFF, _current_stream()))
    return t


SCHED_FIELDS = ("ncols", "rows", "pool_chunks", "pool_waves", "pool_blocks", "th")


def debug_sched(orig_width, orig_height, bytes, params, n_images):
    """k_roll's tile schedule for a batch launch of n_images, as a dict; works without a GPU.  Test hook."""
    out = (ctypes.c_int * 6)()
    _check(lib.fixca_debug_sched(orig_width, orig_height, bytes, ctypes.byref(params), n_images, out))
    return dict(zip(SCHED_FIELDS, list(out)))


def debug_host_cache():
    """(src image, dst image, bytes) of the calling thread's fixca_region_host device cache.  Test hook."""
    a, b, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
    lib.fixca_debug_host_cache(ctypes.byref(a), ctypes.byref(b), ctypes.byref(n))
    return a.value, b.value, n.value


def time_launches(srcs, dsts, launches, orig_width, orig_height, bytes, bpc, params, kernel=KERNEL_AUTO):
    """Mean ms per kernel launch, HIP events on torch's current stream (see fixca_hip.h)."""
    ms = ctypes.c_double()
    _check(lib.fixca_time_launches(_ptr_array(srcs), _ptr_array(dsts), len(srcs), launches, orig_width,
                                   orig_height, bytes, bpc, ctypes.byref(params), kernel, _current_stream(),
                                   ctypes.byref(ms)))
    return ms.value


def time_copy(srcs, dsts, launches, nbytes):
    ms = ctypes.c_double()
    _check(lib.fixca_time_copy(_ptr_array(srcs), _ptr_array(dsts), len(srcs), launches, nbytes,
                               _current_stream(), ctypes.byref(ms)))
    return ms.value


def pinned_empty(shape, dtype=np.uint8):
    """numpy array on page-locked host memory (fixca_host_alloc); free with pinned_free(arr)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib.fixca_host_alloc(n)
    if not ptr:
        raise FixCaError(E_HIP, lib.fixca_last_error().decode("utf-8", "replace"))
    buf = (ctypes.c_uint8 * n).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    return arr


def pinned_free(arr):
    _check(lib.fixca_host_free(ctypes.c_void_p(arr.ctypes.data)))


def device_count():
    return _check(lib.fixca_device_count())
