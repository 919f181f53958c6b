"""ctypes binding of libpysp_hip.so (C ABI: include/pysp_hip.h).

The HIP library is the product; there is no NumPy/CPU fallback.  If the shared object is missing
this module raises ImportError on first use; if no GPU is present, creating a context raises
RuntimeError.  Nothing under oracle/ is ever imported from here.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import sys
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYSP_HIP_LIB") or os.path.join(_HERE, "csrc", "libpysp_hip.so")   # override: kernel A/B experiments

PYSP_OK, PYSP_EBADARG, PYSP_ENOTIMPL, PYSP_EHIP, PYSP_ENOMEM = 0, -1, -2, -3, -4
QUALITY_DRAFT, QUALITY_FAST, QUALITY_BEST = 0, 1, 2

_f32p = ctypes.POINTER(ctypes.c_float)
_intp = ctypes.POINTER(ctypes.c_int)
_f64p = ctypes.POINTER(ctypes.c_double)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p
_int, _sz, _flt, _dbl = ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_double

_SIGNATURES = {
    "pysp_abi_version": (_int, []),
    "pysp_last_error": (ctypes.c_char_p, []),
    "pysp_device_count": (_int, []),
    "pysp_ctx_create": (_vp, [_int, _vp]),
    "pysp_ctx_destroy": (None, [_vp]),
    "pysp_ctx_sync": (_int, [_vp]),
    "pysp_ctx_set_lab_mode": (_int, [_vp, _int]),
    "pysp_ctx_get_lab_mode": (_int, [_vp]),
    "pysp_lab_cv410_lut": (_int, [_vp]),
    "pysp_ctx_set_lab_lut": (_int, [_vp, _vp]),
    "pysp_ctx_set_lab_layout": (_int, [_vp, _int]),
    "pysp_ctx_get_lab_layout": (_int, [_vp]),
    "pysp_ctx_lab_layout_in_use": (_int, [_vp]),
    "pysp_ctx_set_select_form": (_int, [_vp, _int]),
    "pysp_ctx_get_select_form": (_int, [_vp]),
    "pysp_ahd_stream_chunks": (_int, [_int, _int, _int, _vp, _int, _vp, _vp]),
    "pysp_ctx_get_lab_lut": (_int, [_vp, _vp]),
    "pysp_ctx_set_stream": (_int, [_vp, _vp]),
    "pysp_ctx_get_stream": (_vp, [_vp]),
    "pysp_lab_tables": (_int, [_f32p, _f32p]),
    "pysp_ctx_last_kernel_ms": (_int, [_vp, _f32p]),
    "pysp_ctx_set_kernel_timing": (_int, [_vp, _int]),
    "pysp_ctx_kernel_times": (_int, [_vp, _int, _f32p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_int)]),
    "pysp_dev_alloc": (_vp, [_vp, _sz]),
    "pysp_dev_free": (_int, [_vp, _vp]),
    "pysp_dev_upload": (_int, [_vp, _vp, _vp, _sz]),
    "pysp_dev_download": (_int, [_vp, _vp, _vp, _sz]),
    "pysp_host_alloc": (_vp, [_sz]),
    "pysp_host_free": (_int, [_vp]),
    "pysp_wb_scale_dev": (_int, [_vp, _vp, _sz, _f32p, _int, _vp]),
    "pysp_bayer_to_rgbg_f32": (_int, [_vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pysp_bayer_to_rgbg_u16": (_int, [_vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pysp_rgbg_to_bayer_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _vp]),
    "pysp_bayer_normalize_u16": (_int, [_vp, _vp, _int, _int, _f32p, _f32p, _vp]),
    "pysp_resample_g_f32": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "pysp_resample_channel_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "pysp_find_hot_pixels_f32": (_int, [_vp, _vp, _int, _int, _flt, _int, _vp, _vp, _vp, _vp]),
    "pysp_flat_field_f32": (_int, [_vp, _vp, _vp, _int, _int, _f32p, _int, _vp]),
    "pysp_build_map_f32": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "pysp_demosaic_f32": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _vp]),
    "pysp_demosaic_dev": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _vp]),
    "pysp_cam_to_rgb_f32": (_int, [_vp, _vp, _sz, _f64p, _int, _vp]),
    "pysp_cam_to_rgb_dev": (_int, [_vp, _vp, _sz, _f64p, _int, _vp]),
    "pysp_lin_srgb_to_srgb_f32": (_int, [_vp, _vp, _sz, _vp]),
    "pysp_lin_srgb_to_srgb_dev": (_int, [_vp, _vp, _sz, _vp]),
    "pysp_srgb_to_lin_srgb_f32": (_int, [_vp, _vp, _sz, _vp]),
    "pysp_wb_scale_f32": (_int, [_vp, _vp, _sz, _f32p, _int, _vp]),
    "pysp_pipeline_srgb_f32": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_pipeline_srgb_dev": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_pipeline_f32": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_pipeline_dev": (_int, [_vp, _vp, _int, _int, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_pipeline_batch_dev": (_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _int, _int, _int, _f32p, _f64p, _int, _int, _int, _int, ctypes.POINTER(ctypes.c_void_p)]),
    "pysp_pipeline_u16_f32": (_int, [_vp, _vp, _int, _int, _f32p, _f32p, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_pipeline_batch_f32": (_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _int, _int, _int, _f32p, _f64p, _int, _int, _int, _int, ctypes.POINTER(ctypes.c_void_p)]),
    "pysp_pipeline_batch_u16_f32": (_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _int, _int, _int, _f32p, _f32p, _f32p, _f64p, _int, _int, _int, _int, ctypes.POINTER(ctypes.c_void_p)]),
    "pysp_pipeline_u16_dev": (_int, [_vp, _vp, _int, _int, _f32p, _f32p, _f32p, _f64p, _int, _int, _int, _int, _vp]),
    "pysp_fuse_raw_f32": (_int, [_vp, ctypes.POINTER(_vp), _int, _int, _int, _f32p, _f32p, _int, _vp, _vp]),
    "pysp_fuse_raw_dev": (_int, [_vp, ctypes.POINTER(_vp), _int, _int, _int, _f32p, _f32p, _int, _vp, _vp]),
    "pysp_fuse_rgb_f32": (_int, [_vp, ctypes.POINTER(_vp), _int, _sz, _f32p, ctypes.POINTER(_int), _f32p, _f32p, _int, _f64p, _vp, _vp, _int]),
    "pysp_fuse_rgb_dev": (_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _int, _sz, _f32p, ctypes.POINTER(_int), _f32p, _f32p, _int, _f64p, _vp, _vp]),
    "pysp_warp_table_f32": (_int, [_vp, _flt, _flt, _flt, _flt, _flt, _flt, _int, _int, _flt, _flt, _flt, _vp, _vp]),
    "pysp_warp_rectilinear_f32": (_int, [_vp, _vp, _int, _int, _f64p, _int, _dbl, _dbl, _flt]),
    "pysp_warp_rectilinear_prior_f32": (_int, [_vp, _vp, _int, _int, _f64p, _int, _dbl, _dbl, _flt, _vp]),
    "pysp_remap_lanczos4_f32": (_int, [_vp, _vp, _int, _int, _vp, _vp, _vp]),
    "pysp_warp_rectilinear_dev": (_int, [_vp, _vp, _vp, _int, _int, _f64p, _int, _dbl, _dbl, _flt]),
    "pysp_remove_ca_f32": (_int, [_vp, _vp, _int, _int, _vp, _vp, _flt, _vp, _vp, _flt]),
    "pysp_remove_ca_dev": (_int, [_vp, _vp, _int, _int, _vp, _vp, _flt, _vp, _vp, _flt]),
    "pysp_warp_rectilinear_rows_dev": (_int, [_vp, _vp, _vp, _int, _int, _f64p, _int, _dbl, _dbl, _flt, _int, _int]),
    "pysp_warp_source_rows": (_int, [_vp, _int, _int, _f64p, _int, _dbl, _dbl, _flt, _int, _int, _intp, _intp]),
}

_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()
_tls = threading.local()


def _preload_hip_runtime() -> None:
    """Keep ONE HIP runtime in the process.  PyTorch's wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) and asks for it as 'libamdhip64.so'; if /opt/rocm's copy were mapped first a
    later `import torch` would map a second runtime.  Mapping torch's copy first (without importing
    torch) makes both libpysp_hip.so and torch resolve to the same object, in either order."""
    if os.environ.get("PYSP_HIP_RUNTIME", "torch") != "torch" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> ctypes.CDLL:
    """Load libpysp_hip.so (once) and type its entry points."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with `make -C pysp_amd/csrc` (or __graft_entry__.build()). "
                    "pysp_amd has no CPU fallback.")
            _preload_hip_runtime()
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(L, name)      # AttributeError here = ABI mismatch, fail loudly
                fn.restype = res
                fn.argtypes = args
            if L.pysp_abi_version() != 1:
                raise ImportError("libpysp_hip.so ABI version mismatch")
            _lib = L
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def last_error() -> str:
    return lib().pysp_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map C return codes to the exceptions the reference raises."""
    if rc == PYSP_OK:
        return
    msg = last_error()
    if rc == PYSP_EBADARG:
        raise ValueError(msg)
    if rc == PYSP_ENOTIMPL:
        raise NotImplementedError(msg)
    if rc == PYSP_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


class Context:
    """One HIP stream + grow-only device workspace (pysp_ctx).  The C context is not thread-safe: one per thread for compute calls.
    `lock` serialises the buffer calls (pysp_dev_alloc / free / upload / download) that lazy arrays make on it -- also from another
    thread or from the garbage collector.  A context is a handle to a live C object in this process: it cannot be copied or pickled."""

    def __init__(self, device: int = 0, stream: int = 0):
        self.lock = threading.RLock()
        self._h = lib().pysp_ctx_create(int(device), ctypes.c_void_p(stream or None))
        if not self._h:
            raise RuntimeError(f"pysp_ctx_create failed: {last_error()}")
        self.device = int(device)
        if os.environ.get("PYSP_LAB_MODE", "") in ("0", "closed_form"):      # default is 1, the OpenCV 4.10 LUT path
            self.set_lab_mode(0)
        if os.environ.get("PYSP_LAB_LAYOUT", "") in ("0", "1", "packed", "planes"):
            self.set_lab_layout({"0": 0, "1": 1}.get(os.environ["PYSP_LAB_LAYOUT"], os.environ["PYSP_LAB_LAYOUT"]))

    @property
    def handle(self):
        return ctypes.c_void_p(self._h)

    def sync(self) -> None:
        check(lib().pysp_ctx_sync(self.handle))

    def set_lab_mode(self, mode) -> None:
        """AHD's Lab restatement: 1 / "cv410_lut" (default: OpenCV 4.10's LUT + trilinear path) or 0 / "closed_form"."""
        mode = {"closed_form": 0, "cv410_lut": 1, "cv410": 1}.get(mode, mode)
        check(lib().pysp_ctx_set_lab_mode(self.handle, int(mode)))

    def get_lab_mode(self) -> int:
        return int(lib().pysp_ctx_get_lab_mode(self.handle))

    def set_lab_layout(self, layout) -> None:
        """Lab mode 1 inside the AHD select kernel: -1 / "auto" (default: packed, switching to planes while the content keeps sending waves through the float
        form of the vote), 0 / "packed" (integer chroma votes, fastest on ordinary content) or 1 / "planes" (float votes: the same speed on any content, e.g.
        synthetic colour noise).  Same results either way."""
        layout = {"auto": -1, "packed": 0, "planes": 1}.get(layout, layout)
        check(lib().pysp_ctx_set_lab_layout(self.handle, int(layout)))

    def set_select_form(self, form) -> None:
        """Form of the AHD select kernel: 0 / "tile" (one 28x28 px tile per workgroup) or 1 / "stream" (persistent workgroups walking down the columns with
        carried Lab rows and votes).  Same results either way."""
        form = {"tile": 0, "stream": 1}.get(form, form)
        check(lib().pysp_ctx_set_select_form(self.handle, int(form)))

    def get_select_form(self) -> int:
        return int(lib().pysp_ctx_get_select_form(self.handle))

    def get_lab_layout(self) -> int:
        return int(lib().pysp_ctx_get_lab_layout(self.handle))

    def lab_layout_in_use(self) -> int:
        """0 (packed) or 1 (planes): what the next AHD call launches."""
        return int(lib().pysp_ctx_lab_layout_in_use(self.handle))

    def set_lab_lut(self, grid=None) -> None:
        """The 33^3 grid of lab mode 1 as data: a (33,33,33,3) int16 array ([B][G][R] node, (L, a, b) scaled as OpenCV's RGB2LabLUT_s16) recorded
        from real cv2 by tools/gen_cv2_goldens.py, or None for the built-in restatement."""
        if grid is None:
            with self.lock:
                check(lib().pysp_ctx_set_lab_lut(self.handle, None))
            return
        g = np.ascontiguousarray(grid, dtype=np.int16)
        if g.shape != (33, 33, 33, 3):
            raise ValueError("Lab grid must have shape (33, 33, 33, 3)")
        with self.lock:
            check(lib().pysp_ctx_set_lab_lut(self.handle, ctypes.c_void_p(g.ctypes.data)))

    def get_lab_lut(self) -> np.ndarray:
        out = np.empty((33, 33, 33, 3), np.int16)
        with self.lock:
            check(lib().pysp_ctx_get_lab_lut(self.handle, ctypes.c_void_p(out.ctypes.data)))
        return out

    def set_stream(self, stream: int) -> None:
        """Bind to a caller-owned hipStream_t (0 = the device's default stream), e.g. torch's current stream."""
        check(lib().pysp_ctx_set_stream(self.handle, ctypes.c_void_p(int(stream) or None)))

    def get_stream(self) -> int:
        return int(lib().pysp_ctx_get_stream(self.handle) or 0)

    def last_kernel_ms(self) -> float:
        ms = ctypes.c_float()
        check(lib().pysp_ctx_last_kernel_ms(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def set_kernel_timing(self, mode: int) -> None:
        """0: record no events; 1: one event pair per call (default); 2: plus one pair per kernel."""
        check(lib().pysp_ctx_set_kernel_timing(self.handle, int(mode)))

    def kernel_times(self):
        """[(kernel name, ms), ...] of the most recent demosaic/pipeline call (kernel timing must be on)."""
        ms = (ctypes.c_float * 8)()
        names = (ctypes.c_char_p * 8)()
        n = ctypes.c_int()
        check(lib().pysp_ctx_kernel_times(self.handle, 8, ms, names, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().pysp_ctx_destroy(ctypes.c_void_p(self._h))
            self._h = None

    def __copy__(self):
        raise TypeError("a pysp Context owns a HIP stream and device memory: it cannot be copied (create a new Context)")

    def __deepcopy__(self, memo):
        raise TypeError("a pysp Context owns a HIP stream and device memory: it cannot be copied (create a new Context)")

    def __reduce__(self):
        raise TypeError("a pysp Context cannot be pickled: it is a handle to a live HIP stream of this process")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_device() -> int:
    """Device of the drop-in NumPy API: PYSP_DEVICE if set; else torch's current device when torch has already
    initialised the GPU in this process; else LOCAL_RANK modulo the device count (one rank per GPU); else 0."""
    env = os.environ.get("PYSP_DEVICE")
    if env is not None:
        return int(env)
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_initialized():
            return int(torch.cuda.current_device())
    except Exception:
        pass
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None:
        n = lib().pysp_device_count()
        return int(lr) % n if n > 0 else 0
    return 0


def default_context() -> Context:
    """Per-thread default context (device: `default_device`, fixed at first use)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = Context(default_device())
        _tls.ctx = ctx
    return ctx


# ---- small marshalling helpers -----------------------------------------------------------------
def ptr(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


def empty_f32(shape) -> np.ndarray:
    """np.empty(shape, float32) for results: large ones come from the page-locked pool (see _hostpool.py)."""
    from . import _hostpool
    return _hostpool.empty(shape, np.float32)


def empty(shape, dtype) -> np.ndarray:
    """np.empty(shape, dtype) for results the library writes: large ones come from the page-locked pool (no first-touch page faults on reuse)."""
    from . import _hostpool
    return _hostpool.empty(shape, np.dtype(dtype))


def f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def f32_private(a) -> np.ndarray:
    """A private C-contiguous float32 copy of `a` (np.array(a, float32, order="C", copy=True)) in a pooled buffer, copied on the host team."""
    from . import _hostpar
    a = np.asarray(a)
    return _hostpar.copy_into(empty_f32(a.shape), a)


def wb3(wb) -> "ctypes.Array":
    w = np.asarray(wb, dtype=np.float32).reshape(-1)
    if w.size < 3:
        raise ValueError("white-balance coefficients need at least 3 entries")
    return (ctypes.c_float * 3)(*[float(x) for x in w[:3]])


def mat9(M) -> "Optional[ctypes.Array]":
    if M is None:
        return None
    m = np.asarray(M, dtype=np.float64).reshape(-1)
    if m.size != 9:
        raise ValueError("colour matrix must be 3x3")
    return (ctypes.c_double * 9)(*[float(x) for x in m])
