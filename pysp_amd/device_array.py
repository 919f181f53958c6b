"""Lazily materialised arrays: results of the drop-in classes stay in HBM until somebody reads them.

`RawRggbBayerData.demosaic(...)` leaves the demosaiced image on the GPU, `RawDemosaicData.to_lin_srgb()` returns a
`DeviceArray`, and `lin_srgb_to_srgb(DeviceArray)` downloads once at the end -- the README recipe (README.md:55-63 of the
reference) costs one upload, its kernels and one download instead of three round trips.

A DeviceArray quacks like the float32 ndarray the reference returns: `np.asarray(x)` (or any NumPy function, arithmetic,
indexing, item assignment, in-place arithmetic, attribute access) materialises it once.  From that moment the HOST copy is the
array: the device copy is released, so an edit through the ndarray (`a = np.asarray(x); a *= k`, `x[mask] = v`) is what every later
call sees, exactly as with the reference's ndarray (a later GPU call uploads the host copy again).  It is NOT an ndarray instance;
code that needs `isinstance(x, np.ndarray)` calls `np.asarray(x)` first, or switches laziness off with `pysp_amd.set_lazy(False)`
(environment: PYSP_EAGER=1), after which every call returns plain ndarrays exactly like the reference.
`copy.copy`, `copy.deepcopy` and `pickle` give a plain ndarray with the same values (a device pointer means nothing in a copy or
in another process).  What is thread-safe: the BUFFER calls on the owning context (allocate, free, upload, download, and the whole
download-and-release step of `numpy()`) are serialised by the context's re-entrant lock, so an array may be materialised, or garbage collected,
on another thread than the one that made it, also by two threads at once.  What is not: compute calls (demosaic, colour, fusion ...) on one
context from several threads at a time -- the C context owns one stream and one workspace; use one Context per thread for that.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

from . import _lib

_lazy = os.environ.get("PYSP_EAGER", "0") in ("", "0")
_deferred = os.environ.get("PYSP_LAZY", "") == "deferred"


def set_lazy(on) -> None:
    """True (default): results stay in HBM until read.  False: every call returns plain ndarrays, like the reference.  "deferred": as True, and a demosaic of a
    host mosaic is not even started until its result (or what to_lin_srgb() / lin_srgb_to_srgb() make of it) is needed -- the README recipe then runs as ONE
    banded host call whose upload, kernels and download overlap (5.9 ms instead of 7.7 ms at 24 MP).  The price, and why it is opt-in: until then the result
    REFERS to the caller's mosaic (`sensor_scaled`), so that array must not be modified in between (the reference computes eagerly and would not see the edit)."""
    global _lazy, _deferred
    _deferred = on == "deferred"
    _lazy = bool(on)


def lazy_enabled() -> bool:
    return _lazy


def deferred_enabled() -> bool:
    return _lazy and _deferred


class DeviceArray:
    """A C-contiguous float32 array that lives in a device buffer of a pysp context (see module docstring)."""

    __array_priority__ = 100.0

    def __init__(self, ctx: "_lib.Context", shape: Tuple[int, ...], dptr: Optional[int] = None):
        self._ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(np.float32)
        self._host: Optional[np.ndarray] = None
        self._keepalive = None          # host data an enqueued upload / kernel chain still reads; dropped at the first synchronising call
        n = self.nbytes
        if dptr is None:
            with ctx.lock:
                dptr = _lib.lib().pysp_dev_alloc(ctx.handle, ctypes.c_size_t(n))
            if not dptr:
                raise MemoryError(_lib.last_error())
        self._ptr: Optional[int] = int(dptr)

    # ---- construction / destruction
    @classmethod
    def from_host(cls, ctx, a: np.ndarray) -> "DeviceArray":
        a = _lib.f32c(a)
        out = cls(ctx, a.shape)
        with ctx.lock:
            _lib.check(_lib.lib().pysp_dev_upload(ctx.handle, out.ptr, _lib.ptr(a), ctypes.c_size_t(a.nbytes)))
        out._keepalive = a          # the upload is enqueued: the source must outlive it (released at the first synchronising call)
        return out

    def release(self) -> None:
        if getattr(self, "_ptr", None):
            ptr, self._ptr = self._ptr, None
            try:
                with self._ctx.lock:
                    _lib.lib().pysp_dev_free(self._ctx.handle, ctypes.c_void_p(ptr))
            except Exception:
                pass

    def __del__(self):
        self.release()

    # ---- device side
    @property
    def ptr(self) -> ctypes.c_void_p:
        if not self._ptr:
            raise ValueError("the device copy of this array has been released")
        return ctypes.c_void_p(self._ptr)

    @property
    def on_device(self) -> bool:
        return bool(self._ptr)

    @property
    def context(self):
        return self._ctx

    # ---- host side
    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self) -> int:
        return self.size * 4

    def numpy(self) -> np.ndarray:
        """The host copy: downloaded on first use, and from then on THE array -- the device copy is released, because the caller now
        holds a writable ndarray whose edits the GPU copy would not see (the reference hands out plain ndarrays)."""
        if self._host is None:
            out = _lib.empty_f32(self.shape)            # outside the lock: the page-locked pool has its own
            with self._ctx.lock:                        # (re-entrant) check, download and release are ONE step: two threads reading the same
                if self._host is None:                  # array get the same ndarray, and nobody downloads from a block already back in the cache
                    _lib.check(_lib.lib().pysp_dev_download(self._ctx.handle, _lib.ptr(out), self.ptr, ctypes.c_size_t(out.nbytes)))
                    self._host = out
                    self._keepalive = None
                    self.release()
        return self._host

    # a copy / pickle is a plain ndarray: a device pointer must never exist twice, nor travel to another process
    def __copy__(self):
        return self.numpy().copy()

    def __deepcopy__(self, memo):
        return self.numpy().copy()

    def __reduce__(self):
        return (np.array, (self.numpy(),))

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, k):
        return self.numpy()[k]

    def __setitem__(self, k, v):
        self.numpy()[k] = v.numpy() if isinstance(v, DeviceArray) else v

    def __iter__(self):
        return iter(self.numpy())

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, float32, {'on device' if self._ptr else 'released'}{', host copy cached' if self._host is not None else ''})"

    def __getattr__(self, name):        # anything else an ndarray offers (astype, reshape, mean, flags, T, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)


class DeferredImage(DeviceArray):
    """The (H, W, 3) result of a demosaic of a HOST mosaic that has not run yet (pysp_amd.set_lazy("deferred")): it remembers the mosaic (by reference), the
    white balance, matrix, quality, HDR flag, median stages and a colour tail (0 camera RGB, 1 to_lin_srgb, 2 + lin_srgb_to_srgb).  Needed on the host
    (numpy(), np.asarray, .image, lin_srgb_to_srgb) it runs the whole chain as one banded host call (pysp_pipeline_f32: upload || kernels || download);
    needed on the device (wb_undo(), a fusion, another colourspace) it uploads and runs the kernels there and from then on is an ordinary DeviceArray."""

    def __init__(self, ctx, mosaic: np.ndarray, wb, M, quality: int, hdr: bool, stages: int, tail: int = 0):
        H, W = mosaic.shape
        self._ctx = ctx
        self.shape = (int(H), int(W), 3)
        self.dtype = np.dtype(np.float32)
        self._host = None
        self._keepalive = None
        self._ptr = None
        self._plan = {"mosaic": mosaic, "wb": np.asarray(wb, dtype=np.float32).copy(), "M": None if M is None else np.array(M, dtype=np.float64).reshape(3, 3),
                      "quality": int(quality), "hdr": bool(hdr), "stages": int(stages), "tail": int(tail)}

    @property
    def pending(self) -> bool:
        return self._plan is not None

    @property
    def plan(self):
        return self._plan

    def with_tail(self, tail: int, M) -> "DeferredImage":
        """The same pending demosaic followed by colour tail `tail` with matrix M (a NEW array: this one keeps standing for what it stood for)."""
        p = self._plan
        return DeferredImage(self._ctx, p["mosaic"], p["wb"], M, p["quality"], p["hdr"], p["stages"], tail)

    def _args(self):
        p = self._plan
        H, W = p["mosaic"].shape
        return H, W, _lib.wb3(p["wb"]), _lib.mat9(p["M"]), p["quality"], int(p["hdr"]), p["stages"], p["tail"]

    @property
    def on_device(self) -> bool:
        return self._plan is not None or bool(self._ptr)

    @property
    def ptr(self) -> ctypes.c_void_p:
        if self._plan is not None:                      # somebody needs it in HBM: upload, run, become an ordinary device array
            with self._ctx.lock:
                if self._plan is not None:
                    H, W, wb, M, q, hdr, st, tail = self._args()
                    src = DeviceArray.from_host(self._ctx, self._plan["mosaic"])
                    dptr = _lib.lib().pysp_dev_alloc(self._ctx.handle, ctypes.c_size_t(self.nbytes))
                    if not dptr:
                        raise MemoryError(_lib.last_error())
                    try:
                        if tail == 0:
                            _lib.check(_lib.lib().pysp_demosaic_dev(self._ctx.handle, src.ptr, H, W, wb, M, q, hdr, st, ctypes.c_void_p(int(dptr))))
                        else:
                            _lib.check(_lib.lib().pysp_pipeline_dev(self._ctx.handle, src.ptr, H, W, wb, M, q, hdr, st, tail, ctypes.c_void_p(int(dptr))))
                    except Exception:
                        _lib.lib().pysp_dev_free(self._ctx.handle, ctypes.c_void_p(int(dptr)))       # still pending: the next reader tries again (or sees the same error)
                        raise
                    finally:
                        src.release()
                    self._ptr = int(dptr)
                    self._keepalive = self._plan["mosaic"]
                    self._plan = None
        return super().ptr

    def numpy(self) -> np.ndarray:
        if self._host is None and self._plan is not None:
            out = _lib.empty_f32(self.shape)
            with self._ctx.lock:
                if self._host is None and self._plan is not None:
                    H, W, wb, M, q, hdr, st, tail = self._args()
                    _lib.check(_lib.lib().pysp_pipeline_f32(self._ctx.handle, _lib.ptr(self._plan["mosaic"]), H, W, wb, M, q, hdr, st, tail, _lib.ptr(out)))
                    self._host = out
                    self._plan = None
        return self._host if self._host is not None else super().numpy()       # (realised on the device by another reader in the meantime: the ordinary download)

    def release(self) -> None:
        self._plan = None
        super().release()

    def __repr__(self):
        return f"DeferredImage(shape={self.shape}, float32, {'pending: ' + str({k: v for k, v in self._plan.items() if k not in ('mosaic', 'M', 'wb')}) if self._plan else 'realised'})"


def _delegate(name):
    def op(self, *args):
        return getattr(self.numpy(), name)(*[a.numpy() if isinstance(a, DeviceArray) else a for a in args])
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "pow", "mod", "matmul", "and", "or", "xor", "lshift", "rshift"):
    setattr(DeviceArray, f"__{_n}__", _delegate(f"__{_n}__"))
    setattr(DeviceArray, f"__r{_n}__", _delegate(f"__r{_n}__"))
for _n in ("neg", "pos", "abs", "lt", "le", "gt", "ge", "eq", "ne", "bool", "float", "int"):
    setattr(DeviceArray, f"__{_n}__", _delegate(f"__{_n}__"))


def _inplace(name):
    def op(self, other):            # x *= k acts on the host copy (the array from then on) and keeps the object
        getattr(self.numpy(), name)(other.numpy() if isinstance(other, DeviceArray) else other)
        return self
    op.__name__ = name
    return op


for _n in ("iadd", "isub", "imul", "itruediv", "ifloordiv", "ipow", "imod", "iand", "ior", "ixor"):
    setattr(DeviceArray, f"__{_n}__", _inplace(f"__{_n}__"))
DeviceArray.__hash__ = None


def as_device(x, ctx=None) -> DeviceArray:
    """x as a DeviceArray on `ctx` (default context): DeviceArrays of that context pass through, host data is uploaded."""
    ctx = ctx or _lib.default_context()
    if isinstance(x, DeviceArray) and x.on_device and x.context is ctx:
        return x
    return DeviceArray.from_host(ctx, np.asarray(x))
