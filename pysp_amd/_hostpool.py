"""Result arrays from page-locked host memory.

A 24 MP RGB result is 288 MB.  `np.empty` gives untouched pages: the download then pays ~20 ms of page faults on top of the
5 ms DMA, and a copy into pageable memory cannot overlap anything.  Large results are therefore ndarrays laid over blocks of
page-locked memory (`pysp_host_alloc`).  A block returns to this pool when the last ndarray (or view) over it is garbage
collected and is reused by the next result of the same size, so a steady stream of frames settles on a handful of blocks.
The arrays are ordinary writable float32 ndarrays (`OWNDATA` is False, `.base` is the block).

Limits: at most POOL_FREE_PER_SIZE idle blocks per size are kept and at most PINNED_CAP bytes may be handed out at once;
beyond that (a caller hoarding results) new results fall back to plain `np.empty`.  PYSP_PINNED_RESULTS=0 disables the pool.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref

import numpy as np

MIN_BYTES = 8 << 20
POOL_FREE_PER_SIZE = 2
PINNED_CAP = int(os.environ.get("PYSP_PINNED_CAP", str(6 << 30)))
_enabled = os.environ.get("PYSP_PINNED_RESULTS", "1") not in ("0", "")
_lock = threading.Lock()
_free: dict = {}          # nbytes -> [address, ...]
_out = 0                  # bytes currently handed out


def _release(addr: int, nbytes: int) -> None:
    global _out
    from . import _lib
    with _lock:
        _out -= nbytes
        lst = _free.setdefault(nbytes, [])
        if len(lst) < POOL_FREE_PER_SIZE:
            lst.append(addr)
            return
    try:
        _lib.lib().pysp_host_free(ctypes.c_void_p(addr))
    except Exception:
        pass


def empty(shape, dtype=np.float32) -> np.ndarray:
    global _out
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if not _enabled or n < MIN_BYTES:
        return np.empty(shape, dtype)
    from . import _lib
    addr = None
    with _lock:
        if _out + n > PINNED_CAP:
            return np.empty(shape, dtype)
        lst = _free.get(n)
        if lst:
            addr = lst.pop()
        _out += n
    if addr is None:
        addr = _lib.lib().pysp_host_alloc(ctypes.c_size_t(n))
        if not addr:
            with _lock:
                _out -= n
            return np.empty(shape, dtype)
    block = (ctypes.c_char * n).from_address(addr)
    weakref.finalize(block, _release, int(addr), n)          # runs when the last array over the block is gone
    return np.frombuffer(block, dtype=dtype).reshape(shape)


def trim() -> None:
    """Give every idle block back to the driver."""
    from . import _lib
    with _lock:
        blocks = [a for lst in _free.values() for a in lst]
        _free.clear()
    for a in blocks:
        _lib.lib().pysp_host_free(ctypes.c_void_p(a))
