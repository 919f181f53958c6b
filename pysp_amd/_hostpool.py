"""Result arrays from page-locked host memory.

A 24 MP RGB result is 288 MB.  `np.empty` gives untouched pages: the download then pays ~20 ms of page faults on top of the
5 ms DMA, and a copy into pageable memory cannot overlap anything.  Large results are therefore ndarrays laid over blocks of
page-locked memory (`pysp_host_alloc`).  A block returns to this pool when the last ndarray (or view) over it is garbage
collected and is reused by the next result of the same size, so a steady stream of frames settles on a handful of blocks.
The arrays are ordinary writable float32 ndarrays (`OWNDATA` is False, `.base` is the block).

Limits: at most POOL_FREE_PER_SIZE idle blocks per size and IDLE_CAP idle bytes in all are kept (the least recently released go back to the
driver first), and handed-out plus idle bytes never exceed PINNED_CAP; beyond that (a caller hoarding results) new results fall back to plain
`np.empty`.  PYSP_PINNED_RESULTS=0 disables the pool.  The block finalizer can run inside a garbage-collection pass started while this
module holds its lock (same thread): the lock is re-entrant and the finalizer only queues the block; the queue is drained by the next call.
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading
import weakref

import numpy as np

MIN_BYTES = 8 << 20
POOL_FREE_PER_SIZE = 2
PINNED_CAP = int(os.environ.get("PYSP_PINNED_CAP", str(6 << 30)))
_enabled = os.environ.get("PYSP_PINNED_RESULTS", "1") not in ("0", "")
IDLE_CAP = int(os.environ.get("PYSP_PINNED_IDLE_CAP", str(2 << 30)))
_lock = threading.RLock()
_free: "collections.OrderedDict" = collections.OrderedDict()   # (nbytes, address) -> None, oldest release first
_idle = 0                 # bytes held idle in _free
_out = 0                  # bytes currently handed out
_returned: "collections.deque" = collections.deque()           # (address, nbytes) from finalizers (appending takes no lock)


def _release(addr: int, nbytes: int) -> None:
    """Finalizer of a block: may run at any allocation point of any thread (cyclic GC), so it only queues."""
    _returned.append((addr, nbytes))


def _drain() -> list:
    """Move finalized blocks into the pool (called with the lock held); returns the addresses that have to go back to the driver."""
    global _out, _idle
    drop = []
    while _returned:
        addr, n = _returned.popleft()
        _out -= n
        if sum(1 for (m, _a) in _free if m == n) < POOL_FREE_PER_SIZE and n <= IDLE_CAP:
            _free[(n, addr)] = None
            _idle += n
        else:
            drop.append(addr)
    while _idle > IDLE_CAP and _free:
        (n, addr), _ = _free.popitem(last=False)                # least recently released first
        _idle -= n
        drop.append(addr)
    return drop


def _give_back(addrs) -> None:
    from . import _lib
    for a in addrs:
        try:
            _lib.lib().pysp_host_free(ctypes.c_void_p(a))
        except Exception:
            pass


def empty(shape, dtype=np.float32) -> np.ndarray:
    global _out
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if not _enabled or n < MIN_BYTES:
        return np.empty(shape, dtype)
    from . import _lib
    global _idle
    addr = None
    with _lock:
        drop = _drain()
        key = next((k for k in reversed(_free) if k[0] == n), None)
        if key is not None:
            del _free[key]
            _idle -= n
            addr = key[1]
        else:
            while _out + _idle + n > PINNED_CAP and _free:       # idle blocks of OTHER sizes go first: pageable memory is the last resort
                (m, a), _ = _free.popitem(last=False)
                _idle -= m
                drop.append(a)
            if _out + n > PINNED_CAP:
                _give_back(drop)
                return np.empty(shape, dtype)
        _out += n
    _give_back(drop)
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
    global _idle
    with _lock:
        blocks = _drain() + [a for (_n, a) in _free]
        _free.clear()
        _idle = 0
    _give_back(blocks)
