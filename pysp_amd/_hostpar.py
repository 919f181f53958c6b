"""Row-block threading for the few host-side NumPy passes the drop-in API keeps (the lens fields of corr_ca: they stay in NumPy so that
they round exactly like the reference's on the same machine -- `x ** 3` on a float32 array is the C library's / SVML's powf, not x*x*x).
Every expression is elementwise, NumPy releases the GIL inside its loops, so cutting an array into row blocks and evaluating the same expression
per block on a thread each gives the same bits, sooner.  PYSP_HOST_THREADS overrides the team size (1 = the serial code path)."""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np

_POOL: Optional[ThreadPoolExecutor] = None
_POOL_N = 0
_POOL_PID = 0                           # a forked child inherits the object but not its threads: it builds its own
_POOL_LOCK = threading.Lock()
_OLD_POOLS: list = []                  # pools of earlier team sizes stay alive: a concurrent pmap may still be submitting to one
MIN_PARALLEL_ELEMS = 1 << 18          # below this the serial pass is faster than the hand-off


def _cpu_share() -> int:
    env = os.environ.get("PYSP_HOST_THREADS")
    if env:
        try:
            return max(1, int(env))
        except ValueError:
            pass
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:                                                   # cgroup v2 CPU bandwidth limit: the share that may actually be used
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.999)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


_TEAM_CACHE = (None, 0)                 # (key the size was computed under, size)


def _affinity_len() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return -1


def team() -> int:
    """Team size; the cgroup limit is read once per (PYSP_HOST_THREADS, process, width of the affinity mask): a launcher that pins ranks later, or a forked
    child with a narrower mask, gets its own figure -- the mask width is one cheap system call, the cgroup file is not read again."""
    global _TEAM_CACHE
    key = (os.environ.get("PYSP_HOST_THREADS"), os.getpid(), _affinity_len())
    if _TEAM_CACHE[1] == 0 or _TEAM_CACHE[0] != key:
        _TEAM_CACHE = (key, _cpu_share())
    return _TEAM_CACHE[1]


def _pool(n: int) -> ThreadPoolExecutor:
    global _POOL, _POOL_N, _POOL_PID
    pid = os.getpid()
    with _POOL_LOCK:
        if _POOL is None or _POOL_N != n or _POOL_PID != pid:
            if _POOL is not None and _POOL_PID == pid:
                _OLD_POOLS.append(_POOL)            # never shut down under a caller (its idle threads cost nothing; team sizes change rarely)
            _POOL, _POOL_N, _POOL_PID = ThreadPoolExecutor(max_workers=n, thread_name_prefix="pysp-host"), n, pid
        return _POOL


def blocks(n_items: int, n_blocks: int) -> List[slice]:
    """n_items cut into at most n_blocks contiguous, near-equal slices (none empty)."""
    n_blocks = max(1, min(n_blocks, n_items))
    edges = [n_items * k // n_blocks for k in range(n_blocks + 1)]
    return [slice(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def pmap(fn: Callable, parts: Sequence) -> list:
    """[fn(p) for p in parts], the calls spread over the host team (in order; exceptions propagate)."""
    n = min(team(), len(parts))
    if n <= 1:
        return [fn(p) for p in parts]
    return list(_pool(team()).map(fn, parts))


def copy_into(dst: np.ndarray, src) -> np.ndarray:
    """np.copyto(dst, src) (same shape; casting as np.copyto's default), large arrays in row blocks on the host team; returns dst."""
    src = np.asarray(src)
    if dst.ndim == 0 or dst.size < MIN_PARALLEL_ELEMS or team() == 1 or src.shape != dst.shape:
        np.copyto(dst, src)
        return dst
    pmap(lambda sl: np.copyto(dst[sl], src[sl]), blocks(dst.shape[0], team()))
    return dst
