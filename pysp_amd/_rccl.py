"""RCCL without torch: a thin ctypes binding of librccl.so for callers that, like pySP itself, have no torch dependency.

`pysp_amd.multi_gpu` runs its collectives through `torch.distributed` when the caller hands it torch tensors and a torch process group; a pure-NumPy / C caller
gets the same three exchanges of SURVEY.md section 8e through an `RcclGroup` of this module instead (one process per GPU, as everywhere in this package):

    g = RcclGroup(rank, world, ctx)                     # ctx: this rank's pysp_amd._lib.Context (its device, its stream)
    wb, M = g.broadcast_params(wb, M, src=0)            # the 96-byte parameter block (multi_gpu.broadcast_params)
    needs = g.all_gather_pairs(s0, s1)                  # the warp's source-row bounds of every rank (config 5)
    g.exchange_rows(dptr, row_bytes, plan)              # point-to-point row blocks (multi_gpu.plan_row_exchange) on raw device pointers
    g.allgather_bands(dptr, row_bytes, bands)           # the first-cut alternative

Rendezvous: rank 0 draws the ncclUniqueId and serves its 128 bytes over TCP on MASTER_ADDR:MASTER_PORT (the variables torch.distributed.run exports;
127.0.0.1 when unset) -- `exchange_unique_id` below, plain sockets, testable on CPU.  Everything is enqueued on the context's stream; the calls that return
host values wait for it.  xGMI is a full mesh: send/recv pairs of one `ncclGroupStart/End` run concurrently, each on its own link.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import socket
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

NCCL_UNIQUE_ID_BYTES = 128
ncclInt8, ncclInt64, ncclFloat32, ncclFloat64 = 0, 4, 7, 8


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]      # (c_ubyte, not c_char: a c_char array field reads back truncated at its first NUL)


_rccl: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """librccl.so: the copy that ships with torch's wheel when there is one (it is linked against the HIP runtime _lib.py maps first), else ROCm's."""
    global _rccl
    if _rccl is not None:
        return _rccl
    _lib.lib()                                          # one HIP runtime in the process, mapped before RCCL asks for it
    cands = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so"))
    except (ImportError, ValueError):
        pass
    cands += ["/opt/rocm/lib/librccl.so", "librccl.so"]
    err = None
    for c in cands:
        try:
            L = ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError as e:
            err = e
    else:
        raise ImportError(f"librccl.so not found ({err})")
    vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ncclGetErrorString.restype = ctypes.c_char_p
    L.ncclGetErrorString.argtypes = [i]
    for name, args in (("ncclGetUniqueId", [ctypes.POINTER(_UniqueId)]), ("ncclCommInitRank", [ctypes.POINTER(vp), i, _UniqueId, i]), ("ncclCommDestroy", [vp]),
                       ("ncclBroadcast", [vp, vp, sz, i, i, vp, vp]), ("ncclAllGather", [vp, vp, sz, i, vp, vp]), ("ncclSend", [vp, sz, i, i, vp, vp]),
                       ("ncclRecv", [vp, sz, i, i, vp, vp]), ("ncclGroupStart", []), ("ncclGroupEnd", [])):
        f = getattr(L, name)
        f.restype, f.argtypes = i, args
    _rccl = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"RCCL {what} failed: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def exchange_unique_id(rank: int, world: int, payload: Optional[bytes], addr: Optional[str] = None, port: Optional[int] = None, timeout: float = 120.0) -> bytes:
    """Rank 0 hands `payload` (the ncclUniqueId) to every other rank over TCP; every rank returns it.  Plain sockets: no GPU, no torch."""
    if world == 1:
        return payload
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else os.environ.get("PYSP_RCCL_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 17))
    if rank == 0:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _a = srv.accept()
                with conn:
                    conn.sendall(payload)
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                buf = b""
                while len(buf) < NCCL_UNIQUE_ID_BYTES:
                    chunk = c.recv(NCCL_UNIQUE_ID_BYTES - len(buf))
                    if not chunk:
                        raise ConnectionError("rank 0 closed the rendezvous socket early")
                    buf += chunk
                return buf
        except (ConnectionRefusedError, socket.timeout, ConnectionError):
            if time.time() > deadline:
                raise TimeoutError(f"rank {rank}: no RCCL rendezvous at {addr}:{port}")
            time.sleep(0.05)


class RcclGroup:
    """One RCCL communicator over `world` ranks; this rank's operations run on `ctx`'s stream (see module docstring)."""

    def __init__(self, rank: int, world: int, ctx: Optional["_lib.Context"] = None, addr: Optional[str] = None, port: Optional[int] = None):
        self.rank, self.world = int(rank), int(world)
        self.ctx = ctx or _lib.default_context()
        L = lib()
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        raw = exchange_unique_id(self.rank, self.world, ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else None, addr, port)
        if len(raw) != NCCL_UNIQUE_ID_BYTES:
            raise RuntimeError("RCCL rendezvous returned %d bytes instead of %d" % (len(raw), NCCL_UNIQUE_ID_BYTES))
        ctypes.memmove(ctypes.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = ctypes.c_void_p()
        _check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._scratch = None        # a small device block for the host-valued collectives

    def destroy(self) -> None:
        if getattr(self, "_comm", None):
            comm, self._comm = self._comm, None
            try:
                self.ctx.sync()
                lib().ncclCommDestroy(comm)
            except Exception:
                pass
        self._scratch = None

    def __del__(self):
        self.destroy()

    @property
    def stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(self.ctx.get_stream() or None)

    def _dev_scratch(self, nbytes: int):
        from .device_array import DeviceArray
        if self._scratch is None or self._scratch.nbytes < nbytes:
            self._scratch = DeviceArray(self.ctx, (max(64, (nbytes + 3) // 4),))
        return self._scratch

    # ---- the three exchanges of SURVEY.md section 8e
    def broadcast_params(self, wb, M, src: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """multi_gpu.broadcast_params over RCCL: the 12-double block from rank `src` to every rank (ranks other than `src` may pass None)."""
        from .multi_gpu import PARAM_DOUBLES, pack_params, unpack_params
        block = pack_params(wb, M) if self.rank == src else np.zeros(PARAM_DOUBLES)
        d = self._dev_scratch(PARAM_DOUBLES * 8)
        L = _lib.lib()
        with self.ctx.lock:
            _lib.check(L.pysp_dev_upload(self.ctx.handle, d.ptr, _lib.ptr(block), ctypes.c_size_t(block.nbytes)))
            _check(lib().ncclBroadcast(d.ptr, d.ptr, PARAM_DOUBLES, ncclFloat64, src, self._comm, self.stream), "ncclBroadcast")
            out = np.empty(PARAM_DOUBLES, np.float64)
            _lib.check(L.pysp_dev_download(self.ctx.handle, _lib.ptr(out), d.ptr, ctypes.c_size_t(out.nbytes)))
        return unpack_params(out)

    def all_gather_pairs(self, a: int, b: int) -> List[Tuple[int, int]]:
        """Every rank's (a, b) -- the source-row bounds of its warp band -- on every rank."""
        mine = np.array([a, b], dtype=np.int64)
        d = self._dev_scratch(16 * (self.world + 1))
        L = _lib.lib()
        send = ctypes.c_void_p(d.ptr.value + 16 * self.world)
        with self.ctx.lock:
            _lib.check(L.pysp_dev_upload(self.ctx.handle, send, _lib.ptr(mine), ctypes.c_size_t(16)))
            _check(lib().ncclAllGather(send, d.ptr, 2, ncclInt64, self._comm, self.stream), "ncclAllGather")
            out = np.empty(2 * self.world, np.int64)
            _lib.check(L.pysp_dev_download(self.ctx.handle, _lib.ptr(out), d.ptr, ctypes.c_size_t(out.nbytes)))
        return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(self.world)]

    def exchange_rows(self, dptr: int, row_bytes: int, plan: Sequence[Tuple[int, int, int, int]]) -> None:
        """multi_gpu.exchange_rows on a raw device pointer: rows [r0, r1) of the whole-frame buffer at `dptr` (row_bytes per row) travel src -> dst for every
        (src, dst, r0, r1) of `plan`, all pairs inside ONE ncclGroupStart/End (xGMI full mesh: every pair has its own link)."""
        L = lib()
        mine = [(s, d, r0, r1) for (s, d, r0, r1) in plan if self.rank in (s, d)]
        if not mine:
            return
        _check(L.ncclGroupStart(), "ncclGroupStart")
        try:
            for s, d, r0, r1 in mine:
                p = ctypes.c_void_p(int(dptr) + r0 * row_bytes)
                n = (r1 - r0) * row_bytes
                if s == self.rank:
                    _check(L.ncclSend(p, n, ncclInt8, d, self._comm, self.stream), "ncclSend")
                else:
                    _check(L.ncclRecv(p, n, ncclInt8, s, self._comm, self.stream), "ncclRecv")
        finally:
            _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def allgather_bands(self, dptr: int, row_bytes: int, bands: Sequence[Tuple[int, int]]) -> None:
        """multi_gpu.allgather_bands on a raw device pointer: one broadcast per band, in place (band b's rows from rank b)."""
        L = lib()
        for b, (b0, b1) in enumerate(bands):
            p = ctypes.c_void_p(int(dptr) + b0 * row_bytes)
            _check(L.ncclBroadcast(p, p, (b1 - b0) * row_bytes, ncclInt8, b, self._comm, self.stream), "ncclBroadcast")
