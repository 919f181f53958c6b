"""Sharding over the GPUs of one node.

Frame-parallel: frames (and HDR stacks) are independent, so the data path needs no collective: frame i
goes to rank i mod world.  The only exchange is the shared parameter block (white-balance multipliers +
final colour matrix), broadcast once per batch from the rank that owns the camera metadata -- 96 bytes,
latency bound; with backend "nccl" this is RCCL over xGMI, with "gloo" it runs on CPU (tests).

Band-parallel (one very large frame, BASELINE config 5): rank b demosaics band b from host rows that
already include the stencil halo (no GPU-to-GPU traffic), but the lens warp that follows reads source
rows of other bands.  Those rows -- and only those, bounded on the device from the warp polynomial --
are exchanged point to point (xGMI is a full mesh, so every pair has its own link); the first-cut
alternative, an all-gather of all bands, is kept for comparison and as the fallback for extreme warps.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

PARAM_DOUBLES = 12   # wb[3] (float32 values carried exactly in float64) + M[9]


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Indices of the frames rank `rank` processes: round-robin, disjoint, covering."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad rank/world/n_frames")
    return list(range(rank, n_frames, world))


def pack_params(wb: Sequence[float], M: np.ndarray) -> np.ndarray:
    out = np.empty(PARAM_DOUBLES, np.float64)
    out[:3] = np.asarray(wb, dtype=np.float32)[:3].astype(np.float64)
    out[3:] = np.asarray(M, dtype=np.float64).reshape(9)
    return out


def unpack_params(block: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    block = np.asarray(block, dtype=np.float64).reshape(PARAM_DOUBLES)
    return block[:3].astype(np.float32), block[3:].reshape(3, 3).copy()


def broadcast_params(wb, M, src: int = 0, device=None, group=None):
    """Broadcast (wb, M) from `src` to every rank of the default process group; returns them on all ranks.
    Ranks other than `src` may pass None for both.  Without an initialised group it is the identity.
    `group`: a pysp_amd._rccl.RcclGroup runs the broadcast over RCCL without torch (a NumPy / C caller's path)."""
    if group is not None and hasattr(group, "broadcast_params"):
        return group.broadcast_params(wb, M, src)
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_params(pack_params(wb, M))
    t = torch.zeros(PARAM_DOUBLES, dtype=torch.float64, device=device if device is not None else "cpu")
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(pack_params(wb, M)))
    dist.broadcast(t, src=src)
    return unpack_params(t.cpu().numpy())


def band_ranges(H: int, n_bands: int, halo: int) -> List[Tuple[int, int, int, int]]:
    """Intra-frame sharding for one large frame (BASELINE config 5): `n_bands` horizontal bands aligned to
    even rows.  Returns (y0, y1, r0, r1) per band: rows [y0, y1) are the band's output, rows [r0, r1) are
    what it reads -- the band plus `halo` rows on each side, clipped at the true image border (where the
    reference's own border rules apply).  The halo comes from the host-side input (overlapping uploads),
    so the demosaic needs no GPU-to-GPU exchange.  AHD needs halo >= 8 + 4 * postprocess_stages rows."""
    if H < 2 or H % 2 or n_bands < 1 or halo < 0 or halo % 2:
        raise ValueError("H and halo must be even, n_bands >= 1")
    n_bands = min(n_bands, H // 2)
    rows = H // 2
    out = []
    for b in range(n_bands):
        y0 = 2 * (rows * b // n_bands)
        y1 = 2 * (rows * (b + 1) // n_bands)
        out.append((y0, y1, max(0, y0 - halo), min(H, y1 + halo)))
    return out


# ---- band-parallel: row exchange between the demosaic and the warp of one frame -------------------

def plan_row_exchange(bands: Sequence[Tuple[int, int]], needs: Sequence[Tuple[int, int]]) -> List[Tuple[int, int, int, int]]:
    """Transfers (src_rank, dst_rank, r0, r1): rank `src` owns output rows bands[src] = [y0, y1) of the debayered frame,
    rank `dst` needs source rows needs[dst] = [s0, s1) for its warp band; every overlap with another rank's band is one
    contiguous row block.  Deterministic order (by dst, then src) -- every rank derives the same plan."""
    plan = []
    for dst, (s0, s1) in enumerate(needs):
        for src, (y0, y1) in enumerate(bands):
            if src == dst:
                continue
            r0, r1 = max(s0, y0), min(s1, y1)
            if r0 < r1:
                plan.append((src, dst, r0, r1))
    return plan


def exchange_rows(full, plan: Sequence[Tuple[int, int, int, int]], rank: int, group=None, via_host: bool = False):
    """Run `plan` on the whole-frame tensor `full` (rows first): sends this rank's rows, receives the others' in place.
    One batch of point-to-point operations (RCCL send/recv over xGMI with backend "nccl").  `via_host` stages the
    blocks through CPU tensors, which is what the gloo backend needs for device data (tests, one-GPU rehearsal)."""
    import torch
    import torch.distributed as dist
    ops, landing = [], []
    for src, dst, r0, r1 in plan:
        if src == rank:
            blk = full[r0:r1]
            ops.append(dist.P2POp(dist.isend, blk.cpu() if via_host else blk, dst, group))
        elif dst == rank:
            if via_host:
                buf = torch.empty(full[r0:r1].shape, dtype=full.dtype)
                landing.append((buf, r0, r1))
                ops.append(dist.P2POp(dist.irecv, buf, src, group))
            else:
                ops.append(dist.P2POp(dist.irecv, full[r0:r1], src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for buf, r0, r1 in landing:
        full[r0:r1].copy_(buf)
    return full


def allgather_bands(full, bands: Sequence[Tuple[int, int]], rank: int, group=None, via_host: bool = False):
    """First-cut exchange: every rank ends up with the whole debayered frame.  Equal bands: one all-gather straight into
    the frame buffer (band order = rank order = row order); ragged bands: one broadcast per band."""
    import torch
    import torch.distributed as dist
    sizes = {y1 - y0 for y0, y1 in bands}
    y0, y1 = bands[rank]
    if len(sizes) == 1 and bands[0][0] == 0 and bands[-1][1] == full.shape[0]:
        if via_host:
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, full[y0:y1].cpu(), group=group)
            full.copy_(host)
        else:
            mine = full[y0:y1].clone()          # the output must not alias the input
            dist.all_gather_into_tensor(full, mine, group=group)
        return full
    for b, (b0, b1) in enumerate(bands):
        if via_host:
            t = full[b0:b1].cpu()
            dist.broadcast(t, src=b, group=group)
            if b != rank:
                full[b0:b1].copy_(t)
        else:
            dist.broadcast(full[b0:b1], src=b, group=group)
    return full


def ahd_halo_rows(stages: int) -> int:
    """Rows of real neighbour data a band needs on each side for AHD with `stages` median post-process stages."""
    return 8 + 4 * max(0, int(stages))


PHASES = ("demosaic", "bound_allgather", "row_exchange", "warp")


class BandPlan:
    """Band geometry of one frame cut over `world` ranks for AHD(stages) + WarpRectilinear (BASELINE config 5)."""

    def __init__(self, H: int, W: int, world: int, rank: int, stages: int):
        bands4 = band_ranges(H, world, ahd_halo_rows(stages))
        if len(bands4) != world:
            raise ValueError(f"a {H}-row frame cannot be cut into {world} bands")
        self.H, self.W, self.world, self.rank, self.stages = H, W, world, rank, stages
        self.bands = [(b[0], b[1]) for b in bands4]
        self.y0, self.y1, self.r0, self.r1 = bands4[rank]


def demosaic_warp_banded_dev(pipe, sub, plan: BandPlan, wb, M, coeffs, centre, scale: float = 1.0, group=None, exchange: str = "needed",
                             via_host: bool = False, full=None, out=None, mark=None, stats=None):
    """Device-resident core of `demosaic_warp_banded`: `sub` holds mosaic rows [plan.r0, plan.r1) of the frame (the band
    plus its stencil halo) on the device.  `full` / `out` are optional reusable whole-frame (H,W,3) buffers.  `mark(i)`, if
    given, is called at the start of phase PHASES[i] and once more (i = 4) at the end (the benchmark records events there).
    `stats`, if a dict, receives what this rank's exchange moved (rows and bytes received / sent, number of transfers).
    Everything is enqueued on torch's current stream; the only host waits are the two the algorithm itself needs (the row
    bounds come back from the device, and the all-gathered bounds are read to build the exchange plan)."""
    from . import _lib
    torch = pipe.torch
    H, W, world, rank = plan.H, plan.W, plan.world, plan.rank
    y0, y1, r0 = plan.y0, plan.y1, plan.r0
    mark = mark or (lambda i: None)
    mark(0)
    full = torch.empty((H, W, 3), dtype=torch.float32, device=pipe.device) if full is None else full
    # The band's demosaic lands where its rows live in the whole-frame buffer (round 5: until then it went to a buffer of its own and the band was copied over:
    # 2 x 1.2 GB of traffic per 100 MP frame at N = 1, 0.4 ms of a 6.4 ms step).  Its halo rows [r0, y0) and [y1, r1) land there too: rows that belong to the
    # neighbouring bands and are NOT exact (a halo row lacks its own halo) -- the exchange below overwrites every one of them the warp reads (plan_row_exchange
    # covers needs[rank] minus the rank's own band; the all-gather overwrites everything), and nothing else reads them.
    pipe.demosaic(sub, wb, M, _lib.QUALITY_BEST, False, plan.stages, out=full[r0:plan.r1])
    mark(1)
    if world > 1:
        import torch.distributed as dist
        if exchange == "needed":
            s0, s1 = pipe.warp_source_rows(H, W, coeffs, centre, y0, y1, scale)
            mine = torch.tensor([s0, s1], dtype=torch.int64, device="cpu" if via_host else pipe.device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            needs = [(int(t[0]), int(t[1])) for t in every]
            mark(2)
            xfers = plan_row_exchange(plan.bands, needs)
            if stats is not None:
                rr = sum(r1 - r0 for s, d, r0, r1 in xfers if d == rank)
                rs = sum(r1 - r0 for s, d, r0, r1 in xfers if s == rank)
                stats.update(rows_received=rr, rows_sent=rs, bytes_received=rr * W * 12, bytes_sent=rs * W * 12,
                             transfers=sum(1 for s, d, _, _ in xfers if rank in (s, d)), needed_rows=needs[rank])
            exchange_rows(full, xfers, rank, group, via_host)
        elif exchange == "allgather":
            mark(2)
            if stats is not None:
                rr = H - (y1 - y0)
                stats.update(rows_received=rr, rows_sent=(y1 - y0) * (world - 1), bytes_received=rr * W * 12, bytes_sent=(y1 - y0) * (world - 1) * W * 12,
                             transfers=world - 1, needed_rows=None)
            allgather_bands(full, plan.bands, rank, group, via_host)
        else:
            raise ValueError("exchange must be 'needed' or 'allgather'")
    else:
        mark(2)
    mark(3)
    out = torch.empty_like(full) if out is None else out
    pipe.warp_rows(full, coeffs, centre, y0, y1, out, scale)
    mark(4)
    return out[y0:y1]


def demosaic_warp_banded(pipe, bayer_host: np.ndarray, wb, M, coeffs, centre, stages: int = 3, scale: float = 1.0,
                         rank: int = 0, world: int = 1, group=None, exchange: str = "needed", via_host: bool = False):
    """BASELINE config 5 across `world` GPUs: AHD(stages) of band `rank` of the frame, row exchange, WarpRectilinear of
    the same band (chan_distortion_corr.py:86-97 after debayer/ahd.py).  `pipe` is this rank's DevicePipeline.
    Returns (y0, y1, band) with band = rows [y0, y1) of the warped (H,W,3) frame on the device; the rows are identical
    to the same rows of `pipe.demosaic_warp` on the whole frame."""
    torch = pipe.torch
    H, W = bayer_host.shape
    plan = BandPlan(H, W, world, rank, stages)
    sub = torch.from_numpy(np.ascontiguousarray(bayer_host[plan.r0:plan.r1])).to(pipe.device)
    band = demosaic_warp_banded_dev(pipe, sub, plan, wb, M, coeffs, centre, scale, group, exchange, via_host)
    pipe.sync()
    return plan.y0, plan.y1, band


def demosaic_warp_banded_np(ctx, bayer_host: np.ndarray, wb, M, coeffs, centre, stages: int = 3, scale: float = 1.0, group=None, exchange: str = "needed"):
    """`demosaic_warp_banded` for a caller WITHOUT torch (pySP itself has no torch dependency): `ctx` is this rank's pysp_amd._lib.Context, `group` a
    pysp_amd._rccl.RcclGroup (None: one rank), device memory comes from the context (pysp_dev_alloc).  Same three steps -- AHD(stages) of the band from host
    rows with halo, exchange of the rows the warp needs (RCCL send/recv on raw device pointers), WarpRectilinear of the band -- and the same bits.
    Returns (y0, y1, band) with band = rows [y0, y1) of the warped frame as a float32 ndarray."""
    import ctypes
    from . import _lib
    from .device_array import DeviceArray
    rank, world = (group.rank, group.world) if group is not None else (0, 1)
    bay = _lib.f32c(bayer_host)
    H, W = bay.shape
    plan = BandPlan(H, W, world, rank, stages)
    L = _lib.lib()
    sub_h = np.ascontiguousarray(bay[plan.r0:plan.r1])
    sub = DeviceArray.from_host(ctx, sub_h)
    full, out = DeviceArray(ctx, (H, W, 3)), DeviceArray(ctx, (H, W, 3))
    row_bytes = W * 12
    # the band's demosaic lands where its rows live in the whole-frame buffer (its halo rows too: rows other ranks own, overwritten by the exchange if the warp needs them)
    _lib.check(L.pysp_demosaic_dev(ctx.handle, sub.ptr, plan.r1 - plan.r0, W, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 0, int(stages),
                                   ctypes.c_void_p(full.ptr.value + plan.r0 * row_bytes)))
    cf = np.ascontiguousarray(coeffs, dtype=np.float64)
    cptr = cf.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    if world > 1:
        if exchange == "needed":
            s0, s1 = ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(L.pysp_warp_source_rows(ctx.handle, H, W, cptr, cf.shape[0], float(centre[0]), float(centre[1]), float(scale), plan.y0, plan.y1, ctypes.byref(s0), ctypes.byref(s1)))
            needs = group.all_gather_pairs(s0.value, s1.value)
            group.exchange_rows(full.ptr.value, row_bytes, plan_row_exchange(plan.bands, needs))
        elif exchange == "allgather":
            group.allgather_bands(full.ptr.value, row_bytes, plan.bands)
        else:
            raise ValueError("exchange must be 'needed' or 'allgather'")
    _lib.check(L.pysp_warp_rectilinear_rows_dev(ctx.handle, full.ptr, out.ptr, H, W, cptr, cf.shape[0], float(centre[0]), float(centre[1]), float(scale), plan.y0, plan.y1))
    band = np.empty((plan.y1 - plan.y0, W, 3), np.float32)
    with ctx.lock:
        _lib.check(L.pysp_dev_download(ctx.handle, _lib.ptr(band), ctypes.c_void_p(out.ptr.value + plan.y0 * row_bytes), ctypes.c_size_t(band.nbytes)))
    for d in (sub, full, out):
        d.release()
    return plan.y0, plan.y1, band
