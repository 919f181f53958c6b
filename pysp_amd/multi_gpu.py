"""Frame-parallel sharding over the GPUs of one node.

Frames (and HDR stacks) are independent, so the data path needs no collective: frame i goes to rank
i mod world.  The only exchange is the shared parameter block (white-balance multipliers + final
colour matrix), broadcast once per batch from the rank that owns the camera metadata -- 96 bytes,
latency bound; with backend "nccl" this is RCCL over xGMI, with "gloo" it runs on CPU (tests).
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


def broadcast_params(wb, M, src: int = 0, device=None):
    """Broadcast (wb, M) from `src` to every rank of the default process group; returns them on all ranks.
    Ranks other than `src` may pass None for both.  Without an initialised group it is the identity."""
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
