"""Bayer plane split / merge on the GPU (reference bayer_chan_mixer.py:4-42)."""
from typing import Tuple

import numpy as np

from . import _lib


def bayer_to_rgbg(rgbg: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Split an RGGB mosaic into float32 R, G1 (top-right), B, G2 (bottom-left) quarter planes."""
    if rgbg.ndim != 2:
        raise ValueError("Bayer mosaic must be 2-D")
    H, W = rgbg.shape
    outs = [_lib.empty_f32((H // 2, W // 2)) for _ in range(4)]
    ctx = _lib.default_context()
    if rgbg.dtype == np.uint16:
        src = np.ascontiguousarray(rgbg)
        fn = _lib.lib().pysp_bayer_to_rgbg_u16
    else:
        src = _lib.f32c(rgbg)          # any other dtype goes through float32, like .astype(np.float32)
        fn = _lib.lib().pysp_bayer_to_rgbg_f32
    _lib.check(fn(ctx.handle, _lib.ptr(src), H, W, *[_lib.ptr(o) for o in outs]))
    return tuple(outs)


def rgbg_to_bayer(r: np.ndarray, g1, b, g2) -> np.ndarray:
    """Interleave four quarter planes back into one mosaic.  g1/b/g2 may be scalars (raw_hdr.py:130-133)."""
    r = _lib.f32c(r)
    planes = [r] + [_lib.f32c(np.broadcast_to(np.asarray(p, dtype=np.float32), r.shape)) for p in (g1, b, g2)]
    if any(p.shape != r.shape for p in planes):
        raise ValueError("quarter planes must share one shape")
    h, w = r.shape
    out = _lib.empty_f32((2 * h, 2 * w))
    _lib.check(_lib.lib().pysp_rgbg_to_bayer_f32(_lib.default_context().handle, *[_lib.ptr(p) for p in planes], h, w, _lib.ptr(out)))
    return out
