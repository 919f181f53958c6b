"""HDR exposure fusion on the GPU (reference raw_hdr.py:7-158).

`fuse_exposures_to_raw` implements the behaviour the reference intends: at HEAD it raises TypeError
because it constructs `RawRggbBayerData()` without arguments (raw_hdr.py:150).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .base_types.image_base import RawDemosaicData
from .image import RawRggbBayerData


def _ev_offsets(evs: List[float], target_ev: Optional[float]):
    if target_ev is None:
        target_ev = 0
        for ev in evs:
            target_ev += ev
        target_ev /= len(evs)
    else:
        assert target_ev > 0
    return target_ev, [2 ** (ev - target_ev) for ev in evs]


def fuse_exposures_to_raw(in_exposures: List[RawRggbBayerData], target_ev: Optional[float] = None) -> Optional[Tuple[RawRggbBayerData, np.ndarray]]:
    """Weighted merge of K aligned Bayer exposures into one HDR mosaic (+ per-pixel contribution count)."""
    if len(in_exposures) == 0:
        return None
    K = len(in_exposures)
    target_ev, ev_offsets = _ev_offsets([e.current_ev for e in in_exposures], target_ev)
    frames = [_lib.f32c(e.sensor_scaled) for e in in_exposures]
    H, W = frames[0].shape
    if any(f.shape != (H, W) for f in frames):
        raise ValueError("all exposures must share one shape")

    # Host part of raw_hdr.py:128-136: the per-pixel bias takes only four values per frame (one per CFA
    # site), so it is evaluated here with NumPy exactly as written there and passed through bit for bit.
    wb_coeff = in_exposures[0].cam_wb.get_reciprocal_multipliers()
    site_w = np.array([wb_coeff[0], wb_coeff[1], wb_coeff[2], wb_coeff[1]], dtype=np.float32)
    bias = np.stack([1.6 ** (-0.1 * np.abs(off * site_w)) for off in ev_offsets]).astype(np.float32)
    off32 = np.array(ev_offsets, dtype=np.float32)
    kmax = int(np.argmax(ev_offsets))

    fused = np.empty((H, W), np.float32)
    count = np.empty((H, W), np.int32)
    ptrs = (ctypes.c_void_p * K)(*[f.ctypes.data for f in frames])
    _lib.check(_lib.lib().pysp_fuse_raw_f32(_lib.default_context().handle, ptrs, K, H, W,
                                            off32.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            np.ascontiguousarray(bias).ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            kmax, _lib.ptr(fused), _lib.ptr(count)))

    first = in_exposures[0]
    hdr_image = RawRggbBayerData(fused, first.cam_wb.copy(), target_ev, max(ev_offsets), first.source_pattern)
    hdr_image.set_hdr(True)
    return (hdr_image, count)


def fuse_exposures_from_debayer(in_exposures: List[RawDemosaicData], target_ev: Optional[float] = None):
    """Camera-space fusion of debayered exposures (raw_hdr.py:7-83): SURVEY.md 8(f) rank 2, not built yet."""
    raise NotImplementedError("fuse_exposures_from_debayer is scheduled after the section-8 rows (SURVEY.md 8f rank 2)")
