"""Sensor-range normalisation on the GPU (reference normalization.py:4-24)."""
import ctypes

import numpy as np

from . import _lib


def bayer_normalize(rgbg: np.ndarray, chan_black, chan_sat) -> np.ndarray:
    """clip(x - black_c, 0, sat_c) / sat_c per CFA site (r, g1, b, g2), float32 throughout.

    The reference receives Python-int lists from rawpy, so the arithmetic is float32; NumPy integer
    scalars would silently promote its result to float64 -- here they are treated like the lists.
    """
    if rgbg.dtype != np.uint16:
        raise ValueError("bayer_normalize expects the uint16 mosaic delivered by the raw decoder")
    src = np.ascontiguousarray(rgbg)
    H, W = src.shape
    black = (ctypes.c_float * 4)(*[float(chan_black[i]) for i in range(4)])
    sat = (ctypes.c_float * 4)(*[float(chan_sat[i]) for i in range(4)])
    out = _lib.empty_f32((H, W))
    _lib.check(_lib.lib().pysp_bayer_normalize_u16(_lib.default_context().handle, _lib.ptr(src), H, W, black, sat, _lib.ptr(out)))
    return out


def raw_to_rgb(rgbg: np.ndarray, chan_black, chan_sat, cam_wb, quality: int = _lib.QUALITY_BEST, postprocess_steps: int = 1, tail: int = 0) -> np.ndarray:
    """`bayer_normalize` + demosaic (+ colour) in one GPU pass over the raw uint16 mosaic: what
    `RawBayerDataFromRaw(...).demosaic(...)` computes between image.py:229 and :183, without ever materialising
    the float32 mosaic.  tail: 0 camera RGB (RawDemosaicData.image), 1 to_lin_srgb, 2 + lin_srgb_to_srgb."""
    from .colorize.transform import final_matrix
    if rgbg.dtype != np.uint16:
        raise ValueError("raw_to_rgb expects the uint16 mosaic delivered by the raw decoder")
    src = np.ascontiguousarray(rgbg)
    H, W = src.shape
    black = (ctypes.c_float * 4)(*[float(chan_black[i]) for i in range(4)])
    sat = (ctypes.c_float * 4)(*[float(chan_sat[i]) for i in range(4)])
    out = _lib.empty_f32((H, W, 3))
    _lib.check(_lib.lib().pysp_pipeline_u16_f32(_lib.default_context().handle, _lib.ptr(src), H, W, black, sat,
                                                _lib.wb3(cam_wb.get_reciprocal_multipliers()), _lib.mat9(final_matrix(cam_wb.get_matrix())),
                                                int(quality), 0, int(postprocess_steps), int(tail), _lib.ptr(out)))
    return out
