"""Flat-field correction on the GPU (reference raw_correction.py:25-62).  The reference's dark- and
bias-frame functions are stubs that return a copy (:7-23) and are not reproduced."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def _plane_views(mosaic: np.ndarray):
    """The strided views bayer_chan_mixer.py:4-21 returns; np.mean is taken over exactly these so that the
    float32 pairwise summation runs in the reference's order."""
    if mosaic.dtype != np.float32:
        mosaic = mosaic.astype(np.float32)
    # views of the float32 mosaic itself: np.mean walks them in the same order as the reference's views of its row copies (checked bit for bit in
    # tests/test_host_wrappers_cpu.py), without copying the frame twice
    return mosaic[0::2, 0::2], mosaic[0::2, 1::2], mosaic[1::2, 1::2], mosaic[1::2, 0::2]


def flat_frame_correction(image, flat, clamp_high: bool = False):
    """In place on `image.sensor_scaled`: x * mean(flat_c) / flat per colour plane; a division by zero takes the
    plane's largest finite result, negative results clamp to zero, an all-black flat plane leaves the plane alone."""
    bayer = _lib.f32c(image.sensor_scaled)
    fl = _lib.f32c(flat.sensor_scaled)
    if bayer.shape != fl.shape:
        raise ValueError("image and flat frame must share one shape")
    H, W = bayer.shape
    means = (ctypes.c_float * 4)(*[float(np.mean(p)) for p in _plane_views(fl)])
    out = _lib.empty_f32(bayer.shape)
    _lib.check(_lib.lib().pysp_flat_field_f32(_lib.default_context().handle, _lib.ptr(bayer), _lib.ptr(fl), H, W, means, int(bool(clamp_high)), _lib.ptr(out)))
    image.sensor_scaled = out
