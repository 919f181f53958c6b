"""remove_ca_from_raw on the GPU (reference corr_ca/ca_removal.py:48-131).

Green is upsampled without cross-channel help, pulled onto the red (blue) geometry with the inverse lens model, used to
upsample red (blue) to full resolution, and that image is pushed back through the forward model and sampled at the
channel's own photosites.  All of the pixel work -- two bilinear remaps, the upsampling filters, (de)interleaving -- is
one library call; the host only evaluates the two coordinate quadrants of each lens model.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import model as _model  # noqa: F401  (package import path parity)
from .. import _lib
from .model.generic import CaCorrectionModel, ReversibleModelMixin


def _fields(lens_model, probe: np.ndarray):
    if lens_model is None:
        return None, None
    g_at_c = np.ascontiguousarray(lens_model.get_undistorted_quadrant(probe), dtype=np.float32)
    c_at_g = np.ascontiguousarray(lens_model.get_distorted_quadrant(probe), dtype=np.float32)
    return g_at_c, c_at_g


def remove_ca_from_raw(raw, lens_model_r: Optional[CaCorrectionModel], lens_model_b: Optional[CaCorrectionModel]):
    """Overwrites `raw.sensor_scaled` with the mosaic whose red and blue samples are aligned to green."""
    if lens_model_r is None and lens_model_b is None:
        return
    if lens_model_r is not None and not isinstance(lens_model_r, ReversibleModelMixin):
        raise ValueError("Red lens model is not reversible so green cannot be re-aligned to remove error. Use a reversible model and try again.")
    if lens_model_b is not None and not isinstance(lens_model_b, ReversibleModelMixin):
        raise ValueError("Blue lens model is not reversible so green cannot be re-aligned to remove error. Use a reversible model and try again.")

    bayer = _lib.f32_private(raw.sensor_scaled)   # private copy, corrected in place by the library
    if bayer.ndim != 2 or bayer.shape[0] % 2 or bayer.shape[1] % 2:
        raise ValueError("expected a Bayer mosaic with even dimensions")
    H, W = bayer.shape
    wb = raw.cam_wb.get_reciprocal_multipliers()
    g_at_r, r_at_g = _fields(lens_model_r, bayer)
    g_at_b, b_at_g = _fields(lens_model_b, bayer)
    q = [None if a is None else _lib.ptr(a) for a in (g_at_r, r_at_g, g_at_b, b_at_g)]
    _lib.check(_lib.lib().pysp_remove_ca_f32(_lib.default_context().handle, _lib.ptr(bayer), H, W, q[0], q[1], float(np.float32(wb[0])),
                                             q[2], q[3], float(np.float32(wb[2]))))
    raw.sensor_scaled = bayer
