"""Stand-alone GPU forms of the public helpers of the reference's debayer/edge_assisted_gaussian.py
(:51-186); `debayer` itself is the fused kernel behind pysp_amd.debayer.debayer_eag."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .. import _lib
from .gaussian import BayerPatternPosition


def _plane(a) -> np.ndarray:
    p = _lib.f32c(a)
    if p.ndim != 2:
        raise ValueError("expected a 2-D plane")
    return p


def resample_g_to_full_resolution(g1: np.ndarray, g2: np.ndarray, use_bilinear_weighting: bool = True) -> np.ndarray:
    """Full-resolution green from the two green quarter planes of an RGGB mosaic, original samples untouched;
    missing sites are filled by the gradient-weighted (or plain) average of their four neighbours."""
    a, b = _plane(g1), _plane(g2)
    assert a.shape == b.shape
    h, w = a.shape
    out = _lib.empty_f32((2 * h, 2 * w))
    _lib.check(_lib.lib().pysp_resample_g_f32(_lib.default_context().handle, _lib.ptr(a), _lib.ptr(b), h, w, int(bool(use_bilinear_weighting)), _lib.ptr(out)))
    return out


def _pos(bayer_position: BayerPatternPosition) -> int:
    if bayer_position not in (BayerPatternPosition.TOP_LEFT, BayerPatternPosition.BOTTOM_RIGHT):
        raise NotImplementedError("only the TOP_LEFT (red) and BOTTOM_RIGHT (blue) bases occur in an RGGB mosaic")
    return bayer_position.value


def resample_channel(subpixel: np.ndarray, g_at_subpixel: np.ndarray, g_hf_pass: np.ndarray, bayer_position: BayerPatternPosition) -> np.ndarray:
    """Photosite-aware Gaussian upsampling of (channel - green) plus the upsampled green plus its high-pass."""
    s, g, hf = _plane(subpixel), _plane(g_at_subpixel), _plane(g_hf_pass)
    assert s.shape == g.shape
    h, w = s.shape
    if hf.shape != (2 * h, 2 * w):
        raise ValueError("g_hf_pass must be the full-resolution plane")
    out = _lib.empty_f32((2 * h, 2 * w))
    _lib.check(_lib.lib().pysp_resample_channel_f32(_lib.default_context().handle, _lib.ptr(s), _lib.ptr(g), _lib.ptr(hf), None, h, w,
                                                    _pos(bayer_position), _lib.ptr(out)))
    return out


def _resample_from_full(chan: np.ndarray, g_upscaled: np.ndarray, pos: int) -> np.ndarray:
    c, g = _plane(chan), _plane(g_upscaled)
    h, w = c.shape
    if g.shape != (2 * h, 2 * w):
        raise ValueError("g_upscaled must be twice the channel's size")
    out = _lib.empty_f32((2 * h, 2 * w))
    _lib.check(_lib.lib().pysp_resample_channel_f32(_lib.default_context().handle, _lib.ptr(c), None, None, _lib.ptr(g), h, w, pos, _lib.ptr(out)))
    return out


def resample_r(r: np.ndarray, g_upscaled: np.ndarray) -> np.ndarray:
    return _resample_from_full(r, g_upscaled, 0)


def resample_b(b: np.ndarray, g_upscaled: np.ndarray) -> np.ndarray:
    return _resample_from_full(b, g_upscaled, 3)


def resample_rb(r: np.ndarray, b: np.ndarray, g_upscaled: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    return (resample_r(r, g_upscaled), resample_b(b, g_upscaled))


def debayer(image):
    from . import debayer_eag
    return debayer_eag(image)
