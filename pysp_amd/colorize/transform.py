"""Camera RGB -> linear sRGB -> sRGB on the GPU (reference colorize/transform.py:6-111).

The 3x3 is assembled on the host in float64 exactly as transform.py:40-49 does; the per-pixel work
(clip, float64 dot, float32 rounding, gamma) runs in HIP kernels.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib
from ..device_array import DeferredImage, DeviceArray, lazy_enabled
from ..wb_cct.helpers_cam_mat import MatXyzToCamera
from .rgb_space import ArbitraryRgbColorspace, LinRgbColorspace


_FINAL_MATRIX_CACHE: dict = {}


def final_matrix(cam_xyz_matrix: MatXyzToCamera, destination_colorspace: ArbitraryRgbColorspace = LinRgbColorspace.REC709) -> np.ndarray:
    """inv(row-normalised(XYZ->cam @ RGB->XYZ adapted to the camera white)), float64 (transform.py:40-49).  The same nine doubles for the same camera matrix,
    white and colourspace: the last few results are kept (the recipe asks twice per frame -- AHD's homogeneity metric and to_lin_srgb -- and a 3x3 inverse plus
    the Bradford chain cost more host time than enqueueing the frame's kernels); a fresh copy is returned, as the reference returns a fresh array."""
    key = (np.asarray(cam_xyz_matrix.mat).tobytes(), np.asarray(cam_xyz_matrix.xyz).tobytes(), np.asarray(cam_xyz_matrix.mat).dtype.str, id(destination_colorspace))
    hit = _FINAL_MATRIX_CACHE.get(key)
    if hit is not None and hit[0] is destination_colorspace:
        return hit[1].copy()
    to_xyz = destination_colorspace.mat_to_xyz(cam_xyz_matrix.xyz.tolist())
    m = np.matmul(cam_xyz_matrix.mat, to_xyz)
    m = m / m.sum(axis=1)[:, np.newaxis]        # neutral in -> neutral out
    out = np.linalg.inv(m)
    if len(_FINAL_MATRIX_CACHE) >= 16:
        _FINAL_MATRIX_CACHE.pop(next(iter(_FINAL_MATRIX_CACHE)))
    _FINAL_MATRIX_CACHE[key] = (destination_colorspace, out.copy())
    return out


def _on_device(x) -> bool:
    return isinstance(x, DeviceArray) and x.on_device and lazy_enabled()


def _rgb_image(rgb: np.ndarray) -> np.ndarray:
    a = _lib.f32c(rgb)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) RGB image")
    return a


def clip_rgb(rgb: np.ndarray) -> np.ndarray:
    """Clip to [0,1] (transform.py:6-19).  Pure copy semantics; done on the host, it is never the bottleneck."""
    return np.clip(_rgb_image(rgb), 0, 1)


def cam_to_rgb_norm(rgb: np.ndarray, cam_xyz_matrix: MatXyzToCamera, destination_colorspace: ArbitraryRgbColorspace,
                    clip_highlights: bool = True) -> np.ndarray:
    M = final_matrix(cam_xyz_matrix, destination_colorspace)
    if isinstance(rgb, DeferredImage) and rgb.pending and lazy_enabled() and clip_highlights and rgb.plan["tail"] == 0 and \
            (rgb.plan["M"] is None or np.array_equal(rgb.plan["M"], M)):
        return rgb.with_tail(1, M)                   # deferred mode: the pending demosaic grows a colour tail, still nothing runs
    if _on_device(rgb):                              # intermediate of the README recipe: stays in HBM (lazy DeviceArray)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("expected an (H, W, 3) RGB image")
        out = DeviceArray(rgb.context, rgb.shape)
        _lib.check(_lib.lib().pysp_cam_to_rgb_dev(rgb.context.handle, rgb.ptr, rgb.size // 3, _lib.mat9(M), int(bool(clip_highlights)), out.ptr))
        return out
    a = _rgb_image(rgb)
    out = _lib.empty_f32(a.shape)
    _lib.check(_lib.lib().pysp_cam_to_rgb_f32(_lib.default_context().handle, _lib.ptr(a), a.size // 3, _lib.mat9(M),
                                              int(bool(clip_highlights)), _lib.ptr(out)))
    return out


def cam_to_lin_srgb(rgb: np.ndarray, cam_xyz_matrix: MatXyzToCamera, clip_highlights: bool = True) -> np.ndarray:
    return cam_to_rgb_norm(rgb, cam_xyz_matrix, LinRgbColorspace.REC709, clip_highlights)


def cam_to_clean_xyz(rgb: np.ndarray, cam_xyz_matrix: MatXyzToCamera, pcs_colorspace: ArbitraryRgbColorspace = LinRgbColorspace.REC2020,
                     clip_highlights: bool = True) -> np.ndarray:
    """transform.py:55-74: detinted working RGB, then that space's RGB->XYZ (second float64 dot)."""
    work = cam_to_rgb_norm(rgb, cam_xyz_matrix, pcs_colorspace, clip_highlights)
    out = _lib.empty_f32(work.shape)
    _lib.check(_lib.lib().pysp_cam_to_rgb_f32(_lib.default_context().handle, _lib.ptr(work), work.size // 3,
                                              _lib.mat9(pcs_colorspace.mat_to_xyz()), 0, _lib.ptr(out)))
    return out


def _flat(fn_name: str, x: np.ndarray) -> np.ndarray:
    a = _lib.f32c(x)
    out = _lib.empty_f32(a.shape)
    _lib.check(getattr(_lib.lib(), fn_name)(_lib.default_context().handle, _lib.ptr(a), ctypes.c_size_t(a.size), _lib.ptr(out)))
    return out


def lin_srgb_to_srgb(rgb: np.ndarray) -> np.ndarray:
    """Clip to [0,1] and apply the sRGB transfer curve (transform.py:89-99).  Always returns a real ndarray: given the lazy
    result of to_lin_srgb() it encodes on the GPU and downloads once (the end of the README recipe)."""
    if isinstance(rgb, DeferredImage) and rgb.pending and lazy_enabled() and rgb.plan["tail"] == 1:
        return rgb.with_tail(2, rgb.plan["M"]).numpy()      # deferred mode: the whole README recipe as ONE banded host call
    if _on_device(rgb):
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("expected an (H, W, 3) RGB image")
        tmp = DeviceArray(rgb.context, rgb.shape)
        _lib.check(_lib.lib().pysp_lin_srgb_to_srgb_dev(rgb.context.handle, rgb.ptr, ctypes.c_size_t(rgb.size), tmp.ptr))
        out = tmp.numpy()
        tmp.release()
        return out
    return _flat("pysp_lin_srgb_to_srgb_f32", _rgb_image(rgb))


def srgb_to_lin_srgb(srgb: np.ndarray) -> np.ndarray:
    """Clip to [0,1] and remove the sRGB transfer curve (transform.py:101-111)."""
    return _flat("pysp_srgb_to_lin_srgb_f32", _rgb_image(srgb))
