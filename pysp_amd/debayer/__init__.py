"""Demosaic entry points (reference debayer/__init__.py:1-3): debayer_ahd, debayer_eag, debayer_fast."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..base_types.image_base import RawDemosaicData, RawRggbBayerData_BaseType
from ..colorize.transform import final_matrix
from ..device_array import DeferredImage, DeviceArray, deferred_enabled, lazy_enabled
from .ahd_homogeneity import build_map  # noqa: F401


def _run(image: RawRggbBayerData_BaseType, quality: int, stages: int = 0) -> RawDemosaicData:
    bayer = _lib.f32c(image.sensor_scaled)
    if bayer.ndim != 2:
        raise ValueError("sensor_scaled must be a 2-D mosaic")
    H, W = bayer.shape
    if H < 2 or W < 2 or H % 2 or W % 2:
        raise ValueError("demosaic: mosaic dimensions must be even and >= 2 (got %dx%d)" % (H, W))
    wb = image.cam_wb.get_reciprocal_multipliers()
    mat = image.cam_wb.get_matrix()
    # The matrix only matters for AHD's homogeneity metric (ahd.py:46-48).
    M = _lib.mat9(final_matrix(mat)) if quality == _lib.QUALITY_BEST else None
    if deferred_enabled():
        # opt-in (pysp_amd.set_lazy("deferred")): nothing runs yet -- the README recipe collapses into one banded host call when its result is read
        Mf = final_matrix(mat) if quality == _lib.QUALITY_BEST else None
        rgb = DeferredImage(_lib.default_context(), bayer, wb, Mf, quality, bool(image.get_hdr()), int(stages), 0)
    elif lazy_enabled():
        # the result stays in HBM (DeviceArray) until somebody reads RawDemosaicData.image; to_lin_srgb() consumes it there
        ctx = _lib.default_context()
        src = DeviceArray.from_host(ctx, bayer)
        rgb = DeviceArray(ctx, (H, W, 3))
        _lib.check(_lib.lib().pysp_demosaic_dev(ctx.handle, src.ptr, H, W, _lib.wb3(wb), M, quality, int(bool(image.get_hdr())), int(stages), rgb.ptr))
        rgb._keepalive = bayer          # the upload is only enqueued: the mosaic must outlive it
        src.release()                   # back to the context's cache; stream order protects it until the kernels are done
    else:
        rgb = _lib.empty_f32((H, W, 3))
        _lib.check(_lib.lib().pysp_demosaic_f32(_lib.default_context().handle, _lib.ptr(bayer), H, W, _lib.wb3(wb), M, quality,
                                                int(bool(image.get_hdr())), int(stages), _lib.ptr(rgb)))
    out = RawDemosaicData(rgb, wb, wb_norm=False)
    out.mat_xyz = mat
    out.current_ev = image.current_ev
    return out


def debayer_ahd(image: RawRggbBayerData_BaseType, postprocess_stages: int = 1) -> RawDemosaicData:
    """Adaptive Homogeneity-Directed demosaic (ahd.py:14-170); HDR images take the luma/tonemap metric."""
    return _run(image, _lib.QUALITY_BEST, max(int(postprocess_stages), 0))


def debayer_eag(image: RawRggbBayerData_BaseType) -> RawDemosaicData:
    """Edge-Assisted-Gaussian demosaic (edge_assisted_gaussian.py:188-201)."""
    return _run(image, _lib.QUALITY_FAST)


def debayer_fast(image: RawRggbBayerData_BaseType) -> RawDemosaicData:
    """Draft demosaic: aligned quarter-resolution RGB, bilinear x2 (fast_resize.py:7-44)."""
    return _run(image, _lib.QUALITY_DRAFT)
