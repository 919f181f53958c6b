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


_BATCH_TAILS = {"image": 0, "lin_srgb": 1, "srgb": 2}


def debayer_batch(images, quality, postprocess_stages: int = 1, to: str = "image"):
    """A sequence of RawRggbBayerData of ONE geometry and ONE set of camera parameters through one banded transfer chain (pysp_pipeline_batch_f32): the loop
    `[raw.demosaic(q, n).to_lin_srgb() for raw in frames]` of README.md:55-63 / BASELINE config 3 with frame k+1 uploaded and computed while frame k's result
    is still on its way to the host.  Not part of the reference's API (which has no batch call); same bits as the loop.  `to`: "image" -> a list of
    RawDemosaicData (what demosaic() returns), "lin_srgb" -> a list of float32 arrays (... .to_lin_srgb()), "srgb" -> (lin_srgb_to_srgb of that)."""
    import ctypes
    from ..const import QualityDemosaic
    if to not in _BATCH_TAILS:
        raise ValueError("to must be 'image', 'lin_srgb' or 'srgb'")
    q = {QualityDemosaic.Draft: _lib.QUALITY_DRAFT, QualityDemosaic.Fast: _lib.QUALITY_FAST, QualityDemosaic.Best: _lib.QUALITY_BEST}.get(quality, quality)
    if q not in (_lib.QUALITY_DRAFT, _lib.QUALITY_FAST, _lib.QUALITY_BEST):
        raise NotImplementedError("Quality mode not implemented.")
    images = list(images)
    if not images:
        return []
    first = images[0]
    wb = np.asarray(first.cam_wb.get_reciprocal_multipliers(), dtype=np.float32)
    mat = first.cam_wb.get_matrix()
    Mf = final_matrix(mat)
    hdr = bool(first.get_hdr())
    mosaics = []
    shape0 = np.shape(first.sensor_scaled)
    for im in images:
        b = _lib.f32c(im.sensor_scaled)
        if b.ndim != 2 or b.shape != shape0:
            raise ValueError("debayer_batch: every frame must be a 2-D mosaic of the first frame's size")
        if b.shape[0] < 2 or b.shape[1] < 2 or b.shape[0] % 2 or b.shape[1] % 2:
            raise ValueError("demosaic: mosaic dimensions must be even and >= 2 (got %dx%d)" % b.shape)
        if bool(im.get_hdr()) != hdr or not np.array_equal(np.asarray(im.cam_wb.get_reciprocal_multipliers(), dtype=np.float32), wb) \
                or not np.array_equal(final_matrix(im.cam_wb.get_matrix()), Mf):
            raise ValueError("debayer_batch: every frame must share the first frame's white balance, matrix and HDR flag")
        mosaics.append(b)
    H, W = mosaics[0].shape
    outs = [_lib.empty_f32((H, W, 3)) for _ in mosaics]
    n = len(mosaics)
    pin = (ctypes.c_void_p * n)(*[m.ctypes.data for m in mosaics])
    pout = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    ctx = _lib.default_context()
    with ctx.lock:
        _lib.check(_lib.lib().pysp_pipeline_batch_f32(ctx.handle, pin, n, H, W, _lib.wb3(wb), _lib.mat9(Mf), q, int(hdr), max(int(postprocess_stages), 0) if q == _lib.QUALITY_BEST else 0,
                                                      _BATCH_TAILS[to], pout))
    if to != "image":
        return outs
    res = []
    for im, o in zip(images, outs):
        d = RawDemosaicData(o, wb, wb_norm=False)
        d.mat_xyz = im.cam_wb.get_matrix()
        d.current_ev = im.current_ev
        res.append(d)
    return res
