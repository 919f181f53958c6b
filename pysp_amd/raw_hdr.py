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

    fused = _lib.empty_f32((H, W))
    count = _lib.empty((H, W), np.int32)
    ptrs = (ctypes.c_void_p * K)(*[f.ctypes.data for f in frames])
    _lib.check(_lib.lib().pysp_fuse_raw_f32(_lib.default_context().handle, ptrs, K, H, W,
                                            off32.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            np.ascontiguousarray(bias).ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            kmax, _lib.ptr(fused), _lib.ptr(count)))

    first = in_exposures[0]
    hdr_image = RawRggbBayerData(fused, first.cam_wb.copy(), target_ev, max(ev_offsets), first.source_pattern)
    hdr_image.set_hdr(True)
    return (hdr_image, count)


def fuse_exposures_from_debayer(in_exposures: List[RawDemosaicData], target_ev: Optional[float] = None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Camera-space fusion of debayered exposures to HDR linear sRGB (+ contribution counts).

    Weights are taken in pure sensor space (wb_undo), pixels are summed white balanced (wb_apply), exactly
    as raw_hdr.py:54-72 does; like the reference this leaves every exposure white balanced with its image
    replaced by the undo/apply round trip.  One kernel, including the trailing cam_to_lin_srgb.
    """
    valid = [e for e in in_exposures if e.is_valid()]
    if len(valid) == 0:
        return None
    K = len(valid)
    target_ev, ev_offsets = _ev_offsets([e.current_ev for e in valid], target_ev)
    first = valid[0]
    if any(e._wb_normalized for e in valid):
        raise NotImplementedError("normalised white balance (wb_norm=True) is not produced by any demosaic path and is not fused on the GPU")
    coeff = np.ascontiguousarray(np.stack([np.asarray(e._wb_coeff, dtype=np.float32)[:3] for e in valid]))
    applied = (ctypes.c_int * K)(*[int(bool(e._wb_applied)) for e in valid])
    bias = np.array([1.6 ** (-0.1 * off) for off in ev_offsets]).astype(np.float32)          # :60-61
    off32 = np.array(ev_offsets, dtype=np.float32)
    kmax = max(k for k, off in enumerate(ev_offsets) if off == np.max(ev_offsets))        # :67-68, last match wins
    from .colorize.transform import final_matrix
    from .device_array import DeviceArray, as_device, lazy_enabled
    M = _lib.mat9(final_matrix(in_exposures[0].mat_xyz))      # (raw_hdr.py:81 takes the first exposure's matrix, valid or not)
    # Device resident: an exposure that a demosaic left in HBM is fused there (no download + private copy + upload per exposure); host images are
    # uploaded once, straight from the caller's arrays, which stay untouched -- the round trip lands in new buffers, as the reference's
    # wb_undo() / wb_apply() rebind exposure.image to new arrays.
    # Exposures of ANOTHER context / device than the first device-resident one are brought over through the host (as_device downloads them, which
    # releases their device copy, and uploads to `ctx`): correct, one extra PCIe round trip for those exposures only.
    dev_imgs = [e._device_image() for e in valid]
    shape = tuple(dev_imgs[0].shape) if dev_imgs[0] is not None else tuple(first.image.shape)     # (no download just to ask)
    ctx = next((d.context for d in dev_imgs if d is not None), None) or _lib.default_context()
    vpp = ctypes.c_void_p * K
    fp = ctypes.POINTER(ctypes.c_float)
    npx = int(np.prod(shape, dtype=np.int64)) // 3
    lazy = lazy_enabled()
    for e, d in zip(valid, dev_imgs):
        if tuple((d if d is not None else e.image).shape) != tuple(shape):
            raise ValueError("all exposures must share one shape")
    try:
        d_in, d_rt = [], []
        for e, d in zip(valid, dev_imgs):
            if d is not None and d.context is ctx:
                d_in.append(d)
                d_rt.append(DeviceArray(ctx, shape))       # the exposure's own buffer may be shared with other holders: round trip into a new one
            else:
                up = as_device(d if d is not None else e.image, ctx)
                d_in.append(up)
                d_rt.append(up)                            # a private upload takes its round trip in place (the header allows d_frames_rt[k] == d_frames[k])
        d_out, d_cnt = DeviceArray(ctx, shape), DeviceArray(ctx, shape)          # the counts are int32 in a buffer of the same size
    except MemoryError:
        # 2K + 2 frames do not fit beside what the context already holds: the host entry point streams the exposures instead
        d_in = d_rt = None
        return _fuse_from_debayer_host(valid, shape, coeff, applied, off32, bias, kmax, M, lazy)
    with ctx.lock:
        _lib.check(_lib.lib().pysp_fuse_rgb_dev(ctx.handle, vpp(*[d.ptr.value for d in d_in]), vpp(*[d.ptr.value for d in d_rt]), K, ctypes.c_size_t(npx),
                                                coeff.ctypes.data_as(fp), applied, off32.ctypes.data_as(fp), bias.ctypes.data_as(fp), kmax, M, d_out.ptr, d_cnt.ptr))
    count = _lib.empty(shape, np.int32)               # allocated outside the context's lock (it may come from the page-locked pool)
    with ctx.lock:
        _lib.check(_lib.lib().pysp_dev_download(ctx.handle, _lib.ptr(count), d_cnt.ptr, ctypes.c_size_t(count.nbytes)))      # waits for the kernel: the uploads' sources may go
    del d_in, d_cnt
    for e, rt in zip(valid, d_rt):           # the state wb_undo(); wb_apply() leaves behind
        e.image = rt if lazy else rt.numpy()
        e._wb_applied = True
        e._wb_normalized = False
    fused = d_out if lazy else d_out.numpy()
    return (fused, count)


def _fuse_from_debayer_host(valid, shape, coeff, applied, off32, bias, kmax, M, lazy):
    """fuse_exposures_from_debayer through the host-buffer entry point (pysp_fuse_rgb_f32): the fallback when the device-resident form cannot
    allocate its 2K + 2 frames.  Same kernel, same bits; every exposure ends as a host ndarray holding its wb_undo / wb_apply round trip."""
    K = len(valid)
    frames = [np.array(e.image, dtype=np.float32, order="C", copy=True) for e in valid]      # private copies: the caller's arrays stay untouched
    out = _lib.empty_f32(shape)
    count = _lib.empty(shape, np.int32)
    vpp = ctypes.c_void_p * K
    fp = ctypes.POINTER(ctypes.c_float)
    npx = int(np.prod(shape, dtype=np.int64)) // 3
    ctx = _lib.default_context()
    with ctx.lock:
        _lib.check(_lib.lib().pysp_fuse_rgb_f32(ctx.handle, vpp(*[f.ctypes.data for f in frames]), K, ctypes.c_size_t(npx), coeff.ctypes.data_as(fp), applied,
                                                off32.ctypes.data_as(fp), bias.ctypes.data_as(fp), kmax, M, _lib.ptr(out), _lib.ptr(count), 1))
    for e, f in zip(valid, frames):
        e.image = f
        e._wb_applied = True
        e._wb_normalized = False
    return (out, count)
