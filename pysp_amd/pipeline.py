"""Device-resident pipelines for batch callers (frames stay in HBM; torch is only the allocator).

Each method takes/returns torch CUDA tensors and enqueues through the `*_dev` entry points of the C ABI
on TORCH'S CURRENT STREAM for the device (the context is re-bound to it at every call,
`pysp_ctx_set_stream`): library kernels, torch ops on the same tensors, the caching allocator's reuse
of freed blocks and RCCL collectives are therefore ordered like any other torch work -- no host
synchronisation is needed between them, and none is done.  Nothing is copied to the host.

    pipe = DevicePipeline(0)
    srgb = pipe.demosaic_to_srgb(bayer_dev, wb, M)                 # README.md:55-63 recipe, fused
    srgb = pipe.hdr_stack_to_srgb(frames_dev, evs, cam_wb)          # raw_hdr.py:85-158 + README.md:141-158
    rgb  = pipe.demosaic_warp(bayer_dev, wb, M, coeffs, centre)     # AHD + dng_warp_corr opcode 1
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .colorize.transform import final_matrix


def _dp(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr())


class DevicePipeline:
    def __init__(self, device: int = 0):
        import torch   # noqa: F401  (imported first so that one HIP runtime serves both)
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.ctx = _lib.Context(device)
        self.L = _lib.lib()
        self._stream = -1
        self._enter()

    # every tensor handed in was produced on (or made visible to) torch's current stream: enqueue there too
    def _enter(self):
        s = int(self.torch.cuda.current_stream(self.device).cuda_stream)
        if s != self._stream:
            self.ctx.set_stream(s)
            self._stream = s

    def sync(self):
        """Wait for everything enqueued so far (only needed before host-side reads through raw pointers)."""
        self.ctx.sync()

    def _check_bayer(self, bayer):
        if bayer.dtype != self.torch.float32 or bayer.dim() != 2 or not bayer.is_contiguous() or bayer.device != self.device:
            raise ValueError("expected a contiguous float32 (H, W) mosaic on this pipeline's device")
        return int(bayer.shape[0]), int(bayer.shape[1])

    def demosaic(self, bayer, wb, M, quality: int = _lib.QUALITY_BEST, hdr: bool = False, stages: int = 1, out=None):
        H, W = self._check_bayer(bayer)
        self._enter()
        out = self.torch.empty((H, W, 3), dtype=self.torch.float32, device=self.device) if out is None else out
        _lib.check(self.L.pysp_demosaic_dev(self.ctx.handle, _dp(bayer), H, W, _lib.wb3(wb), _lib.mat9(M), quality, int(hdr), int(stages), _dp(out)))
        return out

    def demosaic_to_srgb(self, bayer, wb, M, quality: int = _lib.QUALITY_BEST, hdr: bool = False, stages: int = 1, reinhard: bool = False, out=None):
        H, W = self._check_bayer(bayer)
        self._enter()
        out = self.torch.empty((H, W, 3), dtype=self.torch.float32, device=self.device) if out is None else out
        _lib.check(self.L.pysp_pipeline_srgb_dev(self.ctx.handle, _dp(bayer), H, W, _lib.wb3(wb), _lib.mat9(M), quality, int(hdr), int(stages),
                                                 int(reinhard), _dp(out)))
        return out

    def batch(self, bayers: Sequence, wb, M, quality: int = _lib.QUALITY_FAST, hdr: bool = False, stages: int = 1, tail: int = 1, outs=None):
        """Frames that share camera parameters in one library call (BASELINE config 3 per rank); tail as in `raw_u16_to_rgb`."""
        if not bayers:
            return []
        H, W = self._check_bayer(bayers[0])
        for b in bayers:
            if self._check_bayer(b) != (H, W):
                raise ValueError("all frames of a batch must share one shape")
        outs = [self.torch.empty((H, W, 3), dtype=self.torch.float32, device=self.device) for _ in bayers] if outs is None else outs
        n = len(bayers)
        src = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bayers])
        dst = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
        self._enter()
        _lib.check(self.L.pysp_pipeline_batch_dev(self.ctx.handle, src, n, H, W, _lib.wb3(wb), _lib.mat9(M), int(quality), int(hdr), int(stages), int(tail), dst))
        return outs

    def raw_u16_to_rgb(self, raw_u16, black, sat, wb, M, quality: int = _lib.QUALITY_BEST, stages: int = 1, tail: int = 2, out=None):
        """uint16 sensor mosaic -> normalise (normalization.py:4-24, fused into the tile loader) -> demosaic ->
        colour tail (0 camera RGB, 1 linear sRGB, 2 sRGB, 3 Reinhard + sRGB).  2 B/px of input traffic."""
        if raw_u16.dtype != self.torch.uint16 and raw_u16.dtype != self.torch.int16:
            raise ValueError("expected a uint16 (H, W) mosaic")
        if raw_u16.dim() != 2 or not raw_u16.is_contiguous() or raw_u16.device != self.device:
            raise ValueError("expected a contiguous (H, W) mosaic on this pipeline's device")
        H, W = int(raw_u16.shape[0]), int(raw_u16.shape[1])
        self._enter()
        out = self.torch.empty((H, W, 3), dtype=self.torch.float32, device=self.device) if out is None else out
        bl = (ctypes.c_float * 4)(*[float(black[i]) for i in range(4)])
        sa = (ctypes.c_float * 4)(*[float(sat[i]) for i in range(4)])
        _lib.check(self.L.pysp_pipeline_u16_dev(self.ctx.handle, _dp(raw_u16), H, W, bl, sa, _lib.wb3(wb), _lib.mat9(M), quality, 0, int(stages), int(tail), _dp(out)))
        return out

    def fuse_raw(self, frames: Sequence, evs: Sequence[float], wb, target_ev: Optional[float] = None):
        """raw_hdr.py:85-158 on device mosaics: returns (fused mosaic, count, target_ev, lim_sat)."""
        K = len(frames)
        H, W = self._check_bayer(frames[0])
        for f in frames:
            if self._check_bayer(f) != (H, W):
                raise ValueError("all exposures must share one shape")
        if target_ev is None:
            target_ev = 0
            for ev in evs:
                target_ev += ev
            target_ev /= K
        offs = [2 ** (ev - target_ev) for ev in evs]
        wbc = np.asarray(wb, dtype=np.float32)
        site_w = np.array([wbc[0], wbc[1], wbc[2], wbc[1]], dtype=np.float32)
        bias = np.ascontiguousarray(np.stack([1.6 ** (-0.1 * np.abs(off * site_w)) for off in offs]).astype(np.float32))
        off32 = np.array(offs, dtype=np.float32)
        self._enter()
        fused = self.torch.empty((H, W), dtype=self.torch.float32, device=self.device)
        count = self.torch.empty((H, W), dtype=self.torch.int32, device=self.device)
        ptrs = (ctypes.c_void_p * K)(*[f.data_ptr() for f in frames])
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.check(self.L.pysp_fuse_raw_dev(self.ctx.handle, ptrs, K, H, W, off32.ctypes.data_as(fp), bias.ctypes.data_as(fp),
                                            int(np.argmax(offs)), _dp(fused), _dp(count)))
        return fused, count, target_ev, max(offs)

    def hdr_stack_to_srgb(self, frames: Sequence, evs: Sequence[float], cam_wb, stages: int = 1, out=None):
        """BASELINE config 4: K exposures -> raw fusion -> AHD (HDR metric) -> to_lin_srgb -> x/(1+x) -> sRGB."""
        wb = cam_wb.get_reciprocal_multipliers()
        fused, count, _, _ = self.fuse_raw(frames, evs, wb)
        M = final_matrix(cam_wb.get_matrix())
        H, W = int(fused.shape[0]), int(fused.shape[1])
        out = self.torch.empty((H, W, 3), dtype=self.torch.float32, device=self.device) if out is None else out
        _lib.check(self.L.pysp_pipeline_srgb_dev(self.ctx.handle, _dp(fused), H, W, _lib.wb3(wb), _lib.mat9(M), _lib.QUALITY_BEST, 1, int(stages), 1, _dp(out)))
        return out, count

    def warp(self, rgb, coeffs, centre: Tuple[float, float], scale: float = 1.0, out=None):
        """chan_distortion_corr.py:86-97 on a device (H,W,3) image (out of place)."""
        if rgb.dtype != self.torch.float32 or rgb.dim() != 3 or rgb.shape[2] != 3 or not rgb.is_contiguous():
            raise ValueError("expected a contiguous float32 (H, W, 3) image")
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        cf = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1)
        self._enter()
        out = self.torch.empty_like(rgb) if out is None else out
        _lib.check(self.L.pysp_warp_rectilinear_dev(self.ctx.handle, _dp(rgb), _dp(out), H, W, cf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                    cf.size // 6, float(centre[0]), float(centre[1]), float(scale)))
        return out

    def lens_fields(self, lens_model, shape):
        """Upload the two coordinate quadrants of a corr_ca lens model for frames of `shape` once; pass the result to
        `remove_ca` for every frame taken with that lens."""
        probe = np.zeros(shape, np.float32)
        inv = np.ascontiguousarray(lens_model.get_undistorted_quadrant(probe), dtype=np.float32)
        fwd = np.ascontiguousarray(lens_model.get_distorted_quadrant(probe), dtype=np.float32)
        return self.torch.from_numpy(inv).to(self.device), self.torch.from_numpy(fwd).to(self.device)

    def remove_ca(self, bayer, wb, fields_r=None, fields_b=None):
        """corr_ca/ca_removal.py:48-131 in place on a device-resident float32 mosaic; fields_* from `lens_fields`."""
        self._check_bayer(bayer)
        H, W = int(bayer.shape[0]), int(bayer.shape[1])
        for f in (fields_r, fields_b):
            if f is not None and (len(f) != 2 or any(t.shape != (H // 2, W // 2, 2) or t.dtype != self.torch.float32 or not t.is_contiguous() for t in f)):
                raise ValueError("lens fields must be two contiguous float32 (H/2, W/2, 2) tensors for this frame size")
        q = [_dp(fields_r[0]) if fields_r else None, _dp(fields_r[1]) if fields_r else None, _dp(fields_b[0]) if fields_b else None, _dp(fields_b[1]) if fields_b else None]
        self._enter()
        _lib.check(self.L.pysp_remove_ca_dev(self.ctx.handle, _dp(bayer), H, W, q[0], q[1], float(np.float32(wb[0])), q[2], q[3], float(np.float32(wb[2]))))
        return bayer

    def warp_rows(self, rgb, coeffs, centre: Tuple[float, float], row0: int, row1: int, out, scale: float = 1.0):
        """Output rows [row0, row1) of `warp` only; `rgb` and `out` are whole-frame (H,W,3) device buffers of which
        only the rows `warp_source_rows` names need to hold valid data (one band of a frame sharded over GPUs)."""
        if rgb.dtype != self.torch.float32 or rgb.dim() != 3 or rgb.shape[2] != 3 or not rgb.is_contiguous():
            raise ValueError("expected a contiguous float32 (H, W, 3) image")
        if out.shape != rgb.shape or out.dtype != rgb.dtype or not out.is_contiguous():
            raise ValueError("out must be a whole-frame buffer like the input")
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        cf = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1)
        self._enter()
        _lib.check(self.L.pysp_warp_rectilinear_rows_dev(self.ctx.handle, _dp(rgb), _dp(out), H, W, cf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                         cf.size // 6, float(centre[0]), float(centre[1]), float(scale), int(row0), int(row1)))
        return out

    def warp_source_rows(self, H: int, W: int, coeffs, centre: Tuple[float, float], row0: int, row1: int, scale: float = 1.0) -> Tuple[int, int]:
        """Source rows [s0, s1) the warp of output rows [row0, row1) reads (device reduction of the same coordinates)."""
        cf = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1)
        s0, s1 = ctypes.c_int(0), ctypes.c_int(0)
        self._enter()
        _lib.check(self.L.pysp_warp_source_rows(self.ctx.handle, int(H), int(W), cf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), cf.size // 6,
                                                float(centre[0]), float(centre[1]), float(scale), int(row0), int(row1), ctypes.byref(s0), ctypes.byref(s1)))
        return s0.value, s1.value

    def demosaic_warp(self, bayer, wb, M, coeffs, centre, stages: int = 3, scale: float = 1.0, rgb=None, out=None):
        """BASELINE config 5 on one GPU: AHD(postprocess_stages) then per-channel WarpRectilinear.  `rgb` / `out`: optional reusable (H, W, 3) device
        buffers for the demosaiced and the warped frame (1.2 GB each at 100 MP: a caller that streams frames keeps them)."""
        rgb = self.demosaic(bayer, wb, M, _lib.QUALITY_BEST, False, stages, out=rgb)
        return self.warp(rgb, coeffs, centre, scale, out=out)
