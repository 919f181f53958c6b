"""DNG OpcodeList3 WarpRectilinear on the GPU (reference dng_warp_corr/chan_distortion_corr.py:11-121).

The per-plane coordinate table is evaluated inside the remap kernel and never materialised
(814 MB per plane at 100 MP).  Reading the opcode blob out of a DNG (`get_opcode_3_block`,
tifftools) is file I/O and stays outside this package.
"""
from __future__ import annotations

import ctypes
from struct import unpack
from typing import Optional

import numpy as np

from .. import _lib


def stack_warp_prior(demosaiced_image: np.ndarray, remap_r: Optional[np.ndarray], remap_g: Optional[np.ndarray],
                     remap_b: Optional[np.ndarray]) -> np.ndarray:
    """(H, W, 3, 2) prior from per-channel cv2.remap style maps; missing channels get the identity."""
    if remap_r is None or remap_g is None or remap_b is None:
        h, w = demosaiced_image.shape[:2]
        ident = np.empty((h, w, 2), np.float32)
        ident[..., 0] = np.arange(w, dtype=np.float32)[None, :]
        ident[..., 1] = np.arange(h, dtype=np.float32)[:, None]
        remap_r = ident if remap_r is None else remap_r
        remap_g = ident if remap_g is None else remap_g
        remap_b = ident if remap_b is None else remap_b
    return np.stack((remap_r, remap_g, remap_b), axis=2)


def _warp_rectilinear(image: np.ndarray, data: bytes, scale: float, prior) -> bool:
    if len(data) < 4:
        return False
    planes = int.from_bytes(data[:4], byteorder="big")
    if len(data) != 4 + 48 * planes + 16 or planes != image.shape[2]:
        return False
    coeffs = np.array([unpack(">6d", data[4 + 48 * p: 4 + 48 * (p + 1)]) for p in range(planes)], dtype=np.float64)
    cx, cy = unpack(">2d", data[4 + 48 * planes: 4 + 48 * planes + 16])
    H, W, _ = image.shape
    L, ctx = _lib.lib(), _lib.default_context()
    pr_all = None if prior is None else np.asarray(prior, dtype=np.float32)
    # The kernels work on C-contiguous float32 (H, W, 3) images.  The reference works in place on whatever it is handed --
    # the flipped / rot90 views image.py:181 returns for non-RGGB sensors, any plane count equal to the opcode's -- so
    # anything else is warped through a contiguous float32 copy, three planes at a time (a short last group repeats its
    # final plane), and written back through the view: same in-place semantics.
    direct = planes == 3 and image.dtype == np.float32 and image.flags.c_contiguous
    for p0 in range(0, planes, 3):
        idx = [min(p0 + k, planes - 1) for k in range(3)]
        buf = image if direct else np.ascontiguousarray(image[:, :, idx], dtype=np.float32)
        cf = np.ascontiguousarray(coeffs[idx])
        cptr = cf.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if pr_all is None:      # table evaluated inside the remap kernel, never materialised
            _lib.check(L.pysp_warp_rectilinear_f32(ctx.handle, _lib.ptr(buf), H, W, cptr, 3, cx, cy, scale))
        else:                   # seeded tables (compute_offset_remapping_table) from the prior mapping
            pr = np.ascontiguousarray(pr_all if direct else pr_all[:, :, idx, :])
            _lib.check(L.pysp_warp_rectilinear_prior_f32(ctx.handle, _lib.ptr(buf), H, W, cptr, 3, cx, cy, scale, _lib.ptr(pr)))
        if not direct:
            n = min(3, planes - p0)
            image[:, :, p0:p0 + n] = buf[:, :, :n]
    return True


def apply_opcode_3_warp(demosaiced_image: np.ndarray, ifd_opcode_3_data: bytes, scale: float = 1.0, prior: Optional[np.ndarray] = None):
    """Apply every WarpRectilinear (opcode 1) of an OpcodeList3 blob in place, in order; others are skipped."""
    assert prior is None or prior.shape == demosaiced_image.shape + (2,)
    count = int.from_bytes(ifd_opcode_3_data[:4], byteorder="big")
    offset = 4
    for _ in range(count):
        opcode_id, _ver, _flags, var_len = unpack(">4I", ifd_opcode_3_data[offset:offset + 16])
        offset += 16
        if opcode_id == 1:
            _warp_rectilinear(demosaiced_image, ifd_opcode_3_data[offset:offset + var_len], scale, prior)
        else:
            print("Unimplemented opcode %d" % opcode_id)
        offset += var_len
