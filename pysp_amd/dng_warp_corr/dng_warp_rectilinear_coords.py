"""Drop-in for the Cython unit dng_warp_corr/dng_warp_rectilinear_coords.pyx (pyx:67-96)."""
import numpy as np

from .. import _lib


def _table(seed, kr0, kr1, kr2, kr3, kt0, kt1, width, height, cx, cy, scale) -> np.ndarray:
    width, height = int(width), int(height)
    out = _lib.empty_f32((height, width, 2))
    sp = None
    if seed is not None:
        if seed.dtype != np.float32 or seed.shape != (height, width, 2):
            raise ValueError("Buffer dtype mismatch, expected a (height, width, 2) float32 seed")
        seed = np.ascontiguousarray(seed)
        sp = _lib.ptr(seed)
    _lib.check(_lib.lib().pysp_warp_table_f32(_lib.default_context().handle, kr0, kr1, kr2, kr3, kt0, kt1, width, height, cx, cy, scale,
                                              sp, _lib.ptr(out)))
    return out


def compute_remapping_table(kr0, kr1, kr2, kr3, kt0, kt1, width, height, cam_center_norm_x, cam_center_norm_y, scale) -> np.ndarray:
    """Absolute sample coordinates (x', y') of the DNG 1.4 WarpRectilinear polynomial for every pixel."""
    return _table(None, kr0, kr1, kr2, kr3, kt0, kt1, width, height, cam_center_norm_x, cam_center_norm_y, scale)


def compute_offset_remapping_table(seed, kr0, kr1, kr2, kr3, kt0, kt1, width, height, cam_center_norm_x, cam_center_norm_y, scale) -> np.ndarray:
    """Same polynomial evaluated at the coordinates stored in `seed` (a prior mapping)."""
    return _table(seed, kr0, kr1, kr2, kr3, kt0, kt1, width, height, cam_center_norm_x, cam_center_norm_y, scale)
