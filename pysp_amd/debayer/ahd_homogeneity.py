"""Drop-in for the Cython unit debayer/ahd_homogeneity_cython.pyx (build_map, pyx:61-68)."""
import numpy as np

from .. import _lib


def build_map(lab: np.ndarray, k_pad: int, domain_k: int, is_vertical: bool) -> np.ndarray:
    """3x3 (2*k_pad+1) homogeneity count per pixel of a padded (Hp, Wp, 3) float32 Lab image.
    `domain_k` is accepted and ignored, as in the reference (recomputed at pyx:27)."""
    if lab.dtype != np.float32 or lab.ndim != 3:
        raise ValueError("Buffer dtype mismatch, expected a 3-D float32 array")   # what Cython's typed buffer raises
    a = np.ascontiguousarray(lab)
    Hp, Wp, _ = a.shape
    out = np.empty((Hp - 2 * k_pad, Wp - 2 * k_pad), np.float32)
    _lib.check(_lib.lib().pysp_build_map_f32(_lib.default_context().handle, _lib.ptr(a), Hp, Wp, int(k_pad), int(bool(is_vertical)), _lib.ptr(out)))
    return out
