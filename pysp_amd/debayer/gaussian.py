"""Photosite-aware kernel split (reference debayer/gaussian.py:6-54).  Host-side, exact rationals / 64; the GPU
kernels carry the TOP_LEFT and BOTTOM_RIGHT sets as constants (pysp_amd/csrc/demosaic_common.h)."""
from enum import Enum
from typing import Tuple

import numpy as np

CV2_DEFAULT_KERNEL_SIGMA = 1.0
CV2_DEFAULT_UNNORM_GAUSSIAN_KERNEL = np.outer([1, 4, 6, 4, 1], [1, 4, 6, 4, 1])


class BayerPatternPosition(Enum):
    TOP_LEFT = 0
    TOP_RIGHT = 1
    BOTTOM_LEFT = 2
    BOTTOM_RIGHT = 3


def get_rgbg_kernel(kernel: np.ndarray, base_position: BayerPatternPosition) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Kernels for the (TopLeft, TopRight, BottomLeft, BottomRight) target photosites when the real samples sit at
    `base_position`: every second tap of the odd square `kernel`, zero-padded to its own size on the side that faces
    away from the base, each normalised to sum 1."""
    k = np.asarray(kernel)
    if k.ndim == 3 and k.shape[2] == 1:
        k = k[:, :, 0]
    if k.ndim != 2 or k.shape[0] != k.shape[1] or k.shape[0] % 2 != 1:
        raise AssertionError("kernel must be square with odd side")
    base_row, base_col = divmod(base_position.value, 2)       # 0 = top / left
    out = []
    for target in BayerPatternPosition:
        t_row, t_col = divmod(target.value, 2)
        rows = k[0::2] if t_row == base_row else k[1::2]
        sub = rows[:, 0::2] if t_col == base_col else rows[:, 1::2]
        if t_col != base_col:                                   # pad a zero column on the far side
            zc = np.zeros((sub.shape[0], 1))
            sub = np.hstack([sub, zc]) if t_col == 0 else np.hstack([zc, sub])
        if t_row != base_row:                                   # pad a zero row on the far side
            zr = np.zeros((1, sub.shape[1]))
            sub = np.vstack([zr, sub]) if t_row == 1 else np.vstack([sub, zr])
        out.append(sub / sub.sum())
    return tuple(out)
