"""Synthetic RGGB frames for tests and benchmarks.

scene = 0.25 + 0.2*sin(2*pi*x/257)*cos(2*pi*y/131) + 0.15*(37 px checker) + 0.02*N(0,1), sampled
onto the RGGB lattice with per-channel gains (0.5, 1.0, 0.7), clipped to [0,1], float32.
Frame i of a batch uses seed 1000+i.
"""
from __future__ import annotations

import numpy as np

XYZ_TO_CAM = np.array([[0.9, -0.3, -0.1], [-0.4, 1.2, 0.2], [-0.1, 0.2, 0.6]], dtype=np.float32)
NEUTRAL_MULTIPLIERS = np.array([0.5, 1.0, 0.7], dtype=np.float32)   # AsShotNeutral-like; WB coeffs = 1/these
D65_XY = (0.31272, 0.32903)


def rggb_frame(H: int, W: int, seed: int = 1000, scale: float = 1.0, clip_hi: bool = True) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    s = 0.25 + 0.2 * np.outer(np.cos(2 * np.pi * y / 131), np.sin(2 * np.pi * x / 257))
    s += 0.15 * (((x // 37)[None, :] + (y // 37)[:, None]) % 2)
    s = s.astype(np.float32)
    s += np.float32(0.02) * rng.standard_normal((H, W), dtype=np.float32)
    gains = np.array([[0.5, 1.0], [1.0, 0.7]], dtype=np.float32)
    s *= gains[(np.arange(H) % 2)[:, None], (np.arange(W) % 2)[None, :]]
    if scale != 1.0:
        s *= np.float32(scale)
    np.clip(s, 0, 1 if clip_hi else None, out=s)
    return s


def random_frame(H: int, W: int, seed: int = 0) -> np.ndarray:
    """Pure uniform noise: the worst case for the homogeneity vote (every decision is close)."""
    return np.random.default_rng(seed).random((H, W), dtype=np.float32)


def default_wb():
    """(CameraWhiteBalanceController, reciprocal multipliers) with the survey's probe values."""
    from .wb_cct.cam_wb import CameraWhiteBalanceController
    from .wb_cct.helpers_cam_mat import xy_to_XYZ
    return CameraWhiteBalanceController.from_matrix(XYZ_TO_CAM, xy_to_XYZ(D65_XY), NEUTRAL_MULTIPLIERS)
