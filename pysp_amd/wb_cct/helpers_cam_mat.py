"""Camera matrix containers and Bradford adaptation (reference wb_cct/helpers_cam_mat.py:7-37).
Host-side float64 3x3 algebra; the pixels never pass through here."""
from __future__ import annotations

import numpy as np

_XYZ_TO_LMS = np.array([[0.8951, 0.2664, -0.1614],
                        [-0.7502, 1.7135, 0.0367],
                        [0.0389, -0.0685, 1.0296]], dtype=np.float64)


def xy_to_XYZ(xy) -> np.ndarray:
    """CIE xy -> XYZ with Y = 1 (what colour.xy_to_XYZ returns for a 2-vector)."""
    x, y = float(xy[0]), float(xy[1])
    return np.array([x / y, 1.0, (1.0 - x - y) / y], dtype=np.float64)


def bradford_adapt_matrix(current_xyz: np.ndarray, target_xyz: np.ndarray) -> np.ndarray:
    gain = np.matmul(_XYZ_TO_LMS, target_xyz) / np.matmul(_XYZ_TO_LMS, current_xyz)
    return np.matmul(np.linalg.inv(_XYZ_TO_LMS), np.matmul(np.diag(gain), _XYZ_TO_LMS))


class ChromacityMat:
    """Read-only (matrix, white XYZ) pair."""

    def __init__(self, mat: np.ndarray, xyz: np.ndarray):
        self.mat = np.array(mat, copy=True)
        self.mat.setflags(write=False)
        self.xyz = np.array(xyz, copy=True)
        self.xyz.setflags(write=False)


class MatXyzToCamera(ChromacityMat):
    """XYZ -> camera matrix with the XYZ of the white it was optimised for."""

    def __init__(self, mat: np.ndarray, xyz: np.ndarray, series=None):
        super().__init__(mat, xyz)
        self.series = series

    def interpolate(self, next: "MatXyzToCamera", blend: float) -> np.ndarray:
        blend = np.clip(blend, 0.0, 1.0)
        return self.mat * (1 - blend) + (next.mat * blend)
