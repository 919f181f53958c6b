"""White-balance controller: the part the pixel path consumes (reference wb_cct/cam_wb.py:236-264).

Only `get_reciprocal_multipliers()`, `get_matrix()` and `copy()` are touched between the mosaic and
sRGB (ahd.py:45,48,73,168; edge_assisted_gaussian.py:191,199; fast_resize.py:19,42; image.py:193).
The illuminant solvers (`update_by_reference`, `update_by_temperature`, cam_wb.py:81-234) are host
side scalar maths on colour-science and are out of scope: they raise here.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .helpers_cam_mat import MatXyzToCamera


class CameraWhiteBalanceController:
    def __init__(self, mats: List[MatXyzToCamera], multipliers: np.ndarray, optimal_mat: Optional[MatXyzToCamera] = None):
        if len(mats) < 1:
            raise ValueError("at least one XYZ->camera calibration is required")
        self._mats = list(mats)
        # cam_wb.py:77-78 can leave float64 multipliers behind, which silently promotes the whole
        # pipeline; they are pinned to float32 here, as AsShotNeutral delivers them.
        self._multipliers = np.array(multipliers, dtype=np.float32, copy=True)
        self._optimal_mat = optimal_mat if optimal_mat is not None else self._mats[0]

    @classmethod
    def from_matrix(cls, xyz_to_cam: np.ndarray, white_xyz: np.ndarray, multipliers: np.ndarray) -> "CameraWhiteBalanceController":
        """Build a controller from an already solved matrix / white / neutral multipliers."""
        m = MatXyzToCamera(np.asarray(xyz_to_cam, dtype=np.float32), np.asarray(white_xyz, dtype=np.float64))
        return cls([m], multipliers, m)

    def get_reciprocal_multipliers(self) -> np.ndarray:
        return np.copy(1.0 / self._multipliers)

    def get_matrix(self) -> MatXyzToCamera:
        return self._optimal_mat

    def copy(self) -> "CameraWhiteBalanceController":
        mats = [MatXyzToCamera(m.mat, m.xyz) for m in self._mats]
        return CameraWhiteBalanceController(mats, self._multipliers, MatXyzToCamera(self._optimal_mat.mat, self._optimal_mat.xyz))

    def update_by_reference(self, *_a, **_k):
        raise NotImplementedError("illuminant solver is host-side colour-science code, out of scope for the GPU hot path")

    update_by_temperature = update_by_reference
