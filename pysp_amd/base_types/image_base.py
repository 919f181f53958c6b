"""Image containers (reference base_types/image_base.py:13-124)."""
from __future__ import annotations

from enum import IntEnum, auto
from typing import Optional

import numpy as np

from .. import _lib
from ..colorize.transform import cam_to_lin_srgb
from ..const import QualityDemosaic
from ..wb_cct.helpers_cam_mat import MatXyzToCamera


class BayerPattern(IntEnum):
    Rggb = auto()
    Bggr = auto()
    Grbg = auto()
    Gbrg = auto()


class RawDemosaicData:
    """RGB pixels after demosaicing: (H, W, 3) float32, camera space, white balance applied."""

    def __init__(self, image: np.ndarray, wb_coeff: np.ndarray, wb_norm: bool = False):
        self.image: np.ndarray = image
        self._wb_coeff: np.ndarray = wb_coeff
        self._wb_applied: bool = True
        self._wb_normalized: bool = wb_norm
        self.mat_xyz: Optional[MatXyzToCamera] = None
        self.current_ev: float = np.inf

    def is_valid(self) -> bool:
        return self.image is not None and self._wb_coeff is not None and self.mat_xyz is not None and self.current_ev != np.inf

    def _scale(self, coeff, undo: bool) -> np.ndarray:
        a = _lib.f32c(self.image)
        out = np.empty_like(a)
        _lib.check(_lib.lib().pysp_wb_scale_f32(_lib.default_context().handle, _lib.ptr(a), a.size // 3, _lib.wb3(coeff), int(undo), _lib.ptr(out)))
        return out

    def wb_apply(self):
        """image * coeff[:3] as float32, if not applied yet (image_base.py:45-50)."""
        if not self._wb_applied:
            self.image = self._scale(self._wb_coeff, undo=False)
            self._wb_applied = True

    def wb_undo(self):
        """Back to pure camera space through a float64 divide (image_base.py:52-60)."""
        if self._wb_applied:
            if self._wb_normalized:
                self.image = self.image * max(self._wb_coeff)
            self.image = self._scale(self._wb_coeff, undo=True)
            self._wb_applied = False
            self._wb_normalized = False

    def to_lin_srgb(self) -> np.ndarray:
        self.wb_apply()
        return cam_to_lin_srgb(self.image, self.mat_xyz)


class RawCameraData_BaseType:
    def __init__(self):
        self.sensor_scaled: np.ndarray = None
        self.cam_wb = None
        self.current_ev: float = np.inf
        self.lim_sat: float = 1.0
        self.__is_hdr: bool = False

    def set_hdr(self, is_hdr: bool):
        self.__is_hdr = is_hdr

    def get_hdr(self) -> bool:
        return self.__is_hdr

    def demosaic(self, quality: QualityDemosaic, postprocess_steps: int = 1) -> RawDemosaicData:
        return None


class RawBayerData_BaseType(RawCameraData_BaseType):
    def __init__(self):
        super().__init__()
        self.sensor_pattern: BayerPattern = None

    def to_rggb(self) -> "RawRggbBayerData_BaseType":
        return None


class RawRggbBayerData_BaseType(RawCameraData_BaseType):
    def __init__(self, sensor_scaled: np.ndarray, cam_wb, shot_ev: float, lim_sat: float, source_pattern=BayerPattern.Rggb):
        super().__init__()
        self.sensor_scaled = sensor_scaled
        self.cam_wb = cam_wb
        self.current_ev = shot_ev
        self.lim_sat = lim_sat
        self.source_pattern: BayerPattern = source_pattern
