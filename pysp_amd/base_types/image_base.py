"""Containers for raw mosaics and demosaiced images, attribute-compatible with the reference's
base_types/image_base.py (:13-124); the pixel work they trigger runs on the GPU."""
from __future__ import annotations

import enum
from typing import Optional

import numpy as np

from .. import _lib
from ..colorize.transform import cam_to_lin_srgb
from ..const import QualityDemosaic
from ..device_array import DeviceArray
from ..wb_cct.helpers_cam_mat import MatXyzToCamera

BayerPattern = enum.IntEnum("BayerPattern", ("Rggb", "Bggr", "Grbg", "Gbrg"), module=__name__)


class RawDemosaicData:
    """(H, W, 3) float32 camera-space RGB produced by a demosaic, plus what is needed to colour it."""

    def __init__(self, image: np.ndarray, wb_coeff: np.ndarray, wb_norm: bool = False):
        self.image = image                 # white-balanced camera RGB (a property: see below)
        self.mat_xyz: Optional[MatXyzToCamera] = None
        self.current_ev: float = np.inf
        self._wb_coeff = wb_coeff          # reciprocal neutral multipliers, at least 3 entries
        self._wb_applied = True            # every demosaic path multiplies them in (ahd.py:77-80, eag.py:193-194)
        self._wb_normalized = wb_norm

    # `image` is the plain ndarray attribute of the reference (image_base.py:27) to every reader and writer.  Behind it a
    # demosaic result lives in HBM (DeviceArray) until the attribute is first READ: then it is downloaded once, the device copy
    # is dropped (the caller may change the ndarray in place, which the GPU copy would not see) and the ndarray is the image
    # from then on.  to_lin_srgb() / wb_apply() / wb_undo() use the device copy while it exists and never trigger the download.
    @property
    def image(self):
        if self._dev is not None:
            self._img = self._dev.numpy()
            self._dev.release()
            self._dev = None
        return self._img

    def _device_image(self):
        """The device copy while it is still THE image (None once a host copy has been handed out or the buffer released)."""
        d = self._dev
        return d if d is not None and d.on_device else None

    @image.setter
    def image(self, value):
        if isinstance(value, DeviceArray) and value.on_device:
            self._dev, self._img = value, None
        else:
            self._dev, self._img = None, (value.numpy() if isinstance(value, DeviceArray) else value)

    def __getstate__(self):
        """State for copy / deepcopy / pickle: the image as a host ndarray (a device buffer is never copied or sent to another process)."""
        # resolve through the property first: DeviceArray.numpy() releases the device buffer, so the ORIGINAL must switch to the host copy too
        # (a later wb_undo() / fusion on the original found a released _dev and _img None)
        _ = self.image
        return self.__dict__.copy()

    def is_valid(self) -> bool:
        """Image, coefficients, matrix and exposure value are all present."""
        # a DeviceArray whose device copy another holder released still resolves through `image` (its host copy): it counts as present
        have = (self._dev is not None or self._img is not None, self._wb_coeff is not None, self.mat_xyz is not None, self.current_ev != np.inf)
        return all(have)

    def _gpu_scale(self, undo: bool):
        dev = self._device_image()
        if dev is not None:                            # still in HBM: scale there, stay there
            dst = DeviceArray(dev.context, dev.shape)
            _lib.check(_lib.lib().pysp_wb_scale_dev(dev.context.handle, dev.ptr, dev.size // 3, _lib.wb3(self._wb_coeff), int(undo), dst.ptr))
            return dst
        src = _lib.f32c(self.image)
        dst = _lib.empty_f32(src.shape)
        _lib.check(_lib.lib().pysp_wb_scale_f32(_lib.default_context().handle, _lib.ptr(src), src.size // 3,
                                                _lib.wb3(self._wb_coeff), int(undo), _lib.ptr(dst)))
        return dst

    def wb_apply(self):
        """Multiply the coefficients in (float32) unless they already are (image_base.py:45-50)."""
        if self._wb_applied:
            return
        self.image = self._gpu_scale(undo=False)
        self._wb_applied = True

    def wb_undo(self):
        """Divide the coefficients out through float64 (image_base.py:52-60); drops any normalisation first."""
        if not self._wb_applied:
            return
        if self._wb_normalized:
            self.image = self.image * max(self._wb_coeff)
        self.image = self._gpu_scale(undo=True)
        self._wb_applied = self._wb_normalized = False

    def to_lin_srgb(self) -> np.ndarray:
        """Linear sRGB through the camera matrix, highlights clipped (image_base.py:62-64)."""
        self.wb_apply()
        dev = self._device_image()
        return cam_to_lin_srgb(dev if dev is not None else self.image, self.mat_xyz)


class RawCameraData_BaseType:
    """What every raw container carries: normalised sensor data, white balance, exposure, saturation limit."""

    sensor_scaled: Optional[np.ndarray]

    def __init__(self):
        self.sensor_scaled = None
        self.cam_wb = None
        self.current_ev = np.inf
        self.lim_sat = 1.0
        self._hdr_flag = False

    def set_hdr(self, is_hdr: bool):
        self._hdr_flag = is_hdr

    def get_hdr(self) -> bool:
        return self._hdr_flag

    def demosaic(self, quality: QualityDemosaic, postprocess_steps: int = 1) -> Optional[RawDemosaicData]:
        """Overridden by the concrete containers in pysp_amd.image."""
        return None


class RawBayerData_BaseType(RawCameraData_BaseType):
    """A mosaic in its native CFA orientation."""

    def __init__(self):
        super().__init__()
        self.sensor_pattern: Optional[BayerPattern] = None

    def to_rggb(self) -> Optional["RawRggbBayerData_BaseType"]:
        return None


class RawRggbBayerData_BaseType(RawCameraData_BaseType):
    """A mosaic already flipped/rotated so that it reads RGGB; remembers where it came from."""

    def __init__(self, sensor_scaled: np.ndarray, cam_wb, shot_ev: float, lim_sat: float, source_pattern=BayerPattern.Rggb):
        super().__init__()
        self.sensor_scaled, self.cam_wb = sensor_scaled, cam_wb
        self.current_ev, self.lim_sat = shot_ev, lim_sat
        self.source_pattern = source_pattern
