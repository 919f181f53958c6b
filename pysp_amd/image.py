"""Raw image objects and demosaic dispatch (reference image.py:143-197).

File decoding (rawpy / exifread / tifftools, image.py:199-357) is outside the GPU hot path: build
`RawBayerData` / `RawRggbBayerData` from arrays.
"""
from __future__ import annotations

import numpy as np

from .base_types.image_base import BayerPattern, RawBayerData_BaseType, RawDemosaicData, RawRggbBayerData_BaseType
from .const import QualityDemosaic
from .debayer import debayer_ahd, debayer_eag, debayer_fast


def reversible_transform_rggb(sensor_data: np.ndarray, bayer_pattern: BayerPattern):
    """Flip/rotate so that the CFA reads RGGB; applying it twice is the identity (image.py:143-152)."""
    if bayer_pattern == BayerPattern.Rggb:
        return sensor_data
    if bayer_pattern == BayerPattern.Bggr:
        return np.rot90(sensor_data, k=2)
    if bayer_pattern == BayerPattern.Gbrg:
        return np.flip(sensor_data, axis=1)
    if bayer_pattern == BayerPattern.Grbg:
        return np.flip(sensor_data, axis=0)
    raise NotImplementedError(str(bayer_pattern) + " not implemented!")


class RawRggbBayerData(RawRggbBayerData_BaseType):
    def demosaic(self, quality: QualityDemosaic, postprocess_steps: int = 1) -> RawDemosaicData:
        """Demosaic to a new RawDemosaicData; `sensor_scaled` is never modified (image.py:156-183)."""
        if quality == QualityDemosaic.Best:
            out = debayer_ahd(self, postprocess_stages=postprocess_steps)
        elif quality == QualityDemosaic.Fast:
            out = debayer_eag(self)
        elif quality == QualityDemosaic.Draft:
            out = debayer_fast(self)
        else:
            raise NotImplementedError("Quality mode not implemented: %s" % str(quality))
        if self.source_pattern != BayerPattern.Rggb:      # RGGB: identity -- and reading out.image would download a result that may never be needed on the host
            out.image = reversible_transform_rggb(out.image, self.source_pattern)   # a view, like the reference
        return out

    debayer = demosaic          # README.md:62 spelling


class RawBayerData(RawBayerData_BaseType):
    def to_rggb(self) -> RawRggbBayerData:
        rggb = reversible_transform_rggb(self.sensor_scaled, self.sensor_pattern)
        return RawRggbBayerData(rggb, self.cam_wb.copy(), self.current_ev, self.lim_sat, self.sensor_pattern)

    def demosaic(self, quality: QualityDemosaic, postprocess_steps: int = 1) -> RawDemosaicData:
        return self.to_rggb().demosaic(quality, postprocess_steps)

    debayer = demosaic


class RawBayerDataFromRaw(RawBayerData):
    def __init__(self, filename_or_data):
        raise NotImplementedError("raw-file decoding (rawpy/exifread/tifftools) is outside the GPU hot path; "
                                  "fill a RawBayerData from arrays instead")


# README.md:57,61 spellings
RawRgbgData = RawBayerData
RawRgbgDataFromRaw = RawBayerDataFromRaw
