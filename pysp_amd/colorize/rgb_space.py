"""Linear RGB working spaces: primaries + white -> RGB->XYZ matrix, optionally Bradford-adapted.
Host-side float64 3x3 algebra (what the reference does in colorize/rgb_space.py:19-52)."""
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from ..wb_cct.helpers_cam_mat import bradford_adapt_matrix, xy_to_XYZ
from ..wb_cct.standard_ill import StandardIlluminant, get_chromacity_from_illuminant

XY = Tuple[float, float]
White = Union[Sequence[float], StandardIlluminant]


def _white_xyz(w: White) -> np.ndarray:
    if isinstance(w, StandardIlluminant):
        return xy_to_XYZ(get_chromacity_from_illuminant(w))
    xyz = np.array(w)
    if xyz.ndim != 1 or xyz.shape[0] != 3:
        raise ValueError("white point must be an XYZ triple or a StandardIlluminant")
    return xyz


class ArbitraryRgbColorspace:
    """An RGB space given by the xy chromaticities of its primaries and a standard-illuminant white."""

    def __init__(self, primary_xy_r: XY, primary_xy_g: XY, primary_xy_b: XY, whitepoint: StandardIlluminant):
        self._xy = np.array([primary_xy_r, primary_xy_g, primary_xy_b], dtype=np.float64)   # rows: R, G, B
        self._white = _white_xyz(whitepoint)

    def _unscaled(self) -> np.ndarray:
        x, y = self._xy[:, 0], self._xy[:, 1]
        return np.array([x / y, [1, 1, 1], (1 - x - y) / y], dtype=np.float64)   # columns: XYZ of each primary at Y = 1

    def mat_to_xyz(self, destination_whitepoint: Optional[White] = None) -> np.ndarray:
        """RGB -> XYZ.  Primaries are scaled so that RGB = (1,1,1) lands on the space's white; with a destination
        white the Bradford adaptation from the space's white is applied on the left (rgb_space.py:47-50)."""
        m = self._unscaled()
        gains = np.linalg.inv(m) @ self._white
        for col in range(3):
            m[:, col] *= gains[col]
        if destination_whitepoint is None:
            return m
        return bradford_adapt_matrix(self._white, _white_xyz(destination_whitepoint)) @ m

    def mat_to_rgb(self, source_whitepoint: Optional[White] = None) -> np.ndarray:
        return np.linalg.inv(self.mat_to_xyz(source_whitepoint))


class LinRgbColorspace:
    REC709 = ArbitraryRgbColorspace((0.64, 0.33), (0.3, 0.6), (0.15, 0.06), StandardIlluminant.D65)
    REC2020 = ArbitraryRgbColorspace((0.708, 0.292), (0.170, 0.797), (0.131, 0.046), StandardIlluminant.D65)
