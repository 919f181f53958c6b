"""Linear RGB working spaces (reference colorize/rgb_space.py:7-56).  Host-side float64."""
from typing import Optional, Tuple, Union

import numpy as np

from ..wb_cct.helpers_cam_mat import bradford_adapt_matrix, xy_to_XYZ
from ..wb_cct.standard_ill import StandardIlluminant, get_chromacity_from_illuminant


class ArbitraryRgbColorspace:
    def __init__(self, primary_xy_r: Tuple[float, float], primary_xy_g: Tuple[float, float], primary_xy_b: Tuple[float, float],
                 whitepoint: StandardIlluminant):
        self._prim = (primary_xy_r, primary_xy_g, primary_xy_b)
        self._white = xy_to_XYZ(get_chromacity_from_illuminant(whitepoint))

    def mat_to_rgb(self, source_whitepoint=None) -> np.ndarray:
        return np.linalg.inv(self.mat_to_xyz(source_whitepoint))

    def mat_to_xyz(self, destination_whitepoint: Optional[Union[Tuple[float, float, float], StandardIlluminant]] = None) -> np.ndarray:
        """RGB -> XYZ with the primaries scaled so that RGB white hits the space's white, optionally
        Bradford-adapted to another white (pre-multiplied, rgb_space.py:47-50)."""
        m = np.array([[p[0] / p[1] for p in self._prim],
                      [1, 1, 1],
                      [(1 - p[0] - p[1]) / p[1] for p in self._prim]], dtype=np.float64)
        s = np.linalg.inv(m) @ self._white
        m[:, 0] *= s[0]
        m[:, 1] *= s[1]
        m[:, 2] *= s[2]
        if destination_whitepoint is None:
            return m
        if isinstance(destination_whitepoint, StandardIlluminant):
            dest = xy_to_XYZ(get_chromacity_from_illuminant(destination_whitepoint))
        else:
            dest = np.array(destination_whitepoint)
        if dest.ndim != 1 or dest.shape[0] != 3:
            raise ValueError("white point must be an XYZ triple")
        return bradford_adapt_matrix(self._white, dest) @ m


class LinRgbColorspace:
    REC709 = ArbitraryRgbColorspace((0.64, 0.33), (0.3, 0.6), (0.15, 0.06), StandardIlluminant.D65)
    REC2020 = ArbitraryRgbColorspace((0.708, 0.292), (0.170, 0.797), (0.131, 0.046), StandardIlluminant.D65)
