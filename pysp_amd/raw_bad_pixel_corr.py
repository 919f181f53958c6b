"""Hot-pixel detection on the GPU (reference raw_bad_pixel_corr.py:30-65, :95-133).

Repair (`repair_bad_pixels`, cv2.inpaint Navier-Stokes, :135-152) is sparse host-side work and stays out of
scope."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import _lib


def find_erroneous_pixels_threshold(image, min_delta: float = 0.025, min_neighbour_count: int = 5) -> List[np.ndarray]:
    """Per colour plane (r, g1, b, g2): True where more than `min_neighbour_count` of the 8 same-colour
    neighbours are below `pixel - min_delta`."""
    bayer = _lib.f32c(image.sensor_scaled)
    H, W = bayer.shape
    masks = [np.empty((H // 2, W // 2), np.uint8) for _ in range(4)]
    _lib.check(_lib.lib().pysp_find_hot_pixels_f32(_lib.default_context().handle, _lib.ptr(bayer), H, W, float(min_delta), int(min_neighbour_count),
                                                   *[_lib.ptr(m) for m in masks]))
    return [m.view(np.bool_) for m in masks]


def find_shared_pixels(erroneous_mask: List[List[np.ndarray]], min_ratio: float = 0.1) -> Optional[List[np.ndarray]]:
    """Pixels flagged in at least ceil(n_images * min_ratio) of the per-image masks (host side: a few small sums)."""
    if len(erroneous_mask) == 0:
        return None
    n_chan = len(erroneous_mask[0])
    if any(len(m) != n_chan for m in erroneous_mask[1:]):
        return None
    need = np.ceil(len(erroneous_mask) * min_ratio)
    out = []
    for c in range(n_chan):
        votes = np.sum(np.array([m[c] for m in erroneous_mask]), axis=0, dtype=np.int16)
        out.append(votes >= need)
    return out
